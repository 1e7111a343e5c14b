// Development probe: which hardware slots do the workgroups of a launch with two 74 KB-LDS blocks per CU
// get?  Prints HW_ID fields of the first blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k_probe(uint32_t *out, uint64_t *when) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        when[blockIdx.x] = wall_clock64();
    }
    // keep the block resident for a while so that the first wave of blocks fills the GPU
    for (int i = 0; i < 20; i++) __builtin_amdgcn_s_sleep(127);
    if (lds[(threadIdx.x + 1) & 255] == 12345) out[0] = 0;
}
int main() {
    const int blocks = 2048;
    uint32_t *d;
    uint64_t *w;
    hipMalloc(&d, blocks * 4);
    hipMalloc(&w, blocks * 8);
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 74 * 1024, 0, d, w);
    std::vector<uint32_t> h(blocks);
    std::vector<uint64_t> t(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), w, blocks * 8, hipMemcpyDeviceToHost);
    uint64_t t0 = t[0];
    for (int i = 0; i < blocks; i++) if (t[i] < t0) t0 = t[i];
    for (int i = 0; i < blocks; i += (i < 80 ? 1 : 37)) {
        uint32_t v = h[i];
        printf("block %4d t=%6llu wave %u simd %u pipe %u cu %u sh %u se %u tg %u vm %u queue %u\n", i,
               (unsigned long long)(t[i] - t0), v & 15, (v >> 4) & 3, (v >> 6) & 3, (v >> 8) & 15, (v >> 12) & 1,
               (v >> 13) & 7, (v >> 16) & 15, (v >> 20) & 15, (v >> 24) & 7);
    }
    // how many of the first 512 blocks have an odd TG_ID, and do (cu, sh, se, xcd = block % 8) pairs split evenly?
    int odd = 0;
    for (int i = 0; i < 512; i++) odd += (h[i] >> 16) & 1;
    printf("odd tg among the first 512: %d\n", odd);
    return 0;
}
