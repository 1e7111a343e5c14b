// Driver for the sanitizer build of the verifier's host code (tests/test_verify_sanitized.py): reads
// tau_g2, n and n openings (com, u, v, proof as the C ABI takes them) and runs every entry point of verify.hip.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "plonk_hip.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint8_t tau_g2[96];
    uint64_t n = 0;
    if (fread(tau_g2, 1, 96, f) != 96 || fread(&n, 8, 1, f) != 1) return 2;
    std::vector<uint64_t> coms(n * 18), us(n * 4), vs(n * 4), proofs(n * 18);
    if (fread(coms.data(), 8, n * 18, f) != n * 18 || fread(us.data(), 8, n * 4, f) != n * 4 ||
        fread(vs.data(), 8, n * 4, f) != n * 4 || fread(proofs.data(), 8, n * 18, f) != n * 18) return 2;
    fclose(f);
    int ok = 0;
    for (uint64_t i = 0; i < n; i++) {
        int rc = plonk_hip_kzg_verify(tau_g2, &coms[i * 18], &us[i * 4], &vs[i * 4], &proofs[i * 18], &ok);
        printf("single %llu rc %d ok %d\n", (unsigned long long)i, rc, ok);
    }
    int rc = plonk_hip_kzg_verify_batch(tau_g2, n, coms.data(), us.data(), vs.data(), proofs.data(), us.data(), &ok);
    printf("batch rc %d ok %d\n", rc, ok);
    vs[0] ^= 1;
    rc = plonk_hip_kzg_verify_batch(tau_g2, n, coms.data(), us.data(), vs.data(), proofs.data(), us.data(), &ok);
    printf("tampered rc %d ok %d\n", rc, ok);
    uint64_t tau[4] = {5, 0, 0, 0};
    uint8_t out[96];
    rc = plonk_hip_kzg_setup_tau_g2(tau, out);
    printf("tau_g2 rc %d first %02x\n", rc, out[0]);
    return 0;
}
