// Key/value radix sort of the (bucket, entry) pairs of the MSM.  The sort itself is AMD's
// rocPRIM device radix sort (header-only, compiled here for gfx950); only the low `end_bit`
// key bits are sorted.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "context.hpp"

namespace plonk {

size_t sort_pairs_temp_bytes(size_t n, int end_bit) {
    size_t bytes = 0;
    uint32_t *k = nullptr, *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0, (unsigned)end_bit, (hipStream_t)0, false);
    return bytes;
}

int sort_pairs(plonk_hip_ctx *ctx, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
               uint32_t *vals_out, size_t n, int end_bit, void *temp, size_t temp_bytes) {
    PLONK_HIP_TRY(ctx, rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0,
                                                 (unsigned)end_bit, ctx->stream, false));
    return PLONK_HIP_OK;
}

}  // namespace plonk
