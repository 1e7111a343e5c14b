"""ctypes binding of plonk_amd/_lib/libplonk_hip.so (declared in include/plonk_hip.h).

There is no fallback: if the shared object is missing or does not load, importing the
backend raises.  Build it with `make -C plonk_amd/csrc` (hipcc, --offload-arch=gfx950)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libplonk_hip.so")

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_intp = ctypes.POINTER(ctypes.c_int)
c_floatp = ctypes.POINTER(ctypes.c_float)
c_sizep = ctypes.POINTER(ctypes.c_size_t)
c_ctx = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/plonk_hip.h declares
SIGNATURES = {
    "plonk_hip_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_ctx)]),
    "plonk_hip_ctx_destroy": (None, [c_ctx]),
    "plonk_hip_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "plonk_hip_last_error": (ctypes.c_char_p, [c_ctx]),
    "plonk_hip_ctx_set_window_bits": (ctypes.c_int, [c_ctx, ctypes.c_int]),
    "plonk_hip_ctx_stream": (ctypes.c_void_p, [c_ctx]),
    "plonk_hip_sync": (ctypes.c_int, [c_ctx]),
    "plonk_hip_malloc": (ctypes.c_int, [c_ctx, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "plonk_hip_free": (ctypes.c_int, [c_ctx, ctypes.c_void_p]),
    "plonk_hip_memcpy_h2d": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_memcpy_d2h": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_srs_load_projective": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t]),
    "plonk_hip_srs_size": (ctypes.c_int, [c_ctx, c_sizep]),
    "plonk_hip_srs_read": (ctypes.c_int, [c_ctx, ctypes.c_size_t, ctypes.c_size_t, c_u64p]),
    "plonk_hip_kzg_setup": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_kzg_setup_range": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, ctypes.c_size_t, c_u64p]),
    "plonk_hip_g1_sum_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_msm_bucket_layout": (ctypes.c_int, [c_ctx, c_sizep, c_sizep]),
    "plonk_hip_msm_buckets_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_msm_reduce_bucket_slice_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_kzg_commit": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_kzg_commit_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_kzg_commit_batch": (
        ctypes.c_int, [c_ctx, ctypes.POINTER(c_u64p), c_sizep, ctypes.c_size_t, c_u64p]),
    "plonk_hip_kzg_open": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, c_u64p, c_u64p]),
    "plonk_hip_kzg_open_many": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p, c_u64p]),
    "plonk_hip_kzg_setup_tau_g2": (ctypes.c_int, [c_u64p, ctypes.c_char_p]),
    "plonk_hip_kzg_verify": (ctypes.c_int, [ctypes.c_char_p, c_u64p, c_u64p, c_u64p, c_u64p, c_intp]),
    "plonk_hip_kzg_verify_batch": (
        ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, c_u64p, c_u64p, c_u64p, c_u64p, c_u64p, c_intp]),
    "plonk_hip_profile": (ctypes.c_int, [c_ctx, ctypes.c_int]),
    "plonk_hip_timer_start": (ctypes.c_int, [c_ctx]),
    "plonk_hip_timer_stop": (ctypes.c_int, [c_ctx, c_floatp]),
    "plonk_hip_msm_phase_ms": (ctypes.c_int, [c_ctx, c_floatp]),
    "plonk_hip_msm_phase_name": (ctypes.c_char_p, [ctypes.c_int]),
    "plonk_hip_bench_mad_rate": (ctypes.c_int, [c_ctx, ctypes.POINTER(ctypes.c_double)]),
    "plonk_hip_bench_mad_mix": (
        ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                       ctypes.POINTER(ctypes.c_double)]),
    "plonk_hip_bench_fq_mul_rate": (
        ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "plonk_hip_msm_plan": (ctypes.c_int, [c_ctx, c_intp, c_intp, c_sizep]),
    "plonk_hip_fr_ntt": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int]),
    "plonk_hip_fr_ntt_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int]),
    "plonk_hip_fr_coset_ntt": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_uint32, ctypes.c_size_t, c_u64p, ctypes.c_int]),
    "plonk_hip_fr_coset_ntt_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, c_u64p, ctypes.c_int]),
    "plonk_hip_fr_scale_by_powers": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_eval": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, c_u64p]),
    "plonk_hip_fr_div_vanishing": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_quotient_coset": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_uint32, ctypes.c_size_t, c_u64p, c_intp]),
    "plonk_hip_fr_quotient_coset_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_srs_load_compressed": (
        ctypes.c_int, [c_ctx, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "plonk_hip_g1_decompress": (
        ctypes.c_int, [c_ctx, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, c_u64p, ctypes.POINTER(ctypes.c_size_t)]),
    "plonk_hip_g1_compress": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, ctypes.c_char_p]),
    "plonk_hip_msm_plan_for": (
        ctypes.c_int, [ctypes.c_size_t, ctypes.c_int, c_intp, c_intp, ctypes.POINTER(ctypes.c_size_t)]),
    "plonk_hip_ntt_plan": (
        ctypes.c_int, [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32),
                       ctypes.POINTER(ctypes.c_uint32)]),
    "plonk_hip_bench_op_rate": (ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "plonk_hip_bench_gather_rate": (ctypes.c_int, [c_ctx, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]),
    "plonk_hip_bench_sort_rank_mode": (ctypes.c_int, [c_ctx]),
    "plonk_hip_bench_skip_sort": (ctypes.c_int, [c_ctx, ctypes.c_int]),
    "plonk_hip_bench_msm_tuning": (ctypes.c_int, [c_ctx, ctypes.c_uint32, ctypes.c_int]),
    "plonk_hip_bench_batch_affine": (
        ctypes.c_int, [c_ctx, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), c_u64p]),
    "plonk_hip_fr_ntt_mixed": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]),
    "plonk_hip_fr_ntt_mixed_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]),
    "plonk_hip_fr_eval_many": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_evaluate_over_omega": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, ctypes.c_uint32, c_u64p]),
    "plonk_hip_fr_prefix_product": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t]),
    "plonk_hip_fr_prefix_sum": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t]),
    "plonk_hip_fr_batch_inverse": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t]),
    "plonk_hip_fr_linear_combination": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p, c_u64p, c_u64p, ctypes.c_int,
                       c_u64p]),
    "plonk_hip_memcpy_d2d": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_memset_zero": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_kzg_open_many_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p, c_u64p]),
    "plonk_hip_kzg_commit_many_dev": (
        ctypes.c_int, [c_ctx, ctypes.POINTER(ctypes.c_void_p), c_sizep, ctypes.c_size_t, c_u64p]),
    "plonk_hip_kzg_commit_open_many_dev": (
        ctypes.c_int, [c_ctx, ctypes.POINTER(ctypes.c_void_p), c_sizep, ctypes.POINTER(ctypes.c_uint8), c_u64p,
                       ctypes.c_size_t, c_u64p, c_u64p]),
    "plonk_hip_fr_scale_by_powers_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_poly_mul_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_fr_linear_combination_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, c_u64p, c_u64p,
                       c_u64p, ctypes.c_int, ctypes.c_void_p]),
    "plonk_hip_fr_div_by_roots_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_fr_div_vanishing_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_fr_respects_permutation_dev": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t,
                       ctypes.POINTER(ctypes.c_int)]),
    "plonk_hip_fr_is_zero_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]),
    "plonk_hip_fr_ratio_prefix_product_dev": (ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_fr_ratio_prefix_product_dev_noalloc": (
        ctypes.c_int, [c_ctx, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_hip_fr_ratio_prefix_product": (ctypes.c_int, [c_ctx, c_u64p, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_poly_mul": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_vanishing_from_roots": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_div_by_roots": (
        ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_interpolate": (ctypes.c_int, [c_ctx, c_u64p, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_interpolate_geometric": (ctypes.c_int, [c_ctx, c_u64p, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_fr_interpolate_geometric_dev": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_hip_multi_create": (ctypes.c_int, [c_intp, ctypes.c_int, ctypes.POINTER(c_ctx)]),
    "plonk_hip_multi_destroy": (None, [c_ctx]),
    "plonk_hip_multi_last_error": (ctypes.c_char_p, [c_ctx]),
    "plonk_hip_multi_device_count": (ctypes.c_int, [c_ctx]),
    "plonk_hip_multi_set_window_bits": (ctypes.c_int, [c_ctx, ctypes.c_int]),
    "plonk_hip_multi_srs_load_projective": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t]),
    "plonk_hip_multi_kzg_setup": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_multi_srs_size": (ctypes.c_int, [c_ctx, c_sizep]),
    "plonk_hip_multi_srs_range": (ctypes.c_int, [c_ctx, ctypes.c_int, c_sizep, c_sizep]),
    "plonk_hip_multi_kzg_commit": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p]),
    "plonk_hip_multi_kzg_commit_batch": (
        ctypes.c_int, [c_ctx, ctypes.POINTER(c_u64p), c_sizep, ctypes.c_size_t, c_u64p]),
    "plonk_hip_multi_kzg_open": (ctypes.c_int, [c_ctx, c_u64p, ctypes.c_size_t, c_u64p, c_u64p, c_u64p]),
}

_lib = None


def load_library():
    """Loads the library and types every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C plonk_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
