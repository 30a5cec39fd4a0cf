"""ctypes binding of libzkmi.so (the C ABI in include/zkmi.h).

There is no CPU fallback: if the in-tree library is missing or no gfx950 device is
present, every compute call raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C zk_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkmi.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)

STATUS_NAMES = {
    0: "ZK_OK", 1: "ZK_E_NOT_POW2", 2: "ZK_E_POINTS_LEN", 3: "ZK_E_LEN_MISMATCH", 4: "ZK_E_LAYER_OOB",
    5: "ZK_E_DEGREE", 6: "ZK_E_ALLOC", 7: "ZK_E_HIP", 8: "ZK_E_ARG", 9: "ZK_E_NO_DEVICE", 10: "ZK_E_COMM", 11: "ZK_E_TOO_LARGE",
}

# every symbol include/zkmi.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ("zk_strerror", C.c_char_p, [C.c_int]),
    ("zk_last_error", C.c_char_p, [_VP]),
    ("zk_version", C.c_char_p, []),
    ("zk_fq_from_u64", None, [C.c_uint64, u64p]),
    ("zk_fq_to_le32", None, [u64p, u8p]),
    ("zk_fq_from_le32_mod_order", None, [u8p, u64p]),
    ("zk_fq_add", None, [u64p, u64p, u64p]),
    ("zk_fq_sub", None, [u64p, u64p, u64p]),
    ("zk_fq_mul", None, [u64p, u64p, u64p]),
    ("zk_unipoly_evaluate", None, [u64p, C.c_size_t, u64p, u64p]),
    ("zk_ctx_create", C.c_int, [C.c_int, C.POINTER(_VP)]),
    ("zk_ctx_create_on_stream", C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    ("zk_ctx_destroy", None, [_VP]),
    ("zk_ctx_stream", _VP, [_VP]),
    ("zk_ctx_synchronize", C.c_int, [_VP]),
    ("zk_transcript_new", _VP, []),
    ("zk_transcript_clone", _VP, [_VP]),
    ("zk_transcript_free", None, [_VP]),
    ("zk_transcript_append", C.c_int, [_VP, C.c_char_p, C.c_size_t]),
    ("zk_transcript_sample_challenge", C.c_int, [_VP, u64p]),
    ("zk_transcript_sample_n_challenges", C.c_int, [_VP, C.c_size_t, u64p]),
    ("zk_transcript_peek_digest", C.c_int, [_VP, u8p]),
    ("zk_transcript_selftest", C.c_int, [C.c_size_t]),
    ("zk_mle_upload_mont", C.c_int, [_VP, u64p, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_upload_le32", C.c_int, [_VP, u8p, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_wrap_device", C.c_int, [_VP, _VP, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_generate", C.c_int, [_VP, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_generate_strided", C.c_int, [_VP, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_upload_mont_shard", C.c_int, [_VP, u64p, C.c_size_t, C.c_int, C.c_int, C.POINTER(_VP)]),
    ("zk_mle_clone", C.c_int, [_VP, C.POINTER(_VP)]),
    ("zk_mle_free", None, [_VP]),
    ("zk_mle_len", C.c_size_t, [_VP]),
    ("zk_mle_device_ptr", _VP, [_VP]),
    ("zk_mle_download_mont", C.c_int, [_VP, u64p]),
    ("zk_mle_to_le32", C.c_int, [_VP, u8p]),
    ("zk_mle_partially_evaluate", C.c_int, [_VP, C.c_size_t, u64p, C.POINTER(_VP)]),
    ("zk_mle_quotient_remainder", C.c_int, [_VP, C.c_size_t, u64p, C.POINTER(_VP), C.POINTER(_VP)]),
    ("zk_mle_minus", C.c_int, [_VP, u64p, C.POINTER(_VP)]),
    ("zk_mle_evaluate", C.c_int, [_VP, u8p, u64p, C.c_size_t, C.POINTER(_VP)]),
    ("zk_mle_evaluation_sum", C.c_int, [_VP, u64p]),
    ("zk_mle_round_sums", C.c_int, [_VP, u64p]),
    ("zk_sumcheck_prove_mle", C.c_int, [_VP, _VP, C.c_int, _VP, u64p, u64p, u64p]),
    ("zk_sumcheck_prove_sum2x2", C.c_int, [_VP, C.POINTER(_VP), u64p, _VP, u64p, u64p]),
    ("zk_sumcheck_prove_prod_plus_linear", C.c_int, [_VP, C.POINTER(_VP), u64p, _VP, u64p, u64p]),
    ("zk_sumcheck_rounds_from_block_sums", C.c_int, [_VP, u64p, C.c_int, C.c_int, u64p, u64p, u64p, u64p]),
    ("zk_shard_begin", C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    ("zk_shard_end", None, [_VP]),
    ("zk_shard_vars_left", C.c_int, [_VP]),
    ("zk_shard_next_batch", C.c_int, [_VP, C.c_int]),
    ("zk_shard_copy_table", C.c_int, [_VP, _VP]),
    ("zk_shard_block_sums", C.c_int, [_VP, C.c_int, _VP]),
    ("zk_shard_fold", C.c_int, [_VP, C.c_int, u64p, C.c_int, _VP]),
    ("zk_fold_block_sums_noalloc", C.c_int, [_VP, _VP, C.c_size_t, C.c_int, u64p, _VP, C.c_int, _VP]),
    ("zk_shard4_begin", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP)]),
    ("zk_shard4_end", None, [_VP]),
    ("zk_shard4_len", C.c_size_t, [_VP]),
    ("zk_shard4_eval", C.c_int, [_VP, _VP]),
    ("zk_shard4_fold", C.c_int, [_VP, u64p, _VP]),
    ("zk_shard4_sum9", C.c_int, [_VP, _VP]),
    ("zk_sumcheck_round_pair_from_sum9", C.c_int, [_VP, u64p, C.c_int, u64p, u64p]),
    ("zk_shard4_fold2", C.c_int, [_VP, u64p, _VP]),
    ("zk_shard4_can_triple", C.c_int, [_VP]),
    ("zk_shard4_cross64", C.c_int, [_VP, _VP]),
    ("zk_sumcheck_rounds_from_cross64", C.c_int, [_VP, u64p, C.c_int, u64p, u64p]),
    ("zk_shard4_fold3", C.c_int, [_VP, u64p, _VP, C.POINTER(C.c_int)]),
    ("zk_shard4_copy_tables", C.c_int, [_VP, _VP]),
    ("zk_sumcheck_round_from_evals", C.c_int, [_VP, u64p, C.c_int, u64p, u64p]),
    ("zk_sumcheck_prove_sum2x2_host", C.c_int, [_VP, u64p, C.c_size_t, u64p, u64p]),
    ("zk_comm_unique_id", C.c_int, [u8p]),
    ("zk_comm_init_rccl", C.c_int, [_VP, u8p, C.c_int, C.c_int]),
    ("zk_comm_init_host", C.c_int, [_VP, C.c_int, C.c_int, _VP, _VP]),
    ("zk_comm_destroy", None, [_VP]),
    ("zk_comm_rank", C.c_int, [_VP]),
    ("zk_comm_world", C.c_int, [_VP]),
    ("zk_sumcheck_prove_mle_sharded", C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, u64p, u64p, u64p]),
    ("zk_sumcheck_prove_sum2x2_sharded", C.c_int, [_VP, C.POINTER(_VP), u64p, _VP, C.c_int, u64p, u64p]),
    ("zk_sumcheck_partial_verify", C.c_int, [u64p, u64p, C.c_size_t, C.c_size_t, _VP, C.POINTER(C.c_int), u64p, u64p]),
    ("zk_sumcheck_verify_mle", C.c_int, [_VP, _VP, u64p, u64p, C.c_size_t, C.POINTER(C.c_int)]),
    ("zk_mle_w_add", C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    ("zk_mle_w_mul", C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    ("zk_mle_scalar_mul", C.c_int, [_VP, u64p, C.POINTER(_VP)]),
    ("zk_mle_add", C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    ("zk_circuit_create", C.c_int, [_VP, u64p, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(_VP)]),
    ("zk_circuit_free", None, [_VP]),
    ("zk_circuit_layer_count", C.c_size_t, [_VP]),
    ("zk_circuit_shape", C.c_int, [_VP, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("zk_circuit_evaluate", C.c_int, [_VP, _VP, C.POINTER(_VP)]),
    ("zk_circuit_gate_poly", C.c_int, [_VP, C.c_size_t, C.c_int, C.POINTER(_VP)]),
    ("zk_gkr_prove", C.c_int, [_VP, _VP, _VP, _VP, u64p, u64p, u64p, u64p, C.POINTER(C.c_double)]),
    ("zk_gkr_prove_sharded", C.c_int, [_VP, _VP, _VP, _VP, C.c_int, u64p, u64p, u64p, u64p, C.POINTER(C.c_double)]),
    ("zk_gkr_verify", C.c_int, [_VP, _VP, _VP, _VP, u64p, C.c_size_t, u64p, u64p, u64p, C.POINTER(C.c_int)]),
    ("zk_ctx_set_fold_batch", C.c_int, [_VP, C.c_int, C.c_int]),
    ("zk_ctx_set_wide_fold", C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    ("zk_ctx_set_tail_fold", C.c_int, [_VP, C.c_int, C.c_int]),
    ("zk_ctx_set_deg2_rounds_per_pass", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_deg2_lazy_sums", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_deg2_mfma_sums", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_deg2_triple", C.c_int, [_VP, C.c_int, C.c_int]),
    ("zk_ctx_set_deg2_fused", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_deg2_small_triple", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_deg2_first_triple", C.c_int, [_VP, C.c_int, C.c_int]),
    ("zk_ctx_set_deg2_fold3_fused", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_validate", C.c_int, [_VP, C.c_int]),
    ("zk_mle_count_noncanonical", C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    ("zk_ctx_set_gkr_strict_claims", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_dma_gates", C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    ("zk_ctx_set_deg2_raw_cross", C.c_int, [_VP, C.c_int]),
    ("zk_ctx_set_wide_mfma", C.c_int, [_VP, C.c_int, C.c_int, C.c_int]),
    ("zk_ctx_forms_used", C.c_uint, [_VP, C.c_int]),
    ("zk_prof_enable", C.c_int, [_VP, C.c_int]),
    ("zk_prof_reset", C.c_int, [_VP]),
    ("zk_prof_collect", C.c_int, [_VP]),
    ("zk_prof_read", C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("zk_bench_modmul", C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("zk_bench_mad", C.c_int, [_VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
]

_lib = None


class ZkError(RuntimeError):
    """A non-zero zk_status: where the reference panics, this is raised."""

    def __init__(self, status: int, detail: str = ""):
        self.status = status
        name = STATUS_NAMES.get(status, str(status))
        msg = lib().zk_strerror(status).decode() if _lib is not None else ""
        super().__init__(f"{name}: {msg}" + (f" [{detail}]" if detail else ""))


def lib() -> C.CDLL:
    """Load libzkmi.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (make -C zk_amd/csrc). "
                "zk_amd has no CPU path.")
        try:  # keep ONE HIP runtime in the process: torch's bundled libamdhip64 shares our SONAME
            import torch  # noqa: F401
        except Exception:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)  # AttributeError if the library does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status: int, ctx=None):
    if status != 0:
        detail = ""
        if ctx is not None and status in (7, 10, 6):
            detail = lib().zk_last_error(ctx).decode()
        raise ZkError(status, detail)
