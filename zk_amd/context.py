"""Device context: one GPU, one HIP stream, scratch buffers (zk_ctx in include/zkmi.h)."""
from __future__ import annotations

import ctypes as C

from ._lib import check, lib

KERNELS = {"block_sums": 0, "multifold": 1, "fold": 2, "eval3": 3, "fold4_eval3": 4, "sum9": 5, "fold2x4_sum9": 6, "multifold_wide": 7, "multifold_tail": 8}


FORMS = ["sums_dma", "sums_reg", "wide_dma", "wide_dma_stream", "wide_reg", "wide_lds", "coop", "fold2_dma", "fold2_reg", "sum9_lazy", "sum9",
         "fold2_sum9", "wide_mfma", "sum9_mfma", "cross64", "fold3_dma", "fold3_cross64", "fold3_cross64_mfma", "fold_sum27", "fold3_reg", "pwq", "sum27_first"]


class Context:
    def __init__(self, device_id: int = 0, _stream=None):
        h = C.c_void_p()
        if _stream is None:
            check(lib().zk_ctx_create(device_id, C.byref(h)))
        else:
            check(lib().zk_ctx_create_on_stream(device_id, C.c_void_p(_stream), C.byref(h)))
        self.h = h
        self.device_id = device_id

    @classmethod
    def on_stream(cls, device_id: int, hip_stream: int):
        """Context that launches on a hipStream_t the caller owns (0 = the null stream)."""
        return cls(device_id, _stream=hip_stream or 0)

    @property
    def stream(self) -> int:
        """hipStream_t (as int) every kernel of this context is launched on."""
        return lib().zk_ctx_stream(self.h) or 0

    def synchronize(self):
        check(lib().zk_ctx_synchronize(self.h), self.h)

    def set_fold_batch(self, rounds_per_pass: int, variant: int = 0):
        """Tuning knob of the single-MLE prover (proof bytes do not depend on it).
        rounds_per_pass 1..8 (5..8: the wide kernels); variant 0 = eq-weight inner product (default), 1 = the fold tree
        (the reference's successive partially_evaluate calls, kept as the independent cross-check)."""
        check(lib().zk_ctx_set_fold_batch(self.h, rounds_per_pass, variant))

    def set_wide_fold(self, rounds: int = 8, min_log_out: int = 16, lds_weights_below_log: int = 15):
        """Passes over a large table bind up to `rounds` (5..8; 0 = off) variables while the folded table keeps at least
        2^min_log_out entries; wide passes with fewer than 2^lds_weights_below_log outputs make their weights per
        workgroup in LDS, larger ones read them as scalar loads.  Proof bytes do not depend on it."""
        check(lib().zk_ctx_set_wide_fold(self.h, rounds, min_log_out, lds_weights_below_log))

    def set_tail_fold(self, max_log_table: int = 16, out_below_log: int = 13):
        """Tables of <= 2^max_log_table entries bind eight rounds per pass (0 = off); passes with fewer than
        2^out_below_log outputs use one wave per output.  Proof bytes do not depend on it."""
        check(lib().zk_ctx_set_tail_fold(self.h, max_log_table, out_below_log))

    def set_deg2_rounds_per_pass(self, rounds: int):
        """Degree-2 prover schedule: 2 rounds per device pass (default) or 1.  Proof bytes do not depend on it."""
        check(lib().zk_ctx_set_deg2_rounds_per_pass(self.h, rounds))

    def set_deg2_lazy_sums(self, from_log: int = 18):
        """Degree-2 prover: nine-sum sweeps over >= 2^from_log quarter-entries use unreduced accumulation (negative: never)."""
        check(lib().zk_ctx_set_deg2_lazy_sums(self.h, from_log))

    def set_validate(self, on: bool = True):
        """Uploads of Montgomery limbs refuse entries >= p (default on)."""
        check(lib().zk_ctx_set_validate(self.h, int(on)))

    def set_gkr_strict_claims(self, on: bool = True):
        """zk_gkr_verify also checks that every layer's claim is bound to the output polynomial / the previous layer
        (the reference does not: gkr/src/verifier.rs:63-64).  Default off."""
        check(lib().zk_ctx_set_gkr_strict_claims(self.h, int(on)))

    def set_deg2_mfma_sums(self, from_log: int = 22):
        """Degree-2 prover: nine-sum sweeps over >= 2^from_log quarter-entries run on the matrix cores (64 = never)."""
        check(lib().zk_ctx_set_deg2_mfma_sums(self.h, from_log))

    def set_deg2_fold3_fused(self, dma_slots: int = 8):
        """Degree-2 prover, three-round passes followed by another: fold and the next cross sums in one sweep (0 = off; 3..5 = DMA slots per wave of the VALU-fold form; 8 = the fold on the matrix cores too)."""
        check(lib().zk_ctx_set_deg2_fold3_fused(self.h, dma_slots))

    def set_deg2_fused(self, below_log: int = 20):
        """Degree-2 prover: fold + nine sums of the result in one kernel while the folded tables have < 2^below_log entries (0 = never)."""
        check(lib().zk_ctx_set_deg2_fused(self.h, below_log))

    def set_deg2_small_triple(self, below_log: int = 20):
        """Degree-2 prover: THREE rounds per pass on small tables (fold + the 27 sums of the next three rounds in one kernel) while the
        folded tables have < 2^below_log entries (0 = never)."""
        check(lib().zk_ctx_set_deg2_small_triple(self.h, below_log))

    def set_deg2_first_triple(self, below_log: int = 19, linear_below_log: int = 20):
        """Degree-2 prover: the first pass as 27 sums (three rounds) for tables of < 2^below_log entries (0 = never); linear_below_log:
        the same for relations F G + H * 1 (the GKR phases), 0 = the GKR prover does not use its product-plus-linear kernels at all."""
        check(lib().zk_ctx_set_deg2_first_triple(self.h, below_log, linear_below_log))

    def set_deg2_triple(self, from_log: int = 20, ring: int = 4):
        """Degree-2 prover: passes over tables of >= 2^from_log entries serve three rounds each (8 x 8 cross sums per term on the
        matrix cores + one three-variable fold); 64 = never.  ring = LDS tiles of the sweep (3 or 4).  Defaults = the library's (kernels.h DmaGates)."""
        check(lib().zk_ctx_set_deg2_triple(self.h, from_log, ring))

    def set_deg2_raw_cross(self, on: bool = True):
        """Three-round sweeps leave raw integer digits for the second stage (default) or form field elements per workgroup; zk_ctx_set_deg2_raw_cross."""
        check(lib().zk_ctx_set_deg2_raw_cross(self.h, 1 if on else 0))

    def set_dma_gates(self, sums_from_log: int = 27, wide_stream_from_log: int = 22, fold2_from_log: int = 22):
        """Size thresholds (log2 entries, 64 = never) of the LDS-DMA forms of the big passes.  Proof bytes do not depend on it."""
        check(lib().zk_ctx_set_dma_gates(self.h, sums_from_log, wide_stream_from_log, fold2_from_log))

    def set_wide_mfma(self, from_log: int = 15, workgroups_per_cu: int = 1, depth: int = 2):
        """Wide fold passes with >= 2^from_log outputs (64 = never) contract on the matrix cores; depth = steps in flight per
        wave (2..4).  Proof bytes do not depend on it."""
        check(lib().zk_ctx_set_wide_mfma(self.h, from_log, workgroups_per_cu, depth))

    def forms_used(self, reset: bool = True) -> set:
        """Names of the kernel forms launched from this thread since the last reset."""
        bits = lib().zk_ctx_forms_used(self.h, int(reset))
        return {name for k, name in enumerate(FORMS) if bits >> k & 1}

    # ---- measurement hooks
    def prof_enable(self, on: bool = True):
        check(lib().zk_prof_enable(self.h, int(on)))

    def prof_reset(self):
        check(lib().zk_prof_reset(self.h))

    def prof_collect(self):
        check(lib().zk_prof_collect(self.h), self.h)

    def prof_read(self, kernel: str):
        n, ms, b = C.c_uint64(), C.c_double(), C.c_uint64()
        check(lib().zk_prof_read(self.h, KERNELS[kernel], C.byref(n), C.byref(ms), C.byref(b)))
        return {"launches": n.value, "ms": ms.value, "alg_bytes": b.value}

    def bench_modmul(self, blocks: int, iters: int) -> float:
        ms = C.c_double()
        check(lib().zk_bench_modmul(self.h, blocks, iters, C.byref(ms)), self.h)
        return ms.value

    def bench_mad(self, blocks: int, iters: int) -> float:
        """ms for blocks*256 lanes x iters*256 independent v_mad_u64_u32 (the int32-ALU issue ceiling)."""
        ms = C.c_double()
        check(lib().zk_bench_mad(self.h, blocks, iters, C.byref(ms)), self.h)
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            lib().zk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def default_context() -> Context:
    global _default
    if _default is None:
        _default = Context(0)
    return _default
