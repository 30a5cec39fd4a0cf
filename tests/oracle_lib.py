"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package zk_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)

import pyref  # noqa: E402

P = pyref.P
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_transcript_new.restype = C.c_void_p
        _lib.orc_circuit_new.restype = C.c_void_p
        _lib.orc_gkr_prove.restype = C.c_void_p
        _lib.orc_circuit_gate_poly.restype = C.c_size_t
        _lib.orc_gkr_proof_serialize.restype = C.c_size_t
    return _lib


# ------------------------------------------------------------------ conversions

def ints_to_mont(vals) -> np.ndarray:
    """canonical ints -> (n,4) uint64 Montgomery limbs (via Python big ints, not the C oracle)."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = pyref.to_mont_limbs(int(v) % P)
    return out


def mont_to_ints(arr) -> list:
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [pyref.from_mont_limbs(row) for row in a]


def ptr(a: np.ndarray):
    return a.ctypes.data_as(_u64p)


def bptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def fq1(v: int) -> np.ndarray:
    return ints_to_mont([v]).reshape(4)


# ------------------------------------------------------------------ wrappers

def gen_table(seed: int, first: int, count: int) -> np.ndarray:
    out = np.empty((count, 4), dtype=np.uint64)
    lib().orc_gen_table(C.c_uint64(seed), C.c_uint64(first), C.c_size_t(count), ptr(out))
    return out


class OTranscript:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_transcript_new())

    def append(self, data: bytes):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
        lib().orc_transcript_append(self.h, buf, C.c_size_t(len(data)))

    def sample(self) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        lib().orc_transcript_sample(self.h, ptr(out))
        return out

    def peek_digest(self) -> bytes:
        out = np.empty(32, dtype=np.uint8)
        lib().orc_transcript_peek_digest(self.h, bptr(out))
        return out.tobytes()

    def __del__(self):
        try:
            lib().orc_transcript_free(self.h)
        except Exception:
            pass


def sumcheck_prove(table: np.ndarray):
    """-> (claim[4], coeffs[(n,2,4)], challenges[(n,4)]) in Montgomery limbs; raises on error."""
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    n = table.shape[0]
    nv = max(n.bit_length() - 1, 0)
    claim = np.empty(4, dtype=np.uint64)
    coeffs = np.empty((max(nv, 1), 2, 4), dtype=np.uint64)
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    rc = lib().orc_sumcheck_prove(ptr(table), C.c_size_t(n), ptr(claim), ptr(coeffs), ptr(ch))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return claim, coeffs[:nv], ch[:nv]


def sumcheck_verify(table: np.ndarray, claim: np.ndarray, coeffs: np.ndarray) -> bool:
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    n_rounds = coeffs.shape[0] if coeffs.size else 0
    ncoef = coeffs.shape[1] if coeffs.size else 2
    claim = np.ascontiguousarray(claim, dtype=np.uint64)
    c = coeffs if coeffs.size else np.zeros((1, 2, 4), dtype=np.uint64)
    return bool(lib().orc_sumcheck_verify(ptr(table), C.c_size_t(table.shape[0]), ptr(claim), ptr(c),
                                          C.c_size_t(n_rounds), C.c_size_t(ncoef)) == 1)


def _table_ptrs(tables):
    flat = [np.ascontiguousarray(t, dtype=np.uint64).reshape(-1, 4) for t in tables]
    arr = (_u64p * len(flat))(*[ptr(t) for t in flat])
    lens = (C.c_size_t * len(flat))(*[t.shape[0] for t in flat])
    return flat, arr, lens


def sumcheck_prove_sum(sum_poly, claim: np.ndarray, transcript: OTranscript):
    """sum_poly: list of terms, each a list of factor tables.  -> (coeffs[(n,3,4)], challenges[(n,4)])"""
    n_terms, n_factors = len(sum_poly), len(sum_poly[0])
    tables = [t for term in sum_poly for t in term]
    flat, arr, lens = _table_ptrs(tables)
    nv = flat[0].shape[0].bit_length() - 1
    coeffs = np.empty((max(nv, 1), 3, 4), dtype=np.uint64)
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    claim = np.ascontiguousarray(claim, dtype=np.uint64)
    rc = lib().orc_sumcheck_prove_sum(arr, C.c_size_t(n_terms), C.c_size_t(n_factors), lens, ptr(claim),
                                      transcript.h, ptr(coeffs), ptr(ch))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return coeffs[:nv], ch[:nv]


def sumcheck_partial_verify(claim, coeffs, transcript: OTranscript):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    n_rounds, ncoef = coeffs.shape[0], coeffs.shape[1]
    final = np.empty(4, dtype=np.uint64)
    ch = np.empty((max(n_rounds, 1), 4), dtype=np.uint64)
    nch = C.c_size_t(0)
    claim = np.ascontiguousarray(claim, dtype=np.uint64)
    ok = lib().orc_sumcheck_partial_verify(ptr(claim), ptr(coeffs), C.c_size_t(n_rounds), C.c_size_t(ncoef),
                                           transcript.h, ptr(final), ptr(ch), C.byref(nch))
    return bool(ok), final, ch[:nch.value]


def mle_partially_evaluate(table, var, r):
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    out = np.empty((max(table.shape[0] // 2, 1), 4), dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    rc = lib().orc_mle_partially_evaluate(ptr(table), C.c_size_t(table.shape[0]), C.c_size_t(var), ptr(r), ptr(out))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return out[: table.shape[0] // 2]


def mle_minus(table, v):
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(table)
    rc = lib().orc_mle_minus(ptr(table), C.c_size_t(table.shape[0]), ptr(np.ascontiguousarray(v, dtype=np.uint64)), ptr(out))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return out


def mle_quotient_remainder(table, divisor, var):
    """compute_quotient_remainder -> (quotient (len/2, 4), remainder (len/2, 4))"""
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    half = table.shape[0] // 2
    q = np.empty((max(half, 1), 4), dtype=np.uint64)
    r = np.empty((max(half, 1), 4), dtype=np.uint64)
    rc = lib().orc_mle_quotient_remainder(ptr(table), C.c_size_t(table.shape[0]), ptr(np.ascontiguousarray(divisor, dtype=np.uint64)),
                                          C.c_size_t(var), ptr(q), ptr(r))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return q[:half], r[:half]


def mle_evaluate(table, points):
    """points: list of None | (4,) limb arrays"""
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    has = np.array([0 if p is None else 1 for p in points] or [0], dtype=np.uint8)
    vals = np.zeros((max(len(points), 1), 4), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is not None:
            vals[i] = p
    out = np.empty((table.shape[0], 4), dtype=np.uint64)
    ol = C.c_size_t(0)
    rc = lib().orc_mle_evaluate(ptr(table), C.c_size_t(table.shape[0]), bptr(has), ptr(vals), C.c_size_t(len(points)),
                                ptr(out), C.byref(ol))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return out[: ol.value].copy()


def mle_to_bytes(table) -> bytes:
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    out = np.empty(32 * table.shape[0], dtype=np.uint8)
    lib().orc_mle_to_bytes(ptr(table), C.c_size_t(table.shape[0]), bptr(out))
    return out.tobytes()


def mle_sum(table) -> np.ndarray:
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_mle_sum(ptr(table), C.c_size_t(table.shape[0]), ptr(out))
    return out


class OCircuit:
    """layers: list (input side first) of lists of (left, right, op) with op 0=Add 1=Mul."""

    def __init__(self, layers):
        self.layers = layers
        flat = np.array([x for gates in layers for g in gates for x in g] or [0], dtype=np.uint64)
        lens = (C.c_size_t * max(len(layers), 1))(*[len(g) for g in layers])
        self.h = C.c_void_p(lib().orc_circuit_new(ptr(flat), lens, C.c_size_t(len(layers))))
        if not self.h:
            raise ValueError("circuit_new failed")

    def evaluate(self, inputs: np.ndarray):
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
        cap = inputs.shape[0] + sum(max(1 << max(len(g) - 1, 0).bit_length(), 2) for g in self.layers) + 8
        out = np.empty((cap, 4), dtype=np.uint64)
        lens = (C.c_size_t * (len(self.layers) + 1))()
        rc = lib().orc_circuit_evaluate(self.h, ptr(inputs), C.c_size_t(inputs.shape[0]), ptr(out), C.c_size_t(cap), lens)
        if rc:
            raise ValueError(f"oracle error {rc}")
        res, off = [], 0
        for l in lens:
            res.append(out[off: off + l].copy())
            off += l
        return res

    def gate_poly(self, layer_idx: int, op: int) -> np.ndarray:
        rc = C.c_int(0)
        n = lib().orc_circuit_gate_poly(self.h, C.c_size_t(layer_idx), C.c_int(op), None, C.c_size_t(0), C.byref(rc))
        if rc.value:
            raise ValueError(f"oracle error {rc.value}")
        out = np.empty((n, 4), dtype=np.uint64)
        lib().orc_circuit_gate_poly(self.h, C.c_size_t(layer_idx), C.c_int(op), ptr(out), C.c_size_t(n), C.byref(rc))
        return out

    def __del__(self):
        try:
            lib().orc_circuit_free(self.h)
        except Exception:
            pass


class OGkrProof:
    def __init__(self, h):
        self.h = C.c_void_p(h)

    def serialize(self):
        shape = (C.c_size_t * 64)()
        n = lib().orc_gkr_proof_serialize(self.h, None, C.c_size_t(0), shape, C.c_size_t(64))
        out = np.empty(n, dtype=np.uint8)
        lib().orc_gkr_proof_serialize(self.h, bptr(out), C.c_size_t(n), shape, C.c_size_t(64))
        n_proofs = shape[2]
        return out.tobytes(), {"output_len": shape[0], "n_w_evals": shape[1], "n_proofs": n_proofs,
                               "rounds": [shape[3 + i] for i in range(n_proofs)]}

    def tamper(self, layer, rnd):
        lib().orc_gkr_proof_tamper(self.h, C.c_size_t(layer), C.c_size_t(rnd))

    def __del__(self):
        try:
            lib().orc_gkr_proof_free(self.h)
        except Exception:
            pass


def gkr_prove(circuit: OCircuit, transcript: OTranscript, inputs: np.ndarray) -> OGkrProof:
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
    rc = C.c_int(0)
    h = lib().orc_gkr_prove(circuit.h, transcript.h, ptr(inputs), C.c_size_t(inputs.shape[0]), C.byref(rc))
    if rc.value or not h:
        raise ValueError(f"oracle error {rc.value}")
    return OGkrProof(h)


def gkr_verify(inputs, circuit: OCircuit, transcript: OTranscript, proof: OGkrProof) -> bool:
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
    return bool(lib().orc_gkr_verify(ptr(inputs), C.c_size_t(inputs.shape[0]), circuit.h, transcript.h, proof.h))


def sumcheck_prove_fast(table: np.ndarray, absorb: bool = True, threads: int = 0):
    """The optimised OpenMP CPU variant (oracle/fast.c).  -> (claim, coeffs (n,2,4), challenges (n,4))"""
    table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
    n = table.shape[0]
    nv = max(n.bit_length() - 1, 0)
    claim = np.empty(4, dtype=np.uint64)
    coeffs = np.empty((max(nv, 1), 2, 4), dtype=np.uint64)
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    rc = lib().orc_sumcheck_prove_fast(ptr(table), C.c_size_t(n), C.c_int(int(absorb)), C.c_int(threads), ptr(claim), ptr(coeffs), ptr(ch))
    if rc:
        raise ValueError(f"oracle error {rc}")
    return claim, coeffs[:nv], ch[:nv]
