"""zk_amd -- MI355X-native multilinear sumcheck prover (the hot path of webdott/zk).

Host-side mirror of the reference's prover API over the C ABI of libzkmi.so
(include/zkmi.h).  No CPU fallback: without the HIP library and a gfx950 device, compute
calls raise.
"""
from ._lib import LIB_PATH, SYMBOLS, ZkError, lib
from .circuit import Circuit, Gate, Operation
from .context import Context, default_context
from .gkr import GKRProof, GKRProver, GKRVerifier
from .polynomials import MultiLinearPolynomial, ProductPolynomial, SumPolynomial
from .sumcheck import SumCheckProof, SumcheckProver, SumcheckVerifier
from .transcript import Transcript

__all__ = [
    "LIB_PATH", "SYMBOLS", "ZkError", "lib", "Context", "default_context", "MultiLinearPolynomial",
    "ProductPolynomial", "SumPolynomial", "Circuit", "Gate", "Operation", "GKRProof", "GKRProver", "GKRVerifier", "SumCheckProof", "SumcheckProver", "SumcheckVerifier", "Transcript",
]
