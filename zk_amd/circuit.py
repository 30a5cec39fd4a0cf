"""Host-side mirror of the reference's `arithmetic_circuit` crate (arithmetic_circuit/src/circuit.rs,
gate.rs): a layered circuit of fan-in-2 add/mul gates.  The wiring lives on the device as gate lists
plus CSR indices (zk_circuit in include/zkmi.h); the dense add_i / mul_i tables of the reference are
only materialised on request (get_add_i / get_mul_i), never by the GKR prover."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from enum import IntEnum

import numpy as np

from ._lib import ZkError, check, lib, u64p
from .context import Context, default_context
from .polynomials import MultiLinearPolynomial


class Operation(IntEnum):      # gate.rs:1-5
    Add = 0
    Mul = 1


@dataclass(frozen=True)
class Gate:                    # gate.rs:7-22
    left: int
    right: int
    operation: Operation


class Circuit:
    def __init__(self, layers, ctx: Context | None = None):
        """Circuit::new (circuit.rs:15-20); layers[0] is the layer next to the inputs."""
        arrays = [np.array([(g.left, g.right, int(g.operation)) for g in gs], dtype=np.uint64).reshape(-1, 3) for gs in layers]
        self._create(arrays, ctx)

    @classmethod
    def from_arrays(cls, layers, ctx: Context | None = None):
        """layers: per layer an (n_gates, 3) uint64 array of (left, right, op) rows, op 0 = Add, 1 = Mul."""
        self = cls.__new__(cls)
        self._create([np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 3) for a in layers], ctx)
        return self

    def _create(self, arrays, ctx):
        self.ctx = ctx or default_context()
        flat = np.concatenate(arrays).reshape(-1) if arrays else np.zeros(3, dtype=np.uint64)
        flat = np.ascontiguousarray(flat, dtype=np.uint64)
        lens = (C.c_size_t * max(len(arrays), 1))(*[a.shape[0] for a in arrays])
        h = C.c_void_p()
        check(lib().zk_circuit_create(self.ctx.h, flat.ctypes.data_as(u64p), lens, len(arrays), C.byref(h)), self.ctx.h)
        self.h = h

    def get_layer_count(self) -> int:                      # :149-151
        return lib().zk_circuit_layer_count(self.h)

    def shape(self):
        """(len(W_0), [sumcheck rounds of layer 0 (output), 1, ...])"""
        n = self.get_layer_count()
        out_len = C.c_size_t()
        rounds = (C.c_size_t * max(n, 1))()
        check(lib().zk_circuit_shape(self.h, C.byref(out_len), rounds))
        return out_len.value, [rounds[i] for i in range(n)]

    def evaluate_at_input(self, inputs):                   # :37-59 -> list of MLEs, inputs first
        if not isinstance(inputs, MultiLinearPolynomial):
            inputs = MultiLinearPolynomial(inputs, ctx=self.ctx)
        n = self.get_layer_count()
        arr = (C.c_void_p * (n + 1))()
        check(lib().zk_circuit_evaluate(self.h, inputs.h, arr), self.ctx.h)
        return [MultiLinearPolynomial(ctx=self.ctx, _handle=C.c_void_p(arr[i])) for i in range(n + 1)]

    @staticmethod
    def get_w_i(layer_idx: int, layer_evaluations):        # :126-138, counted from the output
        if layer_idx >= len(layer_evaluations):
            raise ZkError(4)
        return layer_evaluations[len(layer_evaluations) - layer_idx - 1].clone()

    def _gate_poly(self, layer_idx: int, op: int):
        h = C.c_void_p()
        check(lib().zk_circuit_gate_poly(self.h, layer_idx, op, C.byref(h)), self.ctx.h)
        return MultiLinearPolynomial(ctx=self.ctx, _handle=h)

    def get_add_i(self, layer_idx: int):                   # :140-142 (dense)
        return self._gate_poly(layer_idx, 0)

    def get_mul_i(self, layer_idx: int):                   # :144-146 (dense)
        return self._gate_poly(layer_idx, 1)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().zk_circuit_free(self.h)
                self.h = None
        except Exception:
            pass
