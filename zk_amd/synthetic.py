"""Synthetic workloads of SURVEY.md 8(d), shared by bench.py and the full-size tests (no reference counterpart:
the reference has no benchmark inputs)."""
from __future__ import annotations

import numpy as np

SEED = 0x5EED0000


def splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gkr_layers(depth: int, log_w: int, seed: int = SEED):
    """every layer W = 2^log_w gates; gate g reads left = 2g mod W,
    right = (2g+1) mod W; op = Mul if splitmix64(seed, layer, g) & 1 else Add.
    Returns one (W, 3) uint64 array (left, right, op) per layer, layers[0] next to the inputs."""
    W = 1 << log_w
    g = np.arange(W, dtype=np.uint64)
    layers = []
    for layer in range(depth):
        a = np.empty((W, 3), dtype=np.uint64)
        a[:, 0] = (2 * g) % np.uint64(W)
        a[:, 1] = (2 * g + np.uint64(1)) % np.uint64(W)
        a[:, 2] = splitmix64(np.uint64(seed) + np.uint64(layer) * np.uint64(1 << 32) + g) & np.uint64(1)
        layers.append(a)
    return layers
