"""Loads tests/golden/vectors.json (made by tests/golden/make_golden.py from oracle/pyref.py)."""
import json
import os

import numpy as np

import oracle_lib as O

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors.json")
_cache = None


def vectors():
    global _cache
    if _cache is None:
        with open(_PATH) as f:
            _cache = json.load(f)
    return _cache


def unhex(h: str) -> int:
    return int.from_bytes(bytes.fromhex(h), "little")


def case_table(case) -> np.ndarray:
    """Montgomery-limb table of a sumcheck case (explicit table or seeded)."""
    if "table" in case:
        return O.ints_to_mont(case["table"])
    return O.gen_table(case["seed"], 0, 1 << case["log_n"])


def le32_of(arr) -> list:
    """(…,4) Montgomery limbs -> list of LE32 hex strings (via Python big ints)."""
    return [O.pyref.le32(v).hex() for v in O.mont_to_ints(arr)]
