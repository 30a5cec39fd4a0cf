"""fiat_shamir::Transcript (fiat_shamir/src/transcript.rs:6-42): one Keccak-256 stream on the host."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, u64p, u8p


class Transcript:
    def __init__(self, _handle=None):        # Transcript::new, :12-17
        self.h = C.c_void_p(_handle if _handle is not None else lib().zk_transcript_new())
        if not self.h:
            raise MemoryError("zk_transcript_new")

    def clone(self) -> "Transcript":
        return Transcript(_handle=lib().zk_transcript_clone(self.h))

    def append(self, data: bytes):           # :20-22
        check(lib().zk_transcript_append(self.h, bytes(data), len(data)))

    def append_n(self, parts):               # :24-26
        for p in parts:
            self.append(p)

    def sample_challenge(self) -> np.ndarray:  # :28-36 -> 4 x u64 Montgomery limbs
        out = np.empty(4, dtype=np.uint64)
        check(lib().zk_transcript_sample_challenge(self.h, out.ctypes.data_as(u64p)))
        return out

    def sample_n_challenges(self, n: int) -> np.ndarray:  # :39-41
        out = np.empty((max(n, 1), 4), dtype=np.uint64)
        check(lib().zk_transcript_sample_n_challenges(self.h, n, out.ctypes.data_as(u64p)))
        return out[:n]

    def peek_digest(self) -> bytes:
        out = np.empty(32, dtype=np.uint8)
        check(lib().zk_transcript_peek_digest(self.h, out.ctypes.data_as(u8p)))
        return out.tobytes()

    def __del__(self):
        try:
            if self.h:
                lib().zk_transcript_free(self.h)
                self.h = None
        except Exception:
            pass
