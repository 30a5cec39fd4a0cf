"""Pins the CPU oracle's field and hash layers (oracle/fq.c, oracle/keccak.c, oracle/pyref.py)
against public definitions: BN254 Fq constants, Python big-int arithmetic, Keccak-256 known
answers and hashlib's SHA3-256 (same permutation, different pad byte).  SURVEY.md 8c (ii),(iii)."""
import ctypes as C
import hashlib
import random

import numpy as np

import oracle_lib as O
from oracle_lib import P, lib, ptr, pyref


def _bin(fn, a, b):
    out = np.empty(4, dtype=np.uint64)
    getattr(lib(), fn)(ptr(O.fq1(a)), ptr(O.fq1(b)), ptr(out))
    return O.mont_to_ints(out)[0]


def test_constants_survey_a0():
    R = 1 << 256
    assert P.bit_length() == 254
    assert R % P == 0x0E0A77C19A07DF2F666EA36F7879462C0A78EB28F5C70B3DD35D438DC58F0D9D
    assert R * R % P == 0x06D89F71CAB8351F47AB1EFF0A417FF6B5E71911D44501FBF32CFC5B538AFA89
    assert (-pow(P, -1, 1 << 64)) % (1 << 64) == 0x87D20782E4866389
    assert pow(2, -1, P) == 0x183227397098D014DC2822DB40C0AC2ECBC0B548B438E5469E10460B6C3E7EA4
    one = np.empty(4, dtype=np.uint64)
    lib().orc_fq_from_u64(C.c_uint64(1), ptr(one))
    assert sum(int(x) << (64 * i) for i, x in enumerate(one)) == R % P


def test_field_ops_vs_bigint():
    rng = random.Random(1234)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (1 << 253), (1 << 128) - 1]
    vals = edge + [rng.randrange(P) for _ in range(200)]
    for _ in range(400):
        a, b = rng.choice(vals), rng.choice(vals)
        assert _bin("orc_fq_add", a, b) == (a + b) % P
        assert _bin("orc_fq_sub", a, b) == (a - b) % P
        assert _bin("orc_fq_mul", a, b) == (a * b) % P
        if b:
            assert _bin("orc_fq_div", a, b) == a * pow(b, -1, P) % P
    assert _bin("orc_fq_add", P - 1, 1) == 0


def test_montgomery_roundtrip_and_le32():
    rng = random.Random(7)
    for v in [0, 1, P - 1] + [rng.randrange(P) for _ in range(50)]:
        limbs = np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
        m = np.empty(4, dtype=np.uint64)
        lib().orc_fq_from_canonical(ptr(limbs), ptr(m))
        assert list(m) == pyref.to_mont_limbs(v)
        back = np.empty(4, dtype=np.uint64)
        lib().orc_fq_to_canonical(ptr(m), ptr(back))
        assert list(back) == list(limbs)
        b = np.empty(32, dtype=np.uint8)
        lib().orc_fq_to_le32(ptr(m), O.bptr(b))
        assert b.tobytes() == v.to_bytes(32, "little")


def test_from_le_bytes_mod_order():
    rng = random.Random(99)
    cases = [bytes(32), b"\xff" * 32, P.to_bytes(32, "little"), (P - 1).to_bytes(32, "little")]
    cases += [bytes(rng.randrange(256) for _ in range(n)) for n in (1, 7, 31, 32, 32, 32, 33, 64)]
    for data in cases:
        out = np.empty(4, dtype=np.uint64)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        lib().orc_fq_from_le_bytes_mod_order(buf, C.c_size_t(len(data)), ptr(out))
        assert O.mont_to_ints(out)[0] == int.from_bytes(data, "little") % P


def _ck(data: bytes) -> bytes:
    out = np.empty(32, dtype=np.uint8)
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    lib().orc_keccak256(buf, C.c_size_t(len(data)), O.bptr(out))
    return out.tobytes()


def test_keccak256_public_known_answers():
    kats = {
        b"": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470",
        b"abc": "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45",
    }
    for msg, hexd in kats.items():
        assert _ck(msg).hex() == hexd
        assert pyref.keccak256(msg).hex() == hexd


def test_keccak_permutation_vs_hashlib_and_c_vs_python():
    rng = random.Random(5)
    for n in (0, 1, 55, 135, 136, 137, 271, 272, 273, 1000):
        msg = bytes(rng.randrange(256) for _ in range(n))
        # same sponge with NIST's pad byte must equal hashlib's SHA3-256
        assert pyref.keccak256(msg, pad_byte=0x06) == hashlib.sha3_256(msg).digest()
        assert pyref.keccak256(msg) != hashlib.sha3_256(msg).digest()
        assert _ck(msg) == pyref.keccak256(msg)


def test_keccak_incremental_equals_oneshot():
    rng = random.Random(6)
    msg = bytes(rng.randrange(256) for _ in range(700))
    t = O.OTranscript()
    i = 0
    for step in (1, 2, 133, 136, 1, 200, 227):
        t.append(msg[i:i + step])
        i += step
    assert i == len(msg)
    assert t.peek_digest() == pyref.keccak256(msg)


def test_transcript_matches_python_and_reference_tests():
    # transcript.rs:122-137: two transcripts fed "hello","world" agree (determinism);
    # here additionally C == Python on the resulting challenge.
    t1, t2, tp = O.OTranscript(), O.OTranscript(), pyref.Transcript()
    for t in (t1, t2, tp):
        t.append(b"hello")
        t.append(b"world")
    c1, c2 = t1.sample(), t2.sample()
    assert list(c1) == list(c2)
    assert O.mont_to_ints(c1)[0] == tp.sample_challenge()
    # the digest is re-absorbed: the second challenge differs and still matches
    assert O.mont_to_ints(t1.sample())[0] == tp.sample_challenge()
    assert O.mont_to_ints(c1)[0] == int.from_bytes(pyref.keccak256(b"helloworld"), "little") % P


def test_gen_table_c_vs_python():
    a = O.gen_table(0x5EED0000, 5, 40)
    assert O.mont_to_ints(a) == pyref.gen_table(0x5EED0000, 5, 40)
