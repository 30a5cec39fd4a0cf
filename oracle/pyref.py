"""oracle/pyref.py -- TEST INFRASTRUCTURE ONLY (second, independent CPU oracle).

A pure-Python big-int restatement of the same reference path as the C oracle
(oracle/*.c), written separately so the two can be checked against each other
(SURVEY.md 8c (iv), H4).  Field elements are plain canonical ints mod p (no
Montgomery form anywhere), Keccak-256 is written out in Python.  Small cases only.

Reference files followed (paths under /root/reference):
  fiat_shamir/src/transcript.rs:11-42
  polynomials/src/multilinear_polynomial/evaluation_form.rs:45-80,82-112,163-258
  polynomials/src/product_polynomial.rs:31-73, sum_polynomial.rs:24-81
  polynomials/src/univariate_polynomial/dense_coefficient_form.rs:20-88,94-191
  sumcheck/src/prover.rs:23-190, sumcheck/src/verifier.rs:15-94
  arithmetic_circuit/src/circuit.rs:37-146
  gkr/src/prover.rs:24-154, gkr/src/utils.rs:5-65, gkr/src/verifier.rs:21-119

Parity status: the reference holds no golden digests / challenges / proof bytes
(SURVEY.md 8c), so byte-level parity with the Rust binary is "parity unpinned";
what is pinned is (a) every small-integer vector in the reference's own tests,
(b) the public Keccak-256 known answers, (c) the public BN254 modulus.
"""
from __future__ import annotations

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 1 << 256
R_INV = pow(R, -1, P)

# ----------------------------------------------------------------------------- Keccak-256

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_M64 = (1 << 64) - 1


def _rot_offsets():
    # generated from the specification: (x,y) walks (1,0) -> (y, 2x+3y), offset (t+1)(t+2)/2
    r = [[0] * 5 for _ in range(5)]
    x, y = 1, 0
    for t in range(24):
        r[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return r


_ROT = _rot_offsets()


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M64 if n else v


def keccak_f1600(a):
    """a: 5x5 list a[x][y] of 64-bit ints; returns the permuted state."""
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y] & _M64) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return a


def keccak256(data: bytes, pad_byte: int = 0x01) -> bytes:
    """Keccak-256 (pad_byte 0x01) -- with pad_byte 0x06 this is NIST SHA3-256, which lets
    the permutation be checked against hashlib.sha3_256."""
    rate = 136
    msg = bytearray(data)
    msg.append(pad_byte)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] ^= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        a = keccak_f1600(a)
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


# ----------------------------------------------------------------------------- transcript

class Keccak256:
    """Incremental Keccak-256 sponge with a non-consuming digest (sha3::Keccak256 +
    `clone().finalize()`).  Equivalent to keccak256(all bytes so far)."""

    RATE = 136

    def __init__(self):
        self.a = [[0] * 5 for _ in range(5)]
        self.buf = bytearray()

    def _absorb(self, a, block):
        a = [row[:] for row in a]
        for i in range(self.RATE // 8):
            a[i % 5][i // 5] ^= int.from_bytes(block[8 * i: 8 * i + 8], "little")
        return keccak_f1600(a)

    def update(self, data: bytes):
        self.buf += data
        while len(self.buf) >= self.RATE:
            self.a = self._absorb(self.a, self.buf[: self.RATE])
            del self.buf[: self.RATE]

    def digest(self) -> bytes:
        blk = bytearray(self.buf) + bytearray(self.RATE - len(self.buf))
        blk[len(self.buf)] ^= 0x01
        blk[-1] ^= 0x80
        a = self._absorb(self.a, blk)
        return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class Transcript:
    """fiat_shamir/src/transcript.rs:6-42 -- one ever-growing Keccak-256 stream."""

    def __init__(self):
        self.hasher = Keccak256()

    def append(self, data: bytes):
        self.hasher.update(data)

    def append_n(self, parts):
        for p in parts:
            self.append(p)

    def sample_challenge(self) -> int:
        h = self.hasher.digest()                # hasher.clone().finalize()
        self.append(h)                          # digest re-absorbed
        return int.from_bytes(h, "little") % P  # from_le_bytes_mod_order

    def sample_n_challenges(self, n):
        return [self.sample_challenge() for _ in range(n)]


def le32(v: int) -> bytes:
    return (v % P).to_bytes(32, "little")


# ----------------------------------------------------------------------------- MLE

def _check_pow2(n):
    if n == 0 or n & (n - 1):
        raise ValueError("Invalid Multilinear Polynomial: evaluation points length is not a power of two")


def num_vars(table):
    return len(table).bit_length() - 1


def y1_y2_indexes(n_entries, var):
    nv = n_entries.bit_length() - 1
    power = nv - 1 - var
    target = 1 << power
    out, y1 = [], 0
    for _ in range(n_entries // 2):
        y2 = y1 | (1 << power)
        out.append((y1, y2))
        y1 = y2 + 1 if (y1 + 1) % target == 0 else y1 + 1
    return out


def partially_evaluate(table, var, r):
    _check_pow2(len(table))
    return [(table[a] + (table[b] - table[a]) * r) % P for a, b in y1_y2_indexes(len(table), var)]


def minus(table, v):
    """minus (evaluation_form.rs:260-272)"""
    return [(x - v) % P for x in table]


def compute_quotient_remainder(table, divisor, var):
    """compute_quotient_remainder (evaluation_form.rs:280-301) -> (quotient list, remainder table)"""
    _check_pow2(len(table))
    quotient = [(table[b] - table[a]) % P for a, b in y1_y2_indexes(len(table), var)]
    return quotient, partially_evaluate(table, var, divisor)


def evaluate(table, points):
    _check_pow2(len(table))
    if len(points) != num_vars(table):
        raise ValueError("points length does not match number of variables")
    acc, done = list(table), 0
    for idx, pt in enumerate(points):
        if pt is not None:
            acc = partially_evaluate(acc, idx - done, pt)
            done += 1
    return acc


def to_bytes(table):
    return b"".join(le32(v) for v in table)


def operate_w(wb, wc, mul):
    lb, lc = len(wb), len(wc)
    out = []
    for i in range(lb * lc):
        ib, ic = i // lb, i % lc
        out.append((wb[ib] * wc[ic]) % P if mul else (wb[ib] + wc[ic]) % P)
    return out


def w_add(wb, wc):
    return operate_w(wb, wc, False)


def w_mul(self_, other):
    return operate_w(other, self_, True)


# ----------------------------------------------------------------------------- univariate

def uni_evaluate(coeffs, x):
    res, run = 0, 1
    for c in coeffs:
        res = (res + c * run) % P
        run = (run * x) % P
    return res


def _uni_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % P
    return out


def interpolate(xs, ys):
    n = len(xs)
    res = [0] * n
    for i in range(n):
        denom, numer = 1, [1]
        for j in range(n):
            if i == j:
                continue
            denom = denom * (xs[i] - xs[j]) % P
            numer = _uni_mul(numer, [(-xs[j]) % P, 1])
        k = ys[i] * pow(denom, -1, P) % P
        scaled = [c * k % P for c in numer]
        m = min(len(res), len(scaled))          # the reference's min-length _add
        res = [(res[t] + scaled[t]) % P for t in range(m)]
    return res


# ----------------------------------------------------------------------------- sum of products

def sum_partial_evaluate(sum_poly, points):
    return [[evaluate(t, points) for t in prod] for prod in sum_poly]


def sum_reduce(sum_poly):
    n = len(sum_poly[0][0])
    out = []
    for idx in range(n):
        s = 0
        for prod in sum_poly:
            run = 1
            for t in prod:
                run = run * t[idx] % P
            s = (s + run) % P
        out.append(s)
    return out


# ----------------------------------------------------------------------------- sumcheck

def generate_round_polys(poly, transcript, is_sum):
    """sumcheck/src/prover.rs:45-155.  poly: table (list) or sum_poly (list of list of tables)."""
    nv = num_vars(poly[0][0]) if is_sum else num_vars(poly)
    round_polys, challenges = [], []
    cur = [[list(t) for t in prod] for prod in poly] if is_sum else list(poly)
    for _ in range(nv):
        if is_sum:
            degree = len(cur)
            cv = num_vars(cur[0][0])
            ev = []
            for i in range(degree + 1):
                pts = [None] * cv
                pts[0] = i % 256
                ev.append(sum(sum_reduce(sum_partial_evaluate(cur, pts))) % P)
            claimed = (ev[0] + ev[1]) % P
            g = interpolate([0, 1, 2], ev)
        else:
            half = len(cur) // 2
            e0, e1 = sum(cur[:half]) % P, sum(cur[half:]) % P
            claimed = (e0 + e1) % P
            g = interpolate([0, 1], [e0, e1])
        transcript.append_n([le32(claimed), b"".join(le32(c) for c in g)])
        r = transcript.sample_challenge()
        challenges.append(r)
        if is_sum:
            pts = [r] + [None] * (num_vars(cur[0][0]) - 1)
            cur = sum_partial_evaluate(cur, pts)
        else:
            pts = [r] + [None] * (num_vars(cur) - 1)
            cur = evaluate(cur, pts)
        round_polys.append(g)
    return round_polys, challenges


def generate_sumcheck_proof(table):
    _check_pow2(len(table))
    tr = Transcript()
    tr.append(to_bytes(table))
    round_polys, challenges = generate_round_polys(table, tr, False)
    return {"initial_claim_sum": sum(table) % P, "round_polys": round_polys, "challenges": challenges}


def generate_proof_for_partial_verify(claim, sum_poly, transcript):
    round_polys, pts = generate_round_polys(sum_poly, transcript, True)
    return {"initial_claim_sum": claim % P, "round_polys": round_polys}, pts


def partial_verify(proof, transcript):
    challenges, cur = [], proof["initial_claim_sum"]
    for g in proof["round_polys"]:
        if (uni_evaluate(g, 0) + uni_evaluate(g, 1)) % P != cur:
            return False, cur, challenges
        transcript.append_n([le32(cur), b"".join(le32(c) for c in g)])
        r = transcript.sample_challenge()
        challenges.append(r)
        cur = uni_evaluate(g, r)
    return True, cur, challenges


def verify_proof(table, proof):
    tr = Transcript()
    tr.append(to_bytes(table))
    if sum(table) % P != proof["initial_claim_sum"]:
        return False
    ok, final_claim, challenges = partial_verify(proof, tr)
    if not ok or not proof["round_polys"]:
        return False
    return evaluate(table, challenges)[0] == final_claim


# ----------------------------------------------------------------------------- circuit / GKR

ADD, MUL = 0, 1


def _next_pow2(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def circuit_evaluate(layers, inputs):
    _check_pow2(len(inputs))
    evals, running = [list(inputs)], list(inputs)
    for gates in layers:
        nxt = [0] * max(_next_pow2(len(gates)), 2)
        for idx, (l, r, op) in enumerate(gates):
            nxt[idx] = (running[l] + running[r]) % P if op == ADD else (running[l] * running[r]) % P
        evals.append(nxt)
        running = nxt
    return evals


def gate_poly(layers, layer_idx, op):
    if layer_idx >= len(layers):
        raise IndexError("layer index out of bounds")
    gates = layers[len(layers) - layer_idx - 1]
    out_len = _next_pow2(len(gates))
    if out_len == 1:
        out_len = 2
    ibl = _next_pow2(max(max(l + 1, r + 1) for l, r, _ in gates)).bit_length() - 1
    ev = [0] * (out_len << (2 * ibl))
    for idx, (l, r, o) in enumerate(gates):
        if o == op:
            ev[(((idx << ibl) | l) << ibl) | r] = 1
    return ev


def get_w_i(layer_idx, evals):
    return list(evals[len(evals) - layer_idx - 1])


def _folded(alpha, beta, poly, rb, rc):
    nv = num_vars(poly)
    pb = list(rb) + [None] * (nv - len(rb))
    pc = list(rc[:len(rb)]) + [None] * (nv - len(rb))
    eb, ec = evaluate(poly, pb), evaluate(poly, pc)
    return [(alpha * x + beta * y) % P for x, y in zip(eb, ec)]


def gkr_generate_proof(layers, transcript, inputs):
    evals = circuit_evaluate(layers, inputs)
    running = get_w_i(0, evals)
    transcript.append(to_bytes(running))
    rv = transcript.sample_n_challenges(num_vars(running))
    w_evals, proofs = [], []
    for layer_idx in range(len(layers)):
        mul, add = gate_poly(layers, layer_idx, MUL), gate_poly(layers, layer_idx, ADD)
        if layer_idx == 0:
            nv = num_vars(mul)
            pts = list(rv) + [None] * (nv - len(rv))
            new_mul, new_add = evaluate(mul, pts), evaluate(add, pts)
            claim = evaluate(running, rv)[0]
        else:
            rb, rc = rv[:len(rv) // 2], rv[len(rv) // 2:]
            wb, wc = evaluate(running, rb)[0], evaluate(running, rc)[0]
            transcript.append_n([le32(wb), le32(wc)])
            alpha, beta = transcript.sample_challenge(), transcript.sample_challenge()
            new_mul, new_add = _folded(alpha, beta, mul, rb, rc), _folded(alpha, beta, add, rb, rc)
            w_evals.append((wb, wc))
            claim = (alpha * wb + beta * wc) % P
        nxt = get_w_i(layer_idx + 1, evals)
        f = [[new_mul, w_mul(nxt, nxt)], [new_add, w_add(nxt, nxt)]]
        proof, pts = generate_proof_for_partial_verify(claim, f, transcript)
        rv, running = pts, nxt
        proofs.append(proof)
    return {"output_poly": get_w_i(0, evals), "w_polys_evals": w_evals, "sumcheck_proofs": proofs}


def gkr_verify_proof(inputs, layers, transcript, proof):
    out = proof["output_poly"]
    transcript.append(to_bytes(out))
    rv = transcript.sample_n_challenges(num_vars(out))
    for layer_idx in range(len(layers)):
        mul, add = gate_poly(layers, layer_idx, MUL), gate_poly(layers, layer_idx, ADD)
        if layer_idx == 0:
            nv = num_vars(mul)
            pts = list(rv) + [None] * (nv - len(rv))
            new_mul, new_add = evaluate(mul, pts), evaluate(add, pts)
        else:
            alpha, beta = transcript.sample_challenge(), transcript.sample_challenge()
            rb, rc = rv[:len(rv) // 2], rv[len(rv) // 2:]
            new_mul, new_add = _folded(alpha, beta, mul, rb, rc), _folded(alpha, beta, add, rb, rc)
        ok, final_claim, nxt_vals = partial_verify(proof["sumcheck_proofs"][layer_idx], transcript)
        add_eval, mul_eval = evaluate(new_add, nxt_vals)[0], evaluate(new_mul, nxt_vals)[0]
        if layer_idx + 1 == len(layers):
            h = len(nxt_vals) // 2
            wb, wc = evaluate(list(inputs), nxt_vals[:h])[0], evaluate(list(inputs), nxt_vals[h:])[0]
        else:
            wb, wc = proof["w_polys_evals"][layer_idx]
        transcript.append_n([le32(wb), le32(wc)])
        fbc = (add_eval * (wb + wc) + mul_eval * (wb * wc)) % P
        if not ok or fbc != final_claim:
            return False
        rv = nxt_vals
    return True


def gkr_proof_bytes(proof) -> bytes:
    """struct order of gkr_protocol.rs:8-12, LE32 per element."""
    out = bytearray(to_bytes(proof["output_poly"]))
    for wb, wc in proof["w_polys_evals"]:
        out += le32(wb) + le32(wc)
    for sp in proof["sumcheck_proofs"]:
        out += le32(sp["initial_claim_sum"])
        for g in sp["round_polys"]:
            for c in g:
                out += le32(c)
    return bytes(out)


# ----------------------------------------------------------------------------- synthetic tables

def splitmix64_at(seed, i):
    z = (seed + i * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def gen_table(seed, first, count):
    """canonical values; entry j uses draws 4j+1..4j+4 of the stream."""
    out = []
    for j in range(first, first + count):
        v = 0
        for w in range(4):
            v |= splitmix64_at(seed, 4 * j + w + 1) << (64 * w)
        out.append(v % P)
    return out


def to_mont_limbs(v):
    m = v * R % P
    return [(m >> (64 * i)) & _M64 for i in range(4)]


def from_mont_limbs(limbs):
    m = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return m * R_INV % P
