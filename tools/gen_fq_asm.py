#!/usr/bin/env python3
"""Generates zk_amd/csrc/fq_mul_asm.inc: the gfx950 Montgomery multiplication of BN254 Fq on
8 x 32-bit limbs as ONE inline-asm block (kernel 1 of the north star: v_mad_u64_u32 carry chains).

Algorithm: finely integrated product scanning (FIPS).  Column k accumulates
    sum_{i+j=k} a_i*b_j + sum_{i+j=k} m_i*p_j
into a 96-bit accumulator (T1:T0 as one 64-bit VGPR pair fed through v_mad_u64_u32's addend,
T2 collecting the mad's carry-out through v_addc_co_u32): 2 VALU ops per 32x32 product, no
register shuffles inside a column.  m_k = T0 * (-p^-1 mod 2^32) zeroes the column's low word.
Result < 2p is conditionally reduced with one v_sub/v_subb chain + v_cndmask.

Why one block: hipcc pads every `;;#ASMEND` with an s_nop and cannot address the halves of a
64-bit asm operand, so the accumulator lives in three FIXED, clobbered VGPRs and the modulus in
FIXED, clobbered SGPRs (30-operand limit of an asm statement).

Run:  python tools/gen_fq_asm.py > zk_amd/csrc/fq_mul_asm.inc
"""
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
PL = [(P >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
NINV = (-pow(P, -1, 1 << 32)) % (1 << 32)

# fixed temporaries (clobbered): accumulator pair + carry word; modulus + ninv in SGPRs
T01, T0, T1, T2 = "v[2:3]", "v2", "v3", "v4"
SP = [f"s{40 + i}" for i in range(8)]
SN = "s48"
VCLOB = ["v2", "v3", "v4"]
CARRY = ["s[50:51]", "s[52:53]", "s[54:55]"]
SCLOB = [f"s{40 + i}" for i in range(9)] + [f"s{50 + i}" for i in range(6)]


def emit(variant):
    """variant 'vv': b in VGPRs; 'vs': b in SGPRs (wave-uniform multiplier, e.g. the challenge).

    Hazard rule honoured here (hipcc does not pad inside an asm string): on gfx940/gfx950 a VALU
    instruction that READS an SGPR/VCC written by a VALU instruction needs 2 wait states in between
    (LLVM's VALUWriteSGPRVALURead; hipcc itself emits `s_nop 1` between chained v_addc_co).  The
    v_mad_u64_u32 carry-outs therefore go to a ring of three SGPR pairs and each v_addc_co that
    consumes one is issued two instructions later, behind the next products -- no padding needed
    except at column ends and in the final subtract chain."""
    L = []
    A = lambda i: f"%[a{i}]"
    B = lambda i: f"%[b{i}]"
    M = lambda i: f"%[m{i}]"
    for i in range(8):
        L.append(f"s_mov_b32 {SP[i]}, 0x{PL[i]:08x}")
    L.append(f"s_mov_b32 {SN}, 0x{NINV:08x}")
    L += [f"v_mov_b32 {T0}, 0", f"v_mov_b32 {T1}, 0", f"v_mov_b32 {T2}, 0"]

    pending = []  # (carry pair, index in L of the producing mad)
    nmad = [0]

    def addc_oldest():
        pair, at = pending.pop(0)
        gap = len(L) - at - 1  # instructions issued since the mad
        if gap < 2:
            L.append(f"s_nop {1 - gap}")  # s_nop N = N+1 wait states
        L.append(f"v_addc_co_u32_e64 {T2}, vcc, 0, {T2}, {pair}")

    def mac(x, y):
        pair = CARRY[nmad[0] % 3]
        nmad[0] += 1
        L.append(f"v_mad_u64_u32 {T01}, {pair}, {x}, {y}, {T01}")
        pending.append((pair, len(L) - 1))
        if len(pending) == 3:
            addc_oldest()

    def flush():
        while pending:
            addc_oldest()

    def shift():
        flush()
        L.append(f"v_mov_b32 {T0}, {T1}")
        L.append(f"v_mov_b32 {T1}, {T2}")
        L.append(f"v_mov_b32 {T2}, 0")

    for k in range(8):
        for i in range(k + 1):
            mac(A(i), B(k - i))
        for i in range(k):
            mac(M(i), SP[k - i])
        L.append(f"v_mul_lo_u32 {M(k)}, {T0}, {SN}")
        mac(M(k), SP[0])
        shift()
    for k in range(8, 16):
        for i in range(k - 7, 8):
            mac(A(i), B(k - i))
        for i in range(k - 7, 8):
            mac(M(i), SP[k - i])
        # m_{k-8} was last used in column k-1, so its register is free to take result limb k-8
        L.append(f"v_mov_b32 {M(k - 8)}, {T0}")
        if k < 15:
            shift()
    flush()
    # conditional subtraction: a_j <- r_j - p_j (a is dead), select on the final borrow.
    # v_subb reads vcc, so the modulus limb cannot also come from an SGPR (one constant-bus
    # read per VALU op on gfx9): stage p_j through a_j first.  Each v_subb reads the vcc the
    # previous one wrote: 2 wait states apiece.
    for j in range(8):
        L.append(f"v_mov_b32 {A(j)}, {SP[j]}")
    L.append(f"v_sub_co_u32 {A(0)}, vcc, {M(0)}, {A(0)}")
    for j in range(1, 8):
        L.append("s_nop 1")
        L.append(f"v_subb_co_u32 {A(j)}, vcc, {M(j)}, {A(j)}, vcc")
    L.append("s_nop 1")
    for j in range(8):
        L.append(f"v_cndmask_b32 {M(j)}, {A(j)}, {M(j)}, vcc")
    body = "\n".join(f'        "{ins}\\n\\t"' for ins in L)
    outs = ", ".join([f'[m{i}] "=&v"(r.v[{i}])' for i in range(8)] + [f'[a{i}] "+v"(a.v[{i}])' for i in range(8)])
    bc = '"v"' if variant == "vv" else '"s"'
    ins = ", ".join(f'[b{i}] {bc}(b.v[{i}])' for i in range(8))
    clob = ", ".join(f'"{c}"' for c in ["vcc"] + VCLOB + SCLOB)
    name = "fq_mul" if variant == "vv" else "fq_mul_uniform"
    doc = ("b in VGPRs" if variant == "vv" else
           "b wave-uniform (SGPRs): the caller guarantees every lane passes the same b")
    return f"""// r = a*b*R^-1 mod p, fully reduced; inputs < 2p with a*b < p*2^256.  {doc}.
// `a` is passed by value and destroyed (its registers hold r-p at the end).
__device__ __forceinline__ fq {name}(fq a, const fq& b) {{
    fq r;
    asm(
{body}
        : {outs}
        : {ins}
        : {clob});
    return r;
}}
"""


print("// GENERATED by tools/gen_fq_asm.py -- do not edit.  gfx950 (CDNA4) only.")
print(f"// p = 0x{P:064x}, -p^-1 mod 2^32 = 0x{NINV:08x}")
def l32(x):
    return "{" + ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(8)) + "}"


print(f"#define FQ_P_LIMBS {l32(P)}")
print(f"#define FQ_2P_LIMBS {l32(2 * P)}")
print(f"#define FQ_4P_LIMBS {l32(4 * P)}   // 4p < 2^256 < 6p")
print(f"#define FQ_R2_LIMBS {l32((1 << 512) % P)}   // R^2 mod p, R = 2^256")
print(f"#define FQ_R_LIMBS {l32((1 << 256) % P)}    // Montgomery form of 1")
# radix-2^28 view (10 limbs) used by the carry-free multi-product accumulation (fq28_* in fq_dev.hpp)
print("#define FQ_P28_LIMBS {" + ", ".join("0x%07xu" % ((P >> (28 * i)) & 0xFFFFFFF) for i in range(10)) + "}")
print(f"#define FQ_NINV28 0x{(-pow(P, -1, 1 << 28)) % (1 << 28):07x}u   // -p^-1 mod 2^28")
print()
print(emit("vv"))
print(emit("vs"))
