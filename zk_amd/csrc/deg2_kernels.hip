// deg2_kernels.hip -- gfx950 kernels of the degree-2 (SumPolynomial) branch of the sumcheck prover: Sum[Prod[A,B], Prod[C,D]]
// (sumcheck/src/prover.rs:89-107, :131-139), one, two or three rounds per device pass.  Data layout and conventions: kernels.hip.
#include "kernels_common.hpp"

namespace zkk {

// ---- degree-2 round reduction for Sum[Prod[A,B], Prod[C,D]] (prover.rs:89-107) -----------
// For X in {0,1,2}: ev[X] = sum_j  A(X)_j*B(X)_j + C(X)_j*D(X)_j,  T(X)_j = lo_j + X*(hi_j - lo_j),
// lo = T[j], hi = T[j+half].  X = 2 needs only additions: 2*hi - lo.

__device__ __forceinline__ void eval3_accumulate(fq (&acc)[3], const fq (&lo)[4], const fq (&hi)[4]) {
    fq x2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x2[k] = fq_sub(fq_dbl(hi[k]), lo[k]);
    acc[0] = fq_add(acc[0], fq_add(fq_mul(lo[0], lo[1]), fq_mul(lo[2], lo[3])));
    acc[1] = fq_add(acc[1], fq_add(fq_mul(hi[0], hi[1]), fq_mul(hi[2], hi[3])));
    acc[2] = fq_add(acc[2], fq_add(fq_mul(x2[0], x2[1]), fq_mul(x2[2], x2[3])));
}

__global__ void __launch_bounds__(kThreads) k_eval3(Tab4 T, size_t half, fq* __restrict__ partials) {
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < half; i += stride) {
        fq lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = fq_load_stream(T.t[k] + i);
            hi[k] = fq_load_stream(T.t[k] + half + i);
        }
        eval3_accumulate(acc, lo, hi);
    }
    block_sum_store<3>(acc, partials);
}

// fused: fold all four factor tables at r (prover.rs:131-139) and evaluate the NEXT round's
// three points from the folded values while they are still in registers.  src tables have 4*q
// entries, dst tables 2*q.
__global__ void __launch_bounds__(kThreads) k_fold4_eval3(Tab4 S, Tab4w D, size_t q, fq r, fq* __restrict__ partials) {
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < q; i += stride) {
        fq lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fq a0 = fq_load_stream(S.t[k] + i), a1 = fq_load_stream(S.t[k] + 2 * q + i);
            lo[k] = fq_fold(a0, a1, r);
            fq_store(D.t[k] + i, lo[k]);
            fq b0 = fq_load_stream(S.t[k] + q + i), b1 = fq_load_stream(S.t[k] + 3 * q + i);
            hi[k] = fq_fold(b0, b1, r);
            fq_store(D.t[k] + q + i, hi[k]);
        }
        eval3_accumulate(acc, lo, hi);
    }
    block_sum_store<3>(acc, partials);
}

// fold of four tables without the evaluation (last round / API use)
__global__ void __launch_bounds__(kThreads) k_fold4(Tab4 S, Tab4w D, size_t half, fq r) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < half; i += stride) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fq_store(D.t[k] + i, fq_fold(fq_load_stream(S.t[k] + i), fq_load_stream(S.t[k] + half + i), r));
    }
}

// ---- degree-2 branch, TWO rounds per pass ---------------------------------------------------------
// Index = (b1, b2, j): b1, b2 the two top variables, j < q = len/4.  For one product term P = F*G and
// Y in {0,1,2} let f0 = F(0,Y,j), f1 = F(1,Y,j) (F(b1,Y,j) = F[b1,0,j] + Y (F[b1,1,j] - F[b1,0,j])), df = f1 - f0.
// After binding b1 := r the term is (f0 + r df)(g0 + r dg) = f0 g0 + r (f1 g1 - f0 g0 - df dg) + r^2 df dg, so
// the nine sums  S[3Y + 0] = sum f0 g0,  S[3Y + 1] = sum f1 g1,  S[3Y + 2] = sum df dg  (over j and both terms)
// give BOTH the current round's evaluations (prover.rs:93-100: ev(X) = sum_{Y in {0,1}} [S0 + X (S1-S0-S2) + X^2 S2])
// and, once r is known, the next round's (ev'(Y) = S0 + r (S1-S0-S2) + r^2 S2): one sweep and one host round
// trip per TWO rounds, and the fold that follows binds both variables at once.  Same field values as the
// reference's per-round computation (exact arithmetic).
// Work split: the fold and the sums are separate sweeps, each spread over as many lanes as the data allows --
// one lane per (table, output) in the fold, one per (j, term[, Y]) in the sums -- because most calls (GKR layers)
// run on tables of <= 2^20 entries where the serial chain per lane, not bandwidth, sets the time.

// the three products of one term at one Y from its eight values x[factor][b1][b2]
template <int Y>
__device__ __forceinline__ void sum9_products(fq* __restrict__ acc3, const fq (&x)[2][2][2]) {
    fq v0[2], v1[2], d[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        if (Y == 0) { v0[f] = x[f][0][0]; v1[f] = x[f][1][0]; }
        else if (Y == 1) { v0[f] = x[f][0][1]; v1[f] = x[f][1][1]; }
        else { v0[f] = fq_sub(fq_dbl(x[f][0][1]), x[f][0][0]); v1[f] = fq_sub(fq_dbl(x[f][1][1]), x[f][1][0]); }
        d[f] = fq_sub(v1[f], v0[f]);
    }
    acc3[0] = fq_add(acc3[0], fq_mul(v0[0], v0[1]));
    acc3[1] = fq_add(acc3[1], fq_mul(v1[0], v1[1]));
    acc3[2] = fq_add(acc3[2], fq_mul(d[0], d[1]));
}

// YSPLIT = false: gridDim.y = 2 (term); a lane forms all nine products of its term (tables read once): large tables.
//   partials[(term * gridDim.x + bx) * 9 + 3Y + t]
// YSPLIT = true:  gridDim.y = 6 (term * 3 + Y); a lane forms the three products of one Y: small tables, 3x the lanes.
//   partials[((term * 3 + Y) * gridDim.x + bx) * 3 + t]
template <bool YSPLIT>
__global__ void __launch_bounds__(kThreads) k_sum9(Tab4 T, size_t q, fq* __restrict__ partials) {
    constexpr int NS = YSPLIT ? 3 : 9;
    const int term = YSPLIT ? blockIdx.y / 3 : blockIdx.y;
    const int Ysel = YSPLIT ? blockIdx.y % 3 : -1;
    const fq* __restrict__ F = T.t[2 * term];
    const fq* __restrict__ G = T.t[2 * term + 1];
    fq acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = fq_zero();
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t j = (size_t)blockIdx.x * kThreads + threadIdx.x; j < q; j += stride) {
        fq x[2][2][2];
        if (!YSPLIT || Ysel != 1) { // b2 = 0 values: needed by Y = 0 and Y = 2
            x[0][0][0] = fq_load_stream(F + j);
            x[0][1][0] = fq_load_stream(F + 2 * q + j);
            x[1][0][0] = fq_load_stream(G + j);
            x[1][1][0] = fq_load_stream(G + 2 * q + j);
        }
        if (!YSPLIT || Ysel != 0) { // b2 = 1 values: needed by Y = 1 and Y = 2
            x[0][0][1] = fq_load_stream(F + q + j);
            x[0][1][1] = fq_load_stream(F + 3 * q + j);
            x[1][0][1] = fq_load_stream(G + q + j);
            x[1][1][1] = fq_load_stream(G + 3 * q + j);
        }
        if (YSPLIT) {
            if (Ysel == 0) sum9_products<0>(acc, x);
            else if (Ysel == 1) sum9_products<1>(acc, x);
            else sum9_products<2>(acc, x);
        } else {
            sum9_products<0>(acc + 0, x);
            sum9_products<1>(acc + (YSPLIT ? 0 : 3), x);
            sum9_products<2>(acc + (YSPLIT ? 0 : 6), x);
        }
    }
    block_sum_store<NS>(acc, partials + (size_t)blockIdx.y * gridDim.x * NS);
}

// Large tables: the sums are bound by the 18 field products per quarter-index, not by HBM.  Here the products are
// accumulated UNREDUCED in radix 2^28 (fq_dev.hpp) -- 82 v_mad_u64_u32 per product and one Montgomery reduction per
// lane at the very end, instead of a full fq_mul (136 + carries) per product -- and the operands are formed with
// plain 256-bit additions instead of modular differences: with '+' for "entry at 0 plus entry at 1" (= twice the value
// at 1/2) the kernel sums  U[a][b] = sum_j F[a][b] G[a][b],  a, b in {0, 1, +}  (a on the top variable, b on the second),
// nine evaluations of the same biquadratic from which k_sum9_from_u gets the nine S by exact linear maps:
//   V[a][Y] = U[a][Y] (Y = 0, 1),  V[a][2] = 3 U[a][0] + 6 U[a][1] - 2 U[a][+];   S[3Y + 0] = V[0][Y],  S[3Y + 1] = V[1][Y],
//   S[3Y + 2] = 2 V[0][Y] + 2 V[1][Y] - V[+][Y]
// (a quadratic R with R(0), R(1), Q = 4 R(1/2) has leading coefficient 2 R(0) + 2 R(1) - Q and R(2) = 3 R(0) + 6 R(1) - 2 Q).
// An unreduced accumulator is 20 u64 columns: ONE product per wave.  Block = (64, 9), wave w = 3 b + a forms U[a][b]
// for the 64 quarter-indices of a tile; 80 VGPRs, two workgroups (18 waves) per CU.  (A form with three accumulators
// per lane and 3-wave workgroups measured 3.5 % slower: at ~195 VGPRs its six waves per CU sit 2/2/1/1 on the SIMDs
// and meet at a barrier every step.)  The tiles (64 quarter-indices x 8 entries of the term = 16 KiB) are staged by
// LDS-DMA (global_load_lds_dwordx4: no VGPR destination) into a ring of three buffers: two tiles are in flight under
// each tile's arithmetic, every entry is read from HBM exactly once, and all waves take their operands from LDS.
// Counted vmcnt + raw s_barrier (cdna_hip_programming.md, "Pipelining across barriers"); `sh` is the kernel's only
// __shared__ object.  The reduction divides by 2^280 instead of 2^256: every lane multiplies its sum by 2^24 (fix24) once.
//   q must be a multiple of 64.  partials: [((term * 3 + b) * gridDim.x + bx) * 3 + a]
__device__ __forceinline__ fq lds_elem(const u32x4 (&slot)[2][64], int lane) {
    const u32x4 lo = slot[0][lane], hi = slot[1][lane];
    fq x;
#pragma unroll
    for (int w = 0; w < 4; ++w) { x.v[w] = lo[w]; x.v[4 + w] = hi[w]; }
    return x;
}
// PWQ: the relation is P W + Q * 1 (T.t[0] = P, T.t[1] = W, T.t[2] = Q; the constant-one factor T.t[3] is never read): the workgroups of
// term 1 stage Q's four rows only and ADD -- U[a][b] of (Q, ones) is |a| |b| times the plain sum of Q[a][b] (a set of one entry counts 1,
// '+' counts 2) -- so they leave the machine to term 0's products (GKR phases at 2^20 entries: the first sweep of every phase).
template <bool PWQ>
__global__ void __launch_bounds__(576) k_sum9_lazy(Tab4 T, size_t q, fq fix24, fq* __restrict__ partials) {
    __shared__ u32x4 sh[3][8][2][64];
    const int term = blockIdx.y, lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    // which of the nine products this wave forms.  The nine waves of a workgroup sit 3/2/2/2 on the four SIMDs (wave w on
    // SIMD w % 4) and meet at a barrier every tile, and a product over '+' operands costs 1 or 3 extra 256-bit additions
    // per factor: the SIMD with three waves (0, 4, 8) gets three of the plain products, the (+,+) product shares a SIMD
    // with the fourth.  Nibble w of the constant = 3B + A.
    const int role = (int)((0x375416280ull >> (4 * w)) & 15);
    const int A = role % 3, B = role / 3; // 0, 1, 2 = '+'
    const fq* __restrict__ F = T.t[2 * term];
    const fq* __restrict__ G = T.t[2 * term + 1];
    u64 acc[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = 0;
    u32 lin[9]; // linear term: the plain sum of this lane's operands, 256 bits + a word of carries
#pragma unroll
    for (int k = 0; k < 9; ++k) lin[k] = 0;
    const size_t stride = (size_t)gridDim.x * 64, base0 = (size_t)blockIdx.x * 64;
    const int nt = base0 < q ? (int)((q - base0 + stride - 1) / stride) : 0;
    const bool linear = PWQ && term == 1; // workgroup-uniform
    auto stage = [&](int t) { // rows 2w, 2w + 1 (= slot w, both halves) for w < 8
        if (w < (linear ? 4 : 8)) {
            const size_t j = base0 + (size_t)t * stride + lane;
            const fq* src = ((w & 4) ? G : F) + (size_t)(w & 3) * q + j;
            u32x4* buf = &sh[t % 3][w][0][0];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)reinterpret_cast<const u32x4*>(src),
                                             (__attribute__((address_space(3))) void*)buf, 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const u32x4*>(src) + 1),
                                             (__attribute__((address_space(3))) void*)(buf + 64), 16, 0, 2);
        }
    };
    if (nt > 0) stage(0);
    if (nt > 1) stage(1);
    int since = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // leaves tile t+1's two rows in flight (wave 8 has none)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nt) stage(t + 2);
        const u32x4 (&tile)[8][2][64] = sh[t % 3];
        // operand of factor f: the sum of its entries over b1 in a's set and b2 in b's set (one, two or four entries)
        auto opnd = [&](int f) -> fq {
            const int b1lo = A == 1 ? 1 : 0, b2lo = B == 1 ? 1 : 0;
            fq x = lds_elem(tile[4 * f + 2 * b1lo + b2lo], lane);
            if (B == 2) x = fq_add_plain(x, lds_elem(tile[4 * f + 2 * b1lo + 1], lane));
            if (A == 2) {
                fq y = lds_elem(tile[4 * f + 2 + b2lo], lane);
                if (B == 2) y = fq_add_plain(y, lds_elem(tile[4 * f + 3], lane));
                x = fq_add_plain(x, y);
            }
            return x;
        };
        if (linear) {
            const fq x = opnd(0); // < 4p < 2^256
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) lin[k] = __builtin_addc(lin[k], x.v[k], c, &c);
            lin[8] += c;
            continue;
        }
        u32 lf[10], lg[10];
        fq28_unpack(opnd(0), lf);
        fq28_unpack(opnd(1), lg);
        fq28_mac_vv(acc, lf, lg);
        if (++since == 8) { // fq28_mac_vv's bound for operands up to 4p
            since = 0;
            fq28_carry(acc);
        }
    }
    if (linear) {
        fq lo, hi = fq_zero();
#pragma unroll
        for (int k = 0; k < 8; ++k) lo.v[k] = lin[k];
        hi.v[0] = lin[8];
        const fq r2 = {FQ_R2_LIMBS};
        fq sgl = fq_add(fq_reduce_256(lo), fq_mul(hi, r2)); // the sum of the stored (Montgomery) values mod p
        if (A == 2) sgl = fq_add(sgl, sgl);                   // times the ones factor's plain sum over the same set: 1, 2 or 4
        if (B == 2) sgl = fq_add(sgl, sgl);
        const fq rl = fq_wave_sum(sgl);
        if (lane == 0) fq_store(&partials[((size_t)(term * 3 + B) * gridDim.x + blockIdx.x) * 3 + A], rl);
        return;
    }
    const fq r = fq_wave_sum(fq_mul(fq28_redc(acc), fix24));
    if (lane == 0) fq_store(&partials[((size_t)(term * 3 + B) * gridDim.x + blockIdx.x) * 3 + A], r);
}

// ---- the nine sums on the matrix cores ---------------------------------------------------------------------------------------
// U[a][b] = sum_j F[a][b] G[a][b] with F[a][b], G[a][b] plain sums of the term's entries e = 2 b1 + b2 at quarter-index j, so every
// U is a sum of the sixteen cross sums P[e][e'] = sum_j F_e[j] G_e'[j] (as integers of the stored limbs).  Cut into BYTES,
//   P[e][e'] = sum_{i,i'} 256^(i+i') C[i][i'],   C[i][i'] = sum_j F_e[j].byte[i] * G_e'[j].byte[i'],
// and C is a 32 x 32 x (number of j) matrix product with the contraction over j: v_mfma_i32_32x32x32_i8 takes 32 quarter-indices
// per instruction.  Its operands want byte i of 16 consecutive entries per lane -- the transpose of the table's layout -- which
// ds_read_b64_tr_b8 delivers from the entry-major LDS tile for free (per 16 lanes: 8 rows x 16 byte columns, column-major out;
// tools/probe_tr8.hip).  Signed bytes: both operands are presented as byte - 128, and
//   sum_j F G = sum_k 256^k D[k] + 128 c1 (A + B) - 2^14 c1^2 N,   D[k] = sum_{i+i'=k} C_signed[i][i'],  c1 = (2^256 - 1)/255,
// with A = sum_j F_e[j], B = sum_j G_e'[j] (plain modular sums, one 256-bit add per entry) and N the number of j.  Exact integer
// arithmetic: the sums are the same field elements as k_sum9 / k_sum9_lazy give.
// Workgroup = 4 waves on one term; wave w keeps the four accumulators C[e = w][e' = 0..3] (64 registers) and the plain sums of
// rows F_w and G_w; tiles of 64 quarter-indices (8 rows x 2 KiB) come through a three-tile LDS-DMA ring as in k_sum9_lazy.
// partials: [(term * gridDim.x + bx) * 24 + ...]: 16 values P_signed[e][e'] * 2^-256 (index 4 e + e'), then A_e (4), then B_e' (4).
#include "pow256_tab.inc"

__global__ void __launch_bounds__(kThreads) k_sum16_mfma(Tab4 T, size_t q, fq* __restrict__ partials) {
    __shared__ u32x4 sh[3][8][128]; // three tiles x (F_0..F_3, G_0..G_3) x 64 entries x 32 B
    const int term = blockIdx.y, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const fq* __restrict__ F = T.t[2 * term];
    const fq* __restrict__ G = T.t[2 * term + 1];
    const size_t stride = (size_t)gridDim.x * 64, base0 = (size_t)blockIdx.x * 64;
    const int nt = base0 < q ? (int)((q - base0 + stride - 1) / stride) : 0;
    auto stage = [&](int t) { // wave w brings rows 2w, 2w + 1: four DMA instructions
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int r = 2 * w + r2;
            const fq* src = ((r & 4) ? G : F) + (size_t)(r & 3) * q + base0 + (size_t)t * stride;
            const u32x4* p = reinterpret_cast<const u32x4*>(src) + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)&sh[t % 3][r][0], 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 64),
                                             (__attribute__((address_space(3))) void*)&sh[t % 3][r][64], 16, 0, 2);
        }
    };
    v16i acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0;
    u32 sf[9], sg[9]; // plain sums of this lane's entries of rows F_w and G_w (value < 2^256 * tiles)
#pragma unroll
    for (int k = 0; k < 9; ++k) sf[k] = sg[k] = 0;
    // transposed-read address of this lane inside a row: lane 16 gg + 2 qq + pp reads entry 16 (gg >> 1) + qq (+ 8 for the second
    // read, + 32 for the second K step), bytes 16 (gg & 1) + 8 pp .. + 7
    const int gg = lane >> 4, qq = (lane >> 1) & 7, pp = lane & 1;
    const u32 tr_off = (u32)((16 * (gg >> 1) + qq) * 32 + 16 * (gg & 1) + 8 * pp);
    if (nt > 0) stage(0);
    if (nt > 1) stage(1);
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // leaves tile t+1's four instructions in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nt) stage(t + 2);
        const u32 tile = (u32)(uintptr_t)&sh[t % 3][0][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { // 32 quarter-indices per step
            const u32 a0 = tile + (u32)w * 2048u + (u32)ks * 1024u + tr_off; // row F_w
            const u32 b0 = tile + 4u * 2048u + (u32)ks * 1024u + tr_off;      // rows G_0 .. G_3
            u64 fa0, fa1, gb[4][2];
            asm volatile(
                "ds_read_b64_tr_b8 %0, %10\n\t"
                "ds_read_b64_tr_b8 %1, %10 offset:256\n\t"
                "ds_read_b64_tr_b8 %2, %11\n\t"
                "ds_read_b64_tr_b8 %3, %11 offset:256\n\t"
                "ds_read_b64_tr_b8 %4, %11 offset:2048\n\t"
                "ds_read_b64_tr_b8 %5, %11 offset:2304\n\t"
                "ds_read_b64_tr_b8 %6, %11 offset:4096\n\t"
                "ds_read_b64_tr_b8 %7, %11 offset:4352\n\t"
                "ds_read_b64_tr_b8 %8, %11 offset:6144\n\t"
                "ds_read_b64_tr_b8 %9, %11 offset:6400\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fa0), "=&v"(fa1), "=&v"(gb[0][0]), "=&v"(gb[0][1]), "=&v"(gb[1][0]), "=&v"(gb[1][1]), "=&v"(gb[2][0]), "=&v"(gb[2][1]),
                  "=&v"(gb[3][0]), "=&v"(gb[3][1])
                : "v"(a0), "v"(b0)
                : "memory");
            const u64 sx = 0x8080808080808080ull;
            fa0 ^= sx;
            fa1 ^= sx;
            const v4i A = v4i{(int)(u32)fa0, (int)(u32)(fa0 >> 32), (int)(u32)fa1, (int)(u32)(fa1 >> 32)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u64 x0 = gb[e][0] ^ sx, x1 = gb[e][1] ^ sx;
                const v4i B = v4i{(int)(u32)x0, (int)(u32)(x0 >> 32), (int)(u32)x1, (int)(u32)(x1 >> 32)};
                acc[e] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[e], 0, 0, 0);
            }
        }
        { // plain sums: lane l adds entry l of rows F_w and G_w
            const u32x4 flo = sh[t % 3][w][2 * lane], fhi = sh[t % 3][w][2 * lane + 1];
            const u32x4 glo = sh[t % 3][4 + w][2 * lane], ghi = sh[t % 3][4 + w][2 * lane + 1];
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sf[k] = __builtin_addc(sf[k], flo[k], c, &c);
#pragma unroll
            for (int k = 0; k < 4; ++k) sf[4 + k] = __builtin_addc(sf[4 + k], fhi[k], c, &c);
            sf[8] += c;
            c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] = __builtin_addc(sg[k], glo[k], c, &c);
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[4 + k] = __builtin_addc(sg[4 + k], ghi[k], c, &c);
            sg[8] += c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the tile is consumed before its slot is restaged
        }
    }
    // ---- epilogue: once per wave
    __syncthreads(); // every wave is done with the ring: it becomes scratch (4 KiB per wave)
    int* mat = reinterpret_cast<int*>(&sh[0][0][0]) + w * 1024;
    const fq r2 = {FQ_R2_LIMBS};
    fq mine[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        // C layout of the 32x32 result: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int k = 0; k < 16; ++k) mat[((k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[e][k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // D[k], k = lane: the anti-diagonal i + i' = k of the 32 x 32 matrix (lane 63: empty).  Fixed trip count, in-range addresses,
        // the bounds as a select: the 32 LDS reads issue back to back (a loop over [klo, khi] waited for every read)
        long long d = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = lane - i, v = mat[i * 32 + (j & 31)];
            d += (j >= 0 && j < 32) ? v : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // D[k] * 256^k * 2^-256 mod p: |D| < 2^37 as a field element times the constant, sign restored afterwards
        const bool neg = d < 0;
        const unsigned long long ad = (unsigned long long)(neg ? -d : d);
        fq x = fq_zero();
        x.v[0] = (u32)ad;
        x.v[1] = (u32)(ad >> 32);
        fq wk;
#pragma unroll
        for (int k = 0; k < 8; ++k) wk.v[k] = kPow256ModP[lane < 63 ? lane : 0][k];
        fq prod = fq_mul(x, wk);
        if (neg) prod = fq_sub(fq_zero(), prod);
        mine[e] = fq_wave_sum(prod);
    }
    // plain sums -> field elements: low 256 bits reduced, the overflow word times 2^256 mod p (fq_mul by R^2 = times R)
    fq lo, hi;
#pragma unroll
    for (int k = 0; k < 8; ++k) lo.v[k] = sf[k];
    hi = fq_zero();
    hi.v[0] = sf[8];
    const fq asum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
#pragma unroll
    for (int k = 0; k < 8; ++k) lo.v[k] = sg[k];
    hi.v[0] = sg[8];
    const fq bsum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
    if (lane == 0) {
        fq* out = partials + ((size_t)term * gridDim.x + blockIdx.x) * 24;
#pragma unroll
        for (int e = 0; e < 4; ++e) fq_store(out + 4 * w + e, mine[e]);
        fq_store(out + 16 + w, asum);
        fq_store(out + 20 + w, bsum);
    }
}

// second stage of k_sum16_mfma: one block per (term, value): out48[term * 24 + v] = sum over the workgroups (fixed order)
__global__ void __launch_bounds__(kThreads) k_reduce48(const fq* __restrict__ partials, int gx, fq* __restrict__ out48) {
    const int term = blockIdx.x / 24, v = blockIdx.x % 24;
    fq acc[1] = {fq_zero()};
    for (int b = threadIdx.x; b < gx; b += kThreads) acc[0] = fq_add(acc[0], fq_load(partials + ((size_t)term * gx + b) * 24 + v));
    block_sum_store<1>(acc, out48);
}
// 2 x (16 signed cross sums, 4 + 4 plain sums) -> the nine sums S[3Y + t] (through U, as k_sum9_from_u).  One block of 64 threads.
__global__ void __launch_bounds__(64) k_sum9_from_p16(const fq* __restrict__ in48, fq qn /* the number of quarter-indices, as a plain integer */,
                                                      fq* __restrict__ out) {
    __shared__ fq P[2][16];
    __shared__ fq U[9];
    const int t = threadIdx.x;
    if (t < 32) { // P[e][e'] = P_signed + 128 c1 (A_e + B_e') 2^-256 - 2^14 c1^2 N 2^-256
        const int term = t >> 4, e = (t >> 2) & 3, e2 = t & 3;
        const fq k1 = {FQ_SUM16_K1}, k2 = {FQ_SUM16_K2};
        const fq ab = fq_add(fq_load(in48 + term * 24 + 16 + e), fq_load(in48 + term * 24 + 20 + e2));
        P[term][4 * e + e2] = fq_add(fq_add(fq_load(in48 + term * 24 + 4 * e + e2), fq_mul(k1, ab)), fq_mul(k2, qn));
    }
    __syncthreads();
    if (t < 9) { // U[a][b]: a on the top variable (bit 1 of e), b on the second (bit 0); 0, 1, 2 = '+'
        const int a = t % 3, b = t / 3;
        fq u = fq_zero();
        for (int term = 0; term < 2; ++term)
            for (int e = 0; e < 4; ++e)
                for (int e2 = 0; e2 < 4; ++e2) {
                    const bool in1 = (a == 2 || (e >> 1) == a) && (b == 2 || (e & 1) == b);
                    const bool in2 = (a == 2 || (e2 >> 1) == a) && (b == 2 || (e2 & 1) == b);
                    if (in1 && in2) u = fq_add(u, P[term][4 * e + e2]);
                }
        U[3 * b + a] = u;
    }
    __syncthreads();
    if (t < 9) {
        const int Y = t / 3, tt = t % 3;
        auto V = [&](int a) -> fq { // V[a][Y]
            if (Y < 2) return U[3 * Y + a];
            const fq u0 = U[a], u1 = U[3 + a], up = U[6 + a];
            const fq u1x2 = fq_dbl(u1);
            return fq_sub(fq_add(fq_add(fq_dbl(u0), u0), fq_add(fq_dbl(u1x2), u1x2)), fq_dbl(up));
        };
        fq r;
        if (tt < 2) r = V(tt);
        else r = fq_sub(fq_dbl(fq_add(V(0), V(1))), V(2));
        fq_store(out + t, r);
    }
}

// ---- THREE rounds per pass: the 8 x 8 cross sums of a term on the matrix cores -------------------------------------------------
// With the three top variables as index e = 4 b1 + 2 b2 + b3 (tables of 8 q entries, entry e of quarter-index j at e q + j) the
// 64 cross sums  P[e][e'] = sum_j F_e[j] G_e'[j]  per term determine THREE rounds on the host (zkmi.cpp, host_rounds_from_cross:
// round i needs only sum_{e,e'} eq(x,e) eq(x,e') P[e][e'] at x = (r_0..r_{i-1}, X, y) -- exact arithmetic, the same field elements
// as the reference's per-round sums), and one fold with the eight weights eq((r0,r1,r2), .) binds all three variables: per prove
// the tables are read 8/7 + (9/8)(8/7) = 2.43 times instead of 3.0 (two rounds per pass).  Same scheme as k_sum16_mfma -- bytes
// as signed i8, v_mfma_i32_32x32x32_i8 with the contraction over j, operands by ds_read_b64_tr_b8 -- with eight waves per workgroup:
// wave w = (wa, wh) keeps the 2 x 4 block C[2 wa + i][4 wh + k] (128 accumulator registers; six operand rows per step instead of the
// nine of a row-per-wave split: 28 instead of 41 LDS instructions per wave and tile) and the plain sums of rows F_w and G_w; a tile is 16 rows x 2 KiB,
// staged by LDS-DMA through a ring of R tiles (R - 1 in flight; 32 R KiB of LDS: one workgroup per CU).  Measured at 4 x 2^26: 5.9 TB/s;
// the same ring with the matrix and LDS work removed 6.2, two 2-tile workgroups per CU 5.3, default instead of nt load policy 5.6.
// partials: [(term * gridDim.x + bx) * 80 + ...]: 64 values P_signed[e][e'] * 2^-256 (index 8 e + e'), then A_e (8), then B_e' (8).
template <int R>
__global__ void __launch_bounds__(512) k_cross64_mfma(Tab4 T, size_t q, fq* __restrict__ partials, long long* __restrict__ raw) {
    constexpr int NE = 8;
    __shared__ u32x4 sh[R][2 * NE][128];
    const int term = blockIdx.y, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = w & 3, wh = w >> 2; // this wave's products: C[2 wa + i][4 wh + k], i < 2, k < 4 (6 operand rows instead of 9)
    const fq* __restrict__ F = T.t[2 * term];
    const fq* __restrict__ G = T.t[2 * term + 1];
    const size_t stride = (size_t)gridDim.x * 64, base0 = (size_t)blockIdx.x * 64;
    const int nt = base0 < q ? (int)((q - base0 + stride - 1) / stride) : 0;
    auto stage = [&](int t) { // wave w brings rows 2w, 2w + 1: four DMA instructions
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int r = 2 * w + r2;
            const fq* src = ((r & NE) ? G : F) + (size_t)(r & (NE - 1)) * q + base0 + (size_t)t * stride;
            const u32x4* p = reinterpret_cast<const u32x4*>(src) + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)&sh[t % R][r][0], 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 64),
                                             (__attribute__((address_space(3))) void*)&sh[t % R][r][64], 16, 0, 2);
        }
    };
    v16i acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0;
    u32 sf[9], sg[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) sf[k] = sg[k] = 0;
    const int gg = lane >> 4, qq = (lane >> 1) & 7, pp = lane & 1; // transposed-read address inside a row: see k_sum16_mfma
    const u32 tr_off = (u32)((16 * (gg >> 1) + qq) * 32 + 16 * (gg & 1) + 8 * pp);
#pragma unroll
    for (int t = 0; t < R - 1; ++t)
        if (t < nt) stage(t);
    for (int t = 0; t < nt; ++t) {
        if (t + R - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (R - 2)) : "memory"); // tiles t+1 .. t+R-2 stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + R - 1 < nt) stage(t + R - 1); // into the slot every wave finished reading before this barrier
        const u32 tile = (u32)(uintptr_t)&sh[t % R][0][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { // 32 quarter-indices per step; this wave: rows F_2a, F_2a+1 against G_4h .. G_4h+3
            const u32 a0 = tile + (u32)(2 * wa) * 2048u + (u32)ks * 1024u + tr_off;
            const u32 b0 = tile + (u32)(NE + 4 * wh) * 2048u + (u32)ks * 1024u + tr_off;
            u64 fa[2][2], gb[4][2];
            asm volatile(
                "ds_read_b64_tr_b8 %0, %12\n\t"
                "ds_read_b64_tr_b8 %1, %12 offset:256\n\t"
                "ds_read_b64_tr_b8 %2, %12 offset:2048\n\t"
                "ds_read_b64_tr_b8 %3, %12 offset:2304\n\t"
                "ds_read_b64_tr_b8 %4, %13\n\t"
                "ds_read_b64_tr_b8 %5, %13 offset:256\n\t"
                "ds_read_b64_tr_b8 %6, %13 offset:2048\n\t"
                "ds_read_b64_tr_b8 %7, %13 offset:2304\n\t"
                "ds_read_b64_tr_b8 %8, %13 offset:4096\n\t"
                "ds_read_b64_tr_b8 %9, %13 offset:4352\n\t"
                "ds_read_b64_tr_b8 %10, %13 offset:6144\n\t"
                "ds_read_b64_tr_b8 %11, %13 offset:6400\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fa[1][0]), "=&v"(fa[1][1]), "=&v"(gb[0][0]), "=&v"(gb[0][1]), "=&v"(gb[1][0]),
                  "=&v"(gb[1][1]), "=&v"(gb[2][0]), "=&v"(gb[2][1]), "=&v"(gb[3][0]), "=&v"(gb[3][1])
                : "v"(a0), "v"(b0)
                : "memory");
            const u64 sx = 0x8080808080808080ull;
            v4i B[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 x0 = gb[k][0] ^ sx, x1 = gb[k][1] ^ sx;
                B[k] = v4i{(int)(u32)x0, (int)(u32)(x0 >> 32), (int)(u32)x1, (int)(u32)(x1 >> 32)};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u64 f0 = fa[i][0] ^ sx, f1 = fa[i][1] ^ sx;
                const v4i A = v4i{(int)(u32)f0, (int)(u32)(f0 >> 32), (int)(u32)f1, (int)(u32)(f1 >> 32)};
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[4 * i + k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B[k], acc[4 * i + k], 0, 0, 0);
            }
        }
        { // plain sums: lane l adds entry l of rows F_w and G_w
            const u32x4 flo = sh[t % R][w][2 * lane], fhi = sh[t % R][w][2 * lane + 1];
            const u32x4 glo = sh[t % R][NE + w][2 * lane], ghi = sh[t % R][NE + w][2 * lane + 1];
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sf[k] = __builtin_addc(sf[k], flo[k], c, &c);
#pragma unroll
            for (int k = 0; k < 4; ++k) sf[4 + k] = __builtin_addc(sf[4 + k], fhi[k], c, &c);
            sf[8] += c;
            c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[k] = __builtin_addc(sg[k], glo[k], c, &c);
#pragma unroll
            for (int k = 0; k < 4; ++k) sg[4 + k] = __builtin_addc(sg[4 + k], ghi[k], c, &c);
            sg[8] += c;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the tile is consumed before its slot is restaged
        }
    }
    // ---- epilogue: once per wave (as k_sum16_mfma)
    __syncthreads(); // every wave is done with the ring: it becomes scratch (4 KiB per wave)
    int* mat = reinterpret_cast<int*>(&sh[0][0][0]) + w * 1024;
    const fq r2 = {FQ_R2_LIMBS};
    fq* out = partials + ((size_t)term * gridDim.x + blockIdx.x) * (NE * NE + 2 * NE);
    fq wk;
#pragma unroll
    for (int k = 0; k < 8; ++k) wk.v[k] = kPow256ModP[lane < 63 ? lane : 0][k];
    for (int e = 0; e < NE; ++e) { // rolled: the body is large and runs once
        v16i a = acc[0];
#pragma unroll
        for (int e2 = 1; e2 < NE; ++e2)
            if (e == e2) a = acc[e2];
#pragma unroll
        for (int k = 0; k < 16; ++k) mat[((k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = a[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // D[k], k = lane: the anti-diagonal i + i' = k of the 32 x 32 matrix (lane 63: empty).  Fixed trip count, in-range addresses,
        // the bounds as a select: the 32 LDS reads issue back to back (a loop over [klo, khi] waited for every read)
        long long d = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = lane - i, v = mat[i * 32 + (j & 31)];
            d += (j >= 0 && j < 32) ? v : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (raw) { // the digits themselves: k_reduce_cross_raw adds them over the workgroups as integers and multiplies once
            raw[((((size_t)term * gridDim.x + blockIdx.x) * (NE * NE)) + (size_t)(NE * (2 * wa + (e >> 2)) + 4 * wh + (e & 3))) * 64 + lane] = d;
            continue;
        }
        const bool neg = d < 0;
        const unsigned long long ad = (unsigned long long)(neg ? -d : d);
        fq x = fq_zero();
        x.v[0] = (u32)ad;
        x.v[1] = (u32)(ad >> 32);
        fq prod = fq_mul(x, wk);
        if (neg) prod = fq_sub(fq_zero(), prod);
        const fq mine = fq_wave_sum(prod);
        if (lane == 0) fq_store(out + NE * (2 * wa + (e >> 2)) + 4 * wh + (e & 3), mine);
    }
    fq lo, hi;
#pragma unroll
    for (int k = 0; k < 8; ++k) lo.v[k] = sf[k];
    hi = fq_zero();
    hi.v[0] = sf[8];
    const fq asum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
#pragma unroll
    for (int k = 0; k < 8; ++k) lo.v[k] = sg[k];
    hi.v[0] = sg[8];
    const fq bsum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
    if (lane == 0) {
        fq_store(out + NE * NE + w, asum);
        fq_store(out + NE * NE + NE + w, bsum);
    }
}
// ---- the three-variable fold FUSED with the next level's cross sums -----------------------------------------------------------------
// A three-round pass is fold (read 8 entries, write 1) + cross sums of the result (read it again).  Here the fold's outputs go to
// HBM AND into the LDS tile the matrix stage reads, so the folded tables are not re-read: per prove the tables move 1 + 9/7 = 2.29
// times instead of 2.43.  A workgroup = 8 waves on one term; per tile of 64 next-level eighth-indices j'' wave w folds the two
// tile rows r = 2w, 2w + 1 (row r = entry e'' = r & 7 of factor r >> 3: dst[e'' q'' + j''] = sum_b W_b src[b outlen + e'' q'' + j'']),
// each from eight source rows streamed through the wave's own LDS-DMA ring (D slots of 2 KiB, D - 1 in flight, counted vmcnt: the
// stream runs on across tiles and under the matrix stage), stores them, writes them into tile buffer t & 1 and adds them to its two
// plain sums; one s_barrier per tile; then the matrix stage of k_cross64_mfma on the tile (wave w: C[w][0..7]).  Two tile buffers:
// a wave passes barrier t + 1 only after its matrix stage t, so buffer t & 1 is free when tile t + 2 is written.
// wts: the eight weights eq((r0,r1,r2), b) * 2^280 as 10 x 28-bit limbs each (k_eq_weights28), scalar loads.
// partials as k_cross64_mfma: [(term * gridDim.x + bx) * 80 + ...], plain sums at 64 + r.
template <bool BIG, int D>
__global__ void __launch_bounds__(512) k_fold3_cross64(Tab4 S, Tab4w Dst, size_t outlen, const u32* __restrict__ wts, fq* __restrict__ partials) {
    constexpr int NE = 8;
    __shared__ u32x4 tile[2 * NE][128];  // ONE tile buffer: a second barrier per tile guards its reuse (see below)
    __shared__ u32x4 ring[8][D][128];
    __shared__ u32 pl[8][2][9][64];      // per lane: plain sums of the folded rows 2w and 2w + 1 (kept out of the registers)
    const int term = blockIdx.y, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t qn = outlen >> 3; // the next level's eighth
    const size_t stride = (size_t)gridDim.x * 64, base0 = (size_t)blockIdx.x * 64;
    const int nt = base0 < qn ? (int)((qn - base0 + stride - 1) / stride) : 0;
    const int nsteps = nt * 16; // source rows of this wave: step s = 16 t + 8 rr + b
    const bool late = w >= 4;   // wave-uniform
    // rows 2w, 2w + 1 belong to the same factor table (w < 4: F, else G)
    const fq* __restrict__ src = S.t[2 * term + (w >> 2)];
    fq* __restrict__ dst = Dst.t[2 * term + (w >> 2)];
    const size_t e0 = (size_t)((2 * w) & 7); // entry of row 2w; row 2w + 1: e0 + 1
    auto issue = [&](int s) {
        const int t = s >> 4, rr = (s >> 3) & 1, b = s & 7;
        const fq* p0 = src + (size_t)b * outlen + (e0 + (size_t)rr) * qn + base0 + (size_t)t * stride;
        const u32x4* p = reinterpret_cast<const u32x4*>(p0) + lane;
        const int slot = s % D;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)&ring[w][slot][0], 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 64),
                                         (__attribute__((address_space(3))) void*)&ring[w][slot][64], 16, 0, 2);
    };
    v16i acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) pl[w][0][k][lane] = pl[w][1][k][lane] = 0;
    const int gg = lane >> 4, qq = (lane >> 1) & 7, pp = lane & 1; // transposed-read address inside a row: see k_sum16_mfma
    const u32 tr_off = (u32)((16 * (gg >> 1) + qq) * 32 + 16 * (gg & 1) + 8 * pp);
    auto matrix_stage = [&]() {
        const u32 tb = (u32)(uintptr_t)&tile[0][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { // 32 eighth-indices per step; the G rows in two halves (the operands of four products at a time)
            const u32 a0 = tb + (u32)w * 2048u + (u32)ks * 1024u + tr_off;  // row F_w
            u64 fa0, fa1;
            asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:256" : "=&v"(fa0), "=&v"(fa1) : "v"(a0) : "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 b0 = tb + (u32)(NE + 4 * h) * 2048u + (u32)ks * 1024u + tr_off; // rows G_4h .. G_4h+3
                u64 gb[4][2];
                asm volatile(
                    "ds_read_b64_tr_b8 %0, %10\n\t"
                    "ds_read_b64_tr_b8 %1, %10 offset:256\n\t"
                    "ds_read_b64_tr_b8 %2, %10 offset:2048\n\t"
                    "ds_read_b64_tr_b8 %3, %10 offset:2304\n\t"
                    "ds_read_b64_tr_b8 %4, %10 offset:4096\n\t"
                    "ds_read_b64_tr_b8 %5, %10 offset:4352\n\t"
                    "ds_read_b64_tr_b8 %6, %10 offset:6144\n\t"
                    "ds_read_b64_tr_b8 %7, %10 offset:6400\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=&v"(gb[0][0]), "=&v"(gb[0][1]), "=&v"(gb[1][0]), "=&v"(gb[1][1]), "=&v"(gb[2][0]), "=&v"(gb[2][1]), "=&v"(gb[3][0]),
                      "=&v"(gb[3][1]), "+v"(fa0), "+v"(fa1)
                    : "v"(b0)
                    : "memory");
                const u64 sx = 0x8080808080808080ull;
                const u64 f0 = fa0 ^ sx, f1 = fa1 ^ sx;
                const v4i A = v4i{(int)(u32)f0, (int)(u32)(f0 >> 32), (int)(u32)f1, (int)(u32)(f1 >> 32)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u64 x0 = gb[e][0] ^ sx, x1 = gb[e][1] ^ sx;
                    const v4i B = v4i{(int)(u32)x0, (int)(u32)(x0 >> 32), (int)(u32)x1, (int)(u32)(x1 >> 32)};
                    acc[4 * h + e] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[4 * h + e], 0, 0, 0);
                }
            }
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nsteps) issue(s);
    for (int t = 0; t < nt; ++t) {
#pragma nounroll
        for (int rr = 0; rr < 2; ++rr) {
            u64 a28[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) a28[k] = 0;
            // rolled: one source row at a time (an unrolled loop keeps sixteen row addresses and interleaved rows alive in registers
            // this kernel does not have)
#pragma nounroll
            for (int b = 0; b < 8; ++b) {
                const int s = 16 * t + 8 * rr + b;
                // younger than row s when it is waited for: rows s+1 .. s+D-1 (two instructions each) and, for b < D, the two store
                // instructions of the previous output row (none before the first); the last D rows of a wave drain
                if (s + D <= nsteps) {
                    if (b < D && s >= 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1) + 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (D - 1)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                const int slot = s % D;
                const u32x4 lo = ring[w][slot][2 * lane], hi = ring[w][slot][2 * lane + 1];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // in registers before the slot is refilled
                if (s + D < nsteps) issue(s + D);
                fq x;
#pragma unroll
                for (int k = 0; k < 4; ++k) { x.v[k] = lo[k]; x.v[4 + k] = hi[k]; }
                fq28_mac(a28, x, wts + 10 * b);
            }
            const fq o = fq28_redc(a28);
            fq* out = dst + (e0 + (size_t)rr) * qn + base0 + (size_t)t * stride + lane;
            if (BIG) fq_store_stream(out, o); else fq_store(out, o);
            if (rr == 0 && t > 0) { // every wave has finished the matrix stage of tile t - 1: the tile buffer is free
                if (late) matrix_stage();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            tile[2 * w + rr][2 * lane] = u32x4{o.v[0], o.v[1], o.v[2], o.v[3]};
            tile[2 * w + rr][2 * lane + 1] = u32x4{o.v[4], o.v[5], o.v[6], o.v[7]};
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32 v = __builtin_addc(pl[w][rr][k][lane], o.v[k], c, &c);
                pl[w][rr][k][lane] = v;
            }
            pl[w][rr][8][lane] += c;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's two tile rows are written
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!late) matrix_stage(); // waves 4..7 run it after the first row of the next tile: a SIMD's two waves alternate pipes
    }
    if (late && nt > 0) matrix_stage();
    // ---- epilogue: once per wave (as k_cross64_mfma)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // every wave is done with the tiles: they become scratch (4 KiB per wave)
    int* mat = reinterpret_cast<int*>(&tile[0][0]) + w * 1024;
    const fq r2 = {FQ_R2_LIMBS};
    fq* out = partials + ((size_t)term * gridDim.x + blockIdx.x) * (NE * NE + 2 * NE);
    fq wk;
#pragma unroll
    for (int k = 0; k < 8; ++k) wk.v[k] = kPow256ModP[lane < 63 ? lane : 0][k];
    for (int e = 0; e < NE; ++e) { // rolled: the body is large and runs once
        v16i a = acc[0];
#pragma unroll
        for (int e2 = 1; e2 < NE; ++e2)
            if (e == e2) a = acc[e2];
#pragma unroll
        for (int k = 0; k < 16; ++k) mat[((k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = a[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the anti-diagonal i + j = lane of the 32 x 32 matrix (lane 63 has none).  Fixed trip count, in-range addresses, the bounds
        // as a select: the 32 LDS reads issue back to back (a loop over [klo, khi] waited for every read: ~20 us per launch)
        long long d = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int j = lane - i, v = mat[i * 32 + (j & 31)];
            d += (j >= 0 && j < 32) ? v : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool neg = d < 0;
        const unsigned long long ad = (unsigned long long)(neg ? -d : d);
        fq x = fq_zero();
        x.v[0] = (u32)ad;
        x.v[1] = (u32)(ad >> 32);
        fq prod = fq_mul(x, wk);
        if (neg) prod = fq_sub(fq_zero(), prod);
        const fq mine = fq_wave_sum(prod);
        if (lane == 0) fq_store(out + NE * w + e, mine);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        fq lo, hi = fq_zero();
#pragma unroll
        for (int k = 0; k < 8; ++k) lo.v[k] = pl[w][rr][k][lane];
        hi.v[0] = pl[w][rr][8][lane];
        const fq sum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
        if (lane == 0) fq_store(out + NE * NE + 2 * w + rr, sum); // rows 0..7 = A_e, rows 8..15 = B_e'
    }
}

// ---- the same sweep with the FOLD on the matrix cores too (round 3) ---------------------------------------------------------------
// k_fold3_cross64 runs its fold half as radix-2^28 v_mad_u64_u32 chains (81 per source entry) beside the matrix stage; at its two
// waves per SIMD that arithmetic is not hidden (0.55 of the HBM peak, 5.7 x 10^8 vector instructions per 8.6 GB launch).  Here the fold is
// the byte-cut contraction of k_multifold_wide_mfma (kernels.hip): an output row of 64 entries is  sum over 4 b-pairs of
// [64 product columns x 64 K] x [64 K x 64 outputs]  with K = (row of the pair, byte), 16 v_mfma_i32_16x16x64_i8 per b-pair, the
// weights' signed-digit Toeplitz tiles (k_eq_weights_mfma, m = 3: 16 KiB) resident in LDS for the whole kernel, one radix-2^28
// reduction per output and the same constant kfix.
// Registers decide the shape: the 64 cross-sum blocks of a term are 128 accumulator registers per lane over eight waves, the fold's
// columns 64 more, and at 256 registers per wave (two waves per SIMD) that spills into the DMA waits.  So a workgroup is FOUR waves, one
// per SIMD with the whole 512-register file: wave w keeps the 4 x 4 block C[4 (w & 1) + i][4 (w >> 1) + k] (256 registers: exactly the
// accumulator half of the file) and folds tile rows 4w .. 4w + 3.  A step is one b-pair of one row: 2 source rows (4 KiB, four LDS-DMA
// instructions) through the wave's ring of NP steps, refilled as soon as the operands are in registers.  The 64 columns of an output come
// out of the matrix cores spread over the four 16-lane rows of the wave; v_permlane32_swap / v_permlane16_swap transpose them in
// registers (32 swaps per output row; no LDS round trip).  The folded row goes to HBM and, as byte ^ 0x80, into the tile: the matrix
// stage takes its operands as they are.  Plain sums A_e, B_e' per lane in LDS (one 256-bit addition per folded entry).
// LDS: tile 32 KiB + weights 16 KiB + rings 16 NP KiB + plain sums 36 KiB (NP <= 4: 148 KiB), one workgroup per CU.  Same partials
// layout as k_cross64_mfma.
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
// P[o4][k]: on entry this 16-lane row's words for output tile o4; on exit P[g'][k] = the words row g' held for THIS row's tile
template <int NW>
__device__ __forceinline__ void rows_transpose4(u32 (&P)[4][NW]) {
#pragma unroll
    for (int k = 0; k < NW; ++k) { // lanes 32..63 of the first operand <-> lanes 0..31 of the second
        u32x2 r = __builtin_amdgcn_permlane32_swap(P[0][k], P[2][k], false, false);
        P[0][k] = r[0];
        P[2][k] = r[1];
        r = __builtin_amdgcn_permlane32_swap(P[1][k], P[3][k], false, false);
        P[1][k] = r[0];
        P[3][k] = r[1];
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) { // odd rows of the first operand <-> even rows of the second
        u32x2 r = __builtin_amdgcn_permlane16_swap(P[0][k], P[1][k], false, false);
        P[0][k] = r[0];
        P[1][k] = r[1];
        r = __builtin_amdgcn_permlane16_swap(P[2][k], P[3][k], false, false);
        P[2][k] = r[0];
        P[3][k] = r[1];
    }
}
// the 64 x 64 column block of one output row (fa[mt][o4]: SIGNED columns 16 mt + 4 g + r of output (o4, n), lane = 16 g + n, each within
// +-2^27) -> the field element of output `lane`: sum_k 256^k (column_k + 2^27) * 2^-280 mod p + kfix (kfix carries -2^27 per column).
// c1 = 1, c8 = 2^8, c16 = 2^16, c24 = 2^24 as values the compiler cannot see through: the pack then is four v_mad_i64_i32 per word
// (a constant power of two would be turned into 64-bit shifts and additions, three times the instructions)
__device__ __forceinline__ fq mfma_fold_finish(const v4i (&fa)[4][4], const fq& kfix, int c1, int c8, int c16, int c24) {
    u32 P[4][8];
    const long long bias = (1ll << 27) + (1ll << 35) + (1ll << 43) + (1ll << 51); // 2^27 at each of the word's four columns
#pragma unroll
    for (int o4 = 0; o4 < 4; ++o4)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { // four consecutive columns -> one 64-bit word (below 2^53)
            const v4i a = fa[mt][o4];
            const long long x = (long long)a[0] * c1 + ((long long)a[1] * c8 + ((long long)a[2] * c16 + ((long long)a[3] * c24 + bias)));
            P[o4][2 * mt] = (u32)(u64)x;
            P[o4][2 * mt + 1] = (u32)((u64)x >> 32);
        }
    rows_transpose4<8>(P);
    // word 4 mt + g' (weight 2^(32 (4 mt + g'))) of this lane's output came from row g'
    u32 v[18];
    u64 cy = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int mt = k >> 2, gp = k & 3;
        cy += P[gp][2 * mt];
        v[k] = (u32)cy;
        cy = (cy >> 32) + P[gp][2 * mt + 1];
    }
    v[16] = (u32)cy;
    v[17] = (u32)(cy >> 32);
    u64 a28[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) { // bits [28 k, 28 k + 28): a funnel shift over two adjacent words (wd + 1 <= 17 for every k)
        const int bit = 28 * k, wd = bit >> 5, sh = bit & 31;
        a28[k] = __builtin_amdgcn_alignbit(v[wd + 1], v[wd], sh) & FQ28_MASK;
    }
    return fq_add(fq28_redc(a28), kfix);
}

template <bool BIG, int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_fold3_cross64_mfma(Tab4 S, Tab4w Dst, size_t outlen, const u32x4* __restrict__ wtiles, fq kfix, fq* __restrict__ partials,
                     long long* __restrict__ raw) {
    constexpr int NE = 8;
    __shared__ u32x4 tile[2 * NE][128];
    __shared__ u32x4 wl[4][4][64];
    __shared__ u32x4 ring[4][NP][2][128];
    __shared__ u32 pls[4][4][9][64]; // per lane: plain sums of the folded rows 4w .. 4w + 3 (36 KiB: kept out of the registers)
    static_assert(NP >= 2 && NP <= 4, "LDS: 32 + 16 + 16 NP + 36 KiB");
    const int term = blockIdx.y, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = w & 1, gi = w >> 1; // cross products: F rows 4 fi .. + 3 against G rows 4 gi .. + 3
    const int g = lane >> 4, n = lane & 15;
    const size_t qn = outlen >> 3; // the next level's eighth
    const size_t stride = (size_t)gridDim.x * 64, base0 = (size_t)blockIdx.x * 64;
    const int nt = base0 < qn ? (int)((qn - base0 + stride - 1) / stride) : 0;
    const int nsteps = nt * 16; // this wave's steps: s = 16 t + 4 rq + bp
    const fq* __restrict__ src = S.t[2 * term + (w >> 1)]; // rows 4w .. 4w + 3 belong to one factor table
    fq* __restrict__ dst = Dst.t[2 * term + (w >> 1)];
    const size_t e0 = (size_t)((4 * w) & 7);
    { // the weight tiles: 1024 x 16 B, once per workgroup
        u32x4* wflat = &wl[0][0][0];
#pragma unroll
        for (int k = 0; k < 4; ++k) wflat[threadIdx.x + 256 * k] = wtiles[threadIdx.x + 256 * k];
    }
    __syncthreads();
    const char* lanep = reinterpret_cast<const char*>(src) + (size_t)lane * 16; // this lane's 16 bytes of a row's first KiB
    auto issue = [&](int s, int slot) { // source rows 2 bp, 2 bp + 1 of step s
        const int t = s >> 4, rq = (s >> 2) & 3, bp = s & 3;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // The row's offset is wave-uniform (scalar registers); the asm keeps the compiler from hoisting the 32 row addresses of a tile
            // out of the loop as 64-bit pairs in vector registers (64 registers this kernel does not have): one add per row instead
            size_t off = ((size_t)(2 * bp + r) * outlen + (e0 + (size_t)rq) * qn + base0 + (size_t)t * stride) * sizeof(fq);
            asm volatile("" : "+s"(off));
            const char* p = lanep + off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)&ring[w][slot][r][0], 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 1024),
                                             (__attribute__((address_space(3))) void*)&ring[w][slot][r][64], 16, 0, 2);
        }
    };
    v16i acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[e][k] = 0;
    int c1 = 1, c8 = 1 << 8, c16 = 1 << 16, c24 = 1 << 24; // mfma_fold_finish: multipliers the compiler must not fold into shifts
    asm volatile("" : "+s"(c1), "+s"(c8), "+s"(c16), "+s"(c24));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k) pls[w][i][k][lane] = 0;
    const int qq = (lane >> 1) & 7, pp = lane & 1; // transposed-read address inside a row: see k_sum16_mfma
    const u32 tr_off = (u32)((16 * (g >> 1) + qq) * 32 + 16 * (g & 1) + 8 * pp);
    auto matrix_stage = [&]() {
        const u32 tb = (u32)(uintptr_t)&tile[0][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { // 32 eighth-indices per step; the bytes in the tile are already byte - 128
            const u32 a0 = tb + (u32)(4 * fi) * 2048u + (u32)ks * 1024u + tr_off;
            const u32 b0 = tb + (u32)(NE + 4 * gi) * 2048u + (u32)ks * 1024u + tr_off;
            u64 fa[4][2], gb[4][2];
            asm volatile(
                "ds_read_b64_tr_b8 %0, %16\n\t"
                "ds_read_b64_tr_b8 %1, %16 offset:256\n\t"
                "ds_read_b64_tr_b8 %2, %16 offset:2048\n\t"
                "ds_read_b64_tr_b8 %3, %16 offset:2304\n\t"
                "ds_read_b64_tr_b8 %4, %16 offset:4096\n\t"
                "ds_read_b64_tr_b8 %5, %16 offset:4352\n\t"
                "ds_read_b64_tr_b8 %6, %16 offset:6144\n\t"
                "ds_read_b64_tr_b8 %7, %16 offset:6400\n\t"
                "ds_read_b64_tr_b8 %8, %17\n\t"
                "ds_read_b64_tr_b8 %9, %17 offset:256\n\t"
                "ds_read_b64_tr_b8 %10, %17 offset:2048\n\t"
                "ds_read_b64_tr_b8 %11, %17 offset:2304\n\t"
                "ds_read_b64_tr_b8 %12, %17 offset:4096\n\t"
                "ds_read_b64_tr_b8 %13, %17 offset:4352\n\t"
                "ds_read_b64_tr_b8 %14, %17 offset:6144\n\t"
                "ds_read_b64_tr_b8 %15, %17 offset:6400\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fa[1][0]), "=&v"(fa[1][1]), "=&v"(fa[2][0]), "=&v"(fa[2][1]), "=&v"(fa[3][0]),
                  "=&v"(fa[3][1]), "=&v"(gb[0][0]), "=&v"(gb[0][1]), "=&v"(gb[1][0]), "=&v"(gb[1][1]), "=&v"(gb[2][0]), "=&v"(gb[2][1]),
                  "=&v"(gb[3][0]), "=&v"(gb[3][1])
                : "v"(a0), "v"(b0)
                : "memory");
            v4i A[4], B[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                A[i] = v4i{(int)(u32)fa[i][0], (int)(u32)(fa[i][0] >> 32), (int)(u32)fa[i][1], (int)(u32)(fa[i][1] >> 32)};
                B[i] = v4i{(int)(u32)gb[i][0], (int)(u32)(gb[i][0] >> 32), (int)(u32)gb[i][1], (int)(u32)(gb[i][1] >> 32)};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[4 * i + k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i], B[k], acc[4 * i + k], 0, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NP; ++s)
        if (s < nsteps) issue(s, s);
    int slot = 0; // s % NP
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int rows_done = 4 * t + rq; // output rows this wave has stored so far
            v4i fa[4][4]; // [mt][o4]: columns 16 mt + 4 g + r of output (o4, n)
#pragma unroll
            for (int bp = 0; bp < 4; ++bp) {
                const int s = 16 * t + 4 * rq + bp;
                // Younger than R(s) when it is waited for: R(s+1) .. R(s+NP-1), four instructions each, and the two store instructions of
                // every finished row whose stores were issued behind R(s): the row that ended e rows ago issued them behind
                // R(s - bp - 1 - 4 (e - 1) + NP), which is R(s) or older iff bp + 1 + 4 (e - 1) <= NP.  The last NP steps of a wave drain.
                if (s + NP <= nsteps) {
                    constexpr int kLoads = 4 * (NP - 1);
                    const bool st1 = bp + 1 <= NP && rows_done >= 1, st2 = bp + 5 <= NP && rows_done >= 2;
                    if (st2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads + 4) : "memory");
                    else if (st1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads + 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoads) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // Operand reads as explicit instructions: a ds_read the compiler can see behind LDS-DMA into the same ring makes it
                // insert its own s_waitcnt vmcnt(0) (it cannot see the counted waits above) -- a full drain of the stream per row.
                v4i T[4], W[4];
                {
                    const u32 ta = (u32)(uintptr_t)&ring[w][slot][g >> 1][2 * n + (g & 1)]; // + 512 per output tile
                    const u32 wa = (u32)(uintptr_t)&wl[bp][0][lane];                        // + 1024 per column tile
                    asm volatile(
                        "ds_read_b128 %0, %8\n\t"
                        "ds_read_b128 %1, %8 offset:512\n\t"
                        "ds_read_b128 %2, %8 offset:1024\n\t"
                        "ds_read_b128 %3, %8 offset:1536\n\t"
                        "ds_read_b128 %4, %9\n\t"
                        "ds_read_b128 %5, %9 offset:1024\n\t"
                        "ds_read_b128 %6, %9 offset:2048\n\t"
                        "ds_read_b128 %7, %9 offset:3072\n\t"
                        "s_waitcnt lgkmcnt(0)" // operands in registers before the slot is refilled
                        : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3])
                        : "v"(ta), "v"(wa)
                        : "memory");
#pragma unroll
                    for (int o4 = 0; o4 < 4; ++o4) T[o4] ^= v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u}; // bytes as a - 128
                }
                if (s + NP < nsteps) issue(s + NP, slot);
                slot = slot + 1 == NP ? 0 : slot + 1;
                // The 16 products of the step, written out: with more than 256 registers in use the compiler gives every matrix
                // instruction an accumulator-file destination, and that half of the file is exactly the 256 registers of the cross
                // sums -- these 64 columns must live in the vector half, which only an explicit register class gets.  The compiler
                // cannot see hazards inside: vector write -> matrix read is covered by the leading s_nop, matrix write -> vector read
                // (the pack after the row's last step) by the trailing ones; accumulate chains (same destination) need none.  The
                // row's first step starts its columns from the inline constant 0 (no 64 register initialisations per row); the
                // columns are signed, the 2^27 that makes each non-negative is added when they are packed.
                if (bp == 0)
                    asm volatile("s_nop 3\n\t"
                                 "v_mfma_i32_16x16x64_i8 %0, %16, %20, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %1, %16, %21, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %2, %16, %22, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %3, %16, %23, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %4, %17, %20, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %5, %17, %21, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %6, %17, %22, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %7, %17, %23, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %8, %18, %20, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %9, %18, %21, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %10, %18, %22, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %11, %18, %23, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %12, %19, %20, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %13, %19, %21, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %14, %19, %22, 0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %15, %19, %23, 0"
                                 : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fa[0][2]), "=&v"(fa[0][3]), "=&v"(fa[1][0]), "=&v"(fa[1][1]), "=&v"(fa[1][2]),
                                   "=&v"(fa[1][3]), "=&v"(fa[2][0]), "=&v"(fa[2][1]), "=&v"(fa[2][2]), "=&v"(fa[2][3]), "=&v"(fa[3][0]), "=&v"(fa[3][1]),
                                   "=&v"(fa[3][2]), "=&v"(fa[3][3])
                                 : "v"(W[0]), "v"(W[1]), "v"(W[2]), "v"(W[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]));
                else if (bp < 3)
                    asm volatile("s_nop 3\n\t"
                                 "v_mfma_i32_16x16x64_i8 %0, %16, %20, %0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %1, %16, %21, %1\n\t"
                                 "v_mfma_i32_16x16x64_i8 %2, %16, %22, %2\n\t"
                                 "v_mfma_i32_16x16x64_i8 %3, %16, %23, %3\n\t"
                                 "v_mfma_i32_16x16x64_i8 %4, %17, %20, %4\n\t"
                                 "v_mfma_i32_16x16x64_i8 %5, %17, %21, %5\n\t"
                                 "v_mfma_i32_16x16x64_i8 %6, %17, %22, %6\n\t"
                                 "v_mfma_i32_16x16x64_i8 %7, %17, %23, %7\n\t"
                                 "v_mfma_i32_16x16x64_i8 %8, %18, %20, %8\n\t"
                                 "v_mfma_i32_16x16x64_i8 %9, %18, %21, %9\n\t"
                                 "v_mfma_i32_16x16x64_i8 %10, %18, %22, %10\n\t"
                                 "v_mfma_i32_16x16x64_i8 %11, %18, %23, %11\n\t"
                                 "v_mfma_i32_16x16x64_i8 %12, %19, %20, %12\n\t"
                                 "v_mfma_i32_16x16x64_i8 %13, %19, %21, %13\n\t"
                                 "v_mfma_i32_16x16x64_i8 %14, %19, %22, %14\n\t"
                                 "v_mfma_i32_16x16x64_i8 %15, %19, %23, %15"
                                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]),
                                   "+v"(fa[1][3]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3]), "+v"(fa[3][0]), "+v"(fa[3][1]),
                                   "+v"(fa[3][2]), "+v"(fa[3][3])
                                 : "v"(W[0]), "v"(W[1]), "v"(W[2]), "v"(W[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]));
                else
                    asm volatile("s_nop 3\n\t"
                                 "v_mfma_i32_16x16x64_i8 %0, %16, %20, %0\n\t"
                                 "v_mfma_i32_16x16x64_i8 %1, %16, %21, %1\n\t"
                                 "v_mfma_i32_16x16x64_i8 %2, %16, %22, %2\n\t"
                                 "v_mfma_i32_16x16x64_i8 %3, %16, %23, %3\n\t"
                                 "v_mfma_i32_16x16x64_i8 %4, %17, %20, %4\n\t"
                                 "v_mfma_i32_16x16x64_i8 %5, %17, %21, %5\n\t"
                                 "v_mfma_i32_16x16x64_i8 %6, %17, %22, %6\n\t"
                                 "v_mfma_i32_16x16x64_i8 %7, %17, %23, %7\n\t"
                                 "v_mfma_i32_16x16x64_i8 %8, %18, %20, %8\n\t"
                                 "v_mfma_i32_16x16x64_i8 %9, %18, %21, %9\n\t"
                                 "v_mfma_i32_16x16x64_i8 %10, %18, %22, %10\n\t"
                                 "v_mfma_i32_16x16x64_i8 %11, %18, %23, %11\n\t"
                                 "v_mfma_i32_16x16x64_i8 %12, %19, %20, %12\n\t"
                                 "v_mfma_i32_16x16x64_i8 %13, %19, %21, %13\n\t"
                                 "v_mfma_i32_16x16x64_i8 %14, %19, %22, %14\n\t"
                                 "v_mfma_i32_16x16x64_i8 %15, %19, %23, %15\n\t"
                                 "s_nop 15\n\t"
                                 "s_nop 7"
                                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]),
                                   "+v"(fa[1][3]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[2][2]), "+v"(fa[2][3]), "+v"(fa[3][0]), "+v"(fa[3][1]),
                                   "+v"(fa[3][2]), "+v"(fa[3][3])
                                 : "v"(W[0]), "v"(W[1]), "v"(W[2]), "v"(W[3]), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]));
            }
            const fq o = mfma_fold_finish(fa, kfix, c1, c8, c16, c24);
            fq* out = dst + (e0 + (size_t)rq) * qn + base0 + (size_t)t * stride + lane;
            if (BIG) fq_store_stream(out, o); else fq_store(out, o);
            if (rq == 0 && t > 0) { // every wave has finished the matrix stage of tile t - 1: the tile buffer is free
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            tile[4 * w + rq][2 * lane] = u32x4{o.v[0] ^ 0x80808080u, o.v[1] ^ 0x80808080u, o.v[2] ^ 0x80808080u, o.v[3] ^ 0x80808080u};
            tile[4 * w + rq][2 * lane + 1] = u32x4{o.v[4] ^ 0x80808080u, o.v[5] ^ 0x80808080u, o.v[6] ^ 0x80808080u, o.v[7] ^ 0x80808080u};
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) pls[w][rq][k][lane] = __builtin_addc(pls[w][rq][k][lane], o.v[k], c, &c);
            pls[w][rq][8][lane] += c;
            __builtin_amdgcn_sched_barrier(0); // rows are unrolled for the static pl index only: no scheduling across rows (register pressure)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's four tile rows are written
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        matrix_stage();
    }
    // ---- epilogue: once per wave (as k_cross64_mfma)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); // every wave is done with the tile: it becomes scratch (4 KiB per wave)
    int* mat = reinterpret_cast<int*>(&tile[0][0]) + w * 1024;
    const fq r2 = {FQ_R2_LIMBS};
    fq* out = partials + ((size_t)term * gridDim.x + blockIdx.x) * (NE * NE + 2 * NE);
    fq wk;
#pragma unroll
    for (int k = 0; k < 8; ++k) wk.v[k] = kPow256ModP[lane < 63 ? lane : 0][k];
    for (int e = 0; e < 16; ++e) { // rolled: the body is large and runs once
        v16i a = acc[0];
#pragma unroll
        for (int e2 = 1; e2 < 16; ++e2)
            if (e == e2) a = acc[e2];
#pragma unroll
        for (int k = 0; k < 16; ++k) mat[((k & 3) + 8 * (k >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = a[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the anti-diagonal i + j = lane of the 32 x 32 matrix (lane 63 has none).  Fixed trip count, in-range addresses, the bounds
        // as a select: the 32 LDS reads issue back to back (a loop over [klo, khi] waited for every read: ~20 us per launch)
        long long d = 0;
#pragma unroll 8 // eight reads in flight: a full unroll costs this kernel (at its 512-register limit) two more spilled registers
        for (int i = 0; i < 32; ++i) {
            const int j = lane - i, v = mat[i * 32 + (j & 31)];
            d += (j >= 0 && j < 32) ? v : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (raw) { // as in k_cross64_mfma
            raw[((((size_t)term * gridDim.x + blockIdx.x) * (NE * NE)) + (size_t)(NE * (4 * fi + (e >> 2)) + 4 * gi + (e & 3))) * 64 + lane] = d;
            continue;
        }
        const bool neg = d < 0;
        const unsigned long long ad = (unsigned long long)(neg ? -d : d);
        fq xx = fq_zero();
        xx.v[0] = (u32)ad;
        xx.v[1] = (u32)(ad >> 32);
        fq prod = fq_mul(xx, wk);
        if (neg) prod = fq_sub(fq_zero(), prod);
        const fq mine = fq_wave_sum(prod);
        if (lane == 0) fq_store(out + NE * (4 * fi + (e >> 2)) + 4 * gi + (e & 3), mine);
    }
    for (int row = 0; row < 4; ++row) { // rolled; tile rows 4w + row: rows 0..7 = A_e, rows 8..15 = B_e'
        fq lo, hi = fq_zero();
#pragma unroll
        for (int k = 0; k < 8; ++k) lo.v[k] = pls[w][row][k][lane];
        hi.v[0] = pls[w][row][8][lane];
        const fq sum = fq_wave_sum(fq_add(fq_reduce_256(lo), fq_mul(hi, r2)));
        if (lane == 0) fq_store(out + NE * NE + 4 * w + row, sum);
    }
}

// second stage: one block per (term, value): red[term * nv + v] = sum over the workgroups (fixed order)
__global__ void __launch_bounds__(kThreads) k_reduce_cross(const fq* __restrict__ partials, int gx, int nv, fq* __restrict__ red) {
    const int term = blockIdx.x / nv, v = blockIdx.x % nv;
    fq acc[1] = {fq_zero()};
    for (int b = threadIdx.x; b < gx; b += kThreads) acc[0] = fq_add(acc[0], fq_load(partials + ((size_t)term * gx + b) * nv + v));
    block_sum_store<1>(acc, red);
}
// The same second stage when the sweeps leave the 64 cross sums of a workgroup as RAW digits: raw[((term * gx + b) * 64 + v) * 64 + k] =
// the signed integer D_k of  P_signed = sum_k D_k 256^k  (k < 63; |D_k| < 2^37 per workgroup).  Integers add exactly, so the digits
// are summed over the workgroups first and ONE product per digit, 256^k 2^-256 mod p, replaces one per digit and workgroup in the
// sweeps' epilogues.  Values 64..79 (the plain sums) as in k_reduce_cross.
__global__ void __launch_bounds__(kThreads) k_reduce_cross_raw(const long long* __restrict__ raw, const fq* __restrict__ partials, int gx,
                                                               fq* __restrict__ red) {
    const int term = blockIdx.x / 80, v = blockIdx.x % 80;
    if (v >= 64) { // uniform per block
        fq acc[1] = {fq_zero()};
        for (int b = threadIdx.x; b < gx; b += kThreads) acc[0] = fq_add(acc[0], fq_load(partials + ((size_t)term * gx + b) * 80 + v));
        block_sum_store<1>(acc, red);
        return;
    }
    __shared__ long long sh[kThreads / 64][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    long long s = 0;
    const long long* __restrict__ col = raw + ((size_t)term * gx * 64 + v) * 64 + lane; // workgroup b: + b * 4096
    int b = slice;
    for (; b + 7 * (kThreads / 64) < gx; b += 8 * (kThreads / 64)) { // eight loads in flight per lane
        long long x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = col[(size_t)(b + u * (kThreads / 64)) * 4096];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
    }
    for (; b < gx; b += kThreads / 64) s += col[(size_t)b * 4096];
    sh[slice][lane] = s;
    __syncthreads();
    if (slice != 0) return;
    long long d = 0;
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) d += sh[k][lane];
    fq wk;
#pragma unroll
    for (int k = 0; k < 8; ++k) wk.v[k] = kPow256ModP[lane < 63 ? lane : 0][k];
    const bool neg = d < 0;
    const unsigned long long ad = (unsigned long long)(neg ? -d : d);
    fq x = fq_zero();
    x.v[0] = (u32)ad;
    x.v[1] = (u32)(ad >> 32);
    fq prod = fq_mul(x, wk);
    if (neg) prod = fq_sub(fq_zero(), prod);
    const fq mine = fq_wave_sum(prod);
    if (lane == 0) fq_store(red + blockIdx.x, mine);
}
// out[term * 64 + 8 e + e'] = P[e][e'] = P_signed + 128 c1 (A_e + B_e') 2^-256 - 2^14 c1^2 N 2^-256 (see k_sum16_mfma); one block of 128
__global__ void __launch_bounds__(128) k_cross64_fix(const fq* __restrict__ red, fq qn, fq* __restrict__ out) {
    const int t = threadIdx.x, term = t >> 6, e = (t >> 3) & 7, e2 = t & 7;
    const fq k1 = {FQ_SUM16_K1}, k2 = {FQ_SUM16_K2};
    const fq* in = red + term * 80;
    const fq ab = fq_add(fq_load(in + 64 + e), fq_load(in + 72 + e2));
    fq_store(out + t, fq_add(fq_add(fq_load(in + 8 * e + e2), fq_mul(k1, ab)), fq_mul(k2, qn)));
}

// U (in[3 b + a], see k_sum9_lazy) -> the nine sums S[3Y + t].  One block of 64 threads.
__global__ void __launch_bounds__(64) k_sum9_from_u(const fq* __restrict__ in, fq* __restrict__ out) {
    const int s = threadIdx.x;
    if (s >= 9) return;
    const int Y = s / 3, t = s % 3;
    auto V = [&](int a) -> fq { // V[a][Y]
        if (Y < 2) return fq_load(in + 3 * Y + a);
        const fq u0 = fq_load(in + a), u1 = fq_load(in + 3 + a), up = fq_load(in + 6 + a);
        const fq u1x2 = fq_dbl(u1);
        const fq acc = fq_add(fq_add(fq_dbl(u0), u0), fq_add(fq_dbl(u1x2), u1x2)); // 3 u0 + 6 u1
        return fq_sub(acc, fq_dbl(up));
    };
    fq r;
    if (t < 2) r = V(t);
    else r = fq_sub(fq_dbl(fq_add(V(0), V(1))), V(2));
    fq_store(out + s, r);
}

// second stage: block s = 3Y + t adds its sum's partials over both terms and all bx (fixed order)
__global__ void __launch_bounds__(kThreads) k_reduce9(const fq* __restrict__ partials, int gx, int ysplit, fq* __restrict__ out) {
    const int s = blockIdx.x, Y = s / 3, t = s % 3;
    fq acc[1] = {fq_zero()};
    for (int i = threadIdx.x; i < 2 * gx; i += kThreads) {
        const int term = i / gx, bx = i % gx;
        const size_t at = ysplit ? ((size_t)(term * 3 + Y) * gx + bx) * 3 + t : ((size_t)term * gx + bx) * 9 + s;
        acc[0] = fq_add(acc[0], fq_load(partials + at));
    }
    block_sum_store<1>(acc, out);
}

// Bind the two top variables of all four tables at once: weights W = eq((r, s), b) * 2^280, b = 2 b1 + b2; one lane per
// (table = blockIdx.y, output o): dst[o] = sum_b W[b] * src[b * outlen + o], one Montgomery reduction per output.
__global__ void __launch_bounds__(kThreads) k_fold2_tables(Tab4 S, Tab4w D, size_t outlen, W28<2> W) {
    const fq* __restrict__ src = S.t[blockIdx.y];
    fq* __restrict__ dst = D.t[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x; o < outlen; o += stride) {
        const fq in0 = fq_load_stream(src + o), in1 = fq_load_stream(src + outlen + o), in2 = fq_load_stream(src + 2 * outlen + o),
                 in3 = fq_load_stream(src + 3 * outlen + o);
        u64 a28[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) a28[k] = 0;
        fq28_mac(a28, in0, W.w[0]);
        fq28_mac(a28, in1, W.w[1]);
        fq28_mac(a28, in2, W.w[2]);
        fq28_mac(a28, in3, W.w[3]);
        fq_store(dst + o, fq28_redc(a28));
    }
}

// the same for three variables (eight weights): small tables, one launch for all four
// gather (may be null): the folded tables ALSO go there, table after table (gather[t * outlen + o]) -- the device-mapped pinned buffer
// when the host takes over after this fold, instead of one more launch (k_gather4)
__global__ void __launch_bounds__(kThreads) k_fold3_tables(Tab4 S, Tab4w D, size_t outlen, W28<3> W, fq* __restrict__ gather) {
    const fq* __restrict__ src = S.t[blockIdx.y];
    fq* __restrict__ dst = D.t[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x; o < outlen; o += stride) {
        u64 a28[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) a28[k] = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) fq28_mac(a28, fq_load(src + (size_t)b * outlen + o), W.w[b]);
        const fq v = fq28_redc(a28);
        fq_store(dst + o, v);
        if (gather) fq_store(gather + (size_t)blockIdx.y * outlen + o, v);
    }
}

// Small tables (GKR layers: <= 2^20 entries per table, L2-resident): a pass is bound by launch gaps and by each lane's serial chain,
// not by bytes, so the two-variable fold and the nine sums of its result are ONE kernel.  A wave serves 16 next-level quarter-indices
// j' of one term with FOUR lanes each: lane (e, jj) folds entry e q' + j' of both factor tables (8 loads, 8 unreduced products, 2
// reductions), stores them, and the four lanes of a j' exchange their folded values through LDS (wave-private: program order, no
// barrier).  Lane e < 3 then forms the three products of Y = e (operands picked by select, not by branch: the wave runs one code
// path); lane e = 3 contributes zero.  Chain per lane ~1900 vector instructions instead of ~500 + ~1000 in two launches with a
// second-stage launch between.  partials: [(term * gridDim.x + bx) * 9 + 3Y + t] (k_reduce9, ysplit = 0).
__global__ void __launch_bounds__(kThreads) k_fold2_sum9(Tab4 S, Tab4w D, size_t outlen, W28<2> W, fq* __restrict__ partials) {
    __shared__ u32x4 xch[kThreads / 64][2][4][16][2]; // [wave][factor][e][jj][half]: 4 KiB per wave
    __shared__ fq red[kThreads / 64][9];
    const int term = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, jj = lane & 15;
    const fq* __restrict__ F = S.t[2 * term];
    const fq* __restrict__ G = S.t[2 * term + 1];
    fq* __restrict__ DF = D.t[2 * term];
    fq* __restrict__ DG = D.t[2 * term + 1];
    const size_t qn = outlen >> 2; // next-level quarter length
    fq acc[3] = {fq_zero(), fq_zero(), fq_zero()};
    const size_t stride = (size_t)gridDim.x * 64; // j' per grid step
    for (size_t j0 = (size_t)blockIdx.x * 64 + (size_t)wave * 16; j0 < qn; j0 += stride) { // wave-uniform
        const size_t j = j0 + jj;
        const bool live = j < qn;
        const size_t o = (size_t)e * qn + (live ? j : 0);
        fq f[4], g[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f[b] = fq_load(F + (size_t)b * outlen + o);
            g[b] = fq_load(G + (size_t)b * outlen + o);
        }
        u64 a28[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) a28[k] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) fq28_mac(a28, f[b], W.w[b]);
        const fq fo = fq28_redc(a28);
#pragma unroll
        for (int k = 0; k < 20; ++k) a28[k] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) fq28_mac(a28, g[b], W.w[b]);
        const fq go = fq28_redc(a28);
        if (live) {
            fq_store(DF + o, fo);
            fq_store(DG + o, go);
        }
        xch[wave][0][e][jj][0] = u32x4{fo.v[0], fo.v[1], fo.v[2], fo.v[3]};
        xch[wave][0][e][jj][1] = u32x4{fo.v[4], fo.v[5], fo.v[6], fo.v[7]};
        xch[wave][1][e][jj][0] = u32x4{go.v[0], go.v[1], go.v[2], go.v[3]};
        xch[wave][1][e][jj][1] = u32x4{go.v[4], go.v[5], go.v[6], go.v[7]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private exchange: every lane of the wave has written
        // operands of Y = e: the b2 = 0 entries (e'' = 0, 2) for Y = 0, the b2 = 1 entries (1, 3) for Y = 1, 2 x[.][1] - x[.][0] for Y = 2
        fq v0[2], v1[2];
#pragma unroll
        for (int fac = 0; fac < 2; ++fac) {
            fq x[4];
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const u32x4 lo = xch[wave][fac][ee][jj][0], hi = xch[wave][fac][ee][jj][1];
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[ee].v[k] = lo[k]; x[ee].v[4 + k] = hi[k]; }
            }
            const fq y20 = fq_sub(fq_dbl(x[1]), x[0]), y21 = fq_sub(fq_dbl(x[3]), x[2]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v0[fac].v[k] = e == 0 ? x[0].v[k] : (e == 1 ? x[1].v[k] : y20.v[k]);
                v1[fac].v[k] = e == 0 ? x[2].v[k] : (e == 1 ? x[3].v[k] : y21.v[k]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // values in registers before the next trip overwrites the exchange space
        const fq d0 = fq_sub(v1[0], v0[0]), d1 = fq_sub(v1[1], v0[1]);
        const fq p0 = fq_mul(v0[0], v0[1]), p1 = fq_mul(v1[0], v1[1]), p2 = fq_mul(d0, d1);
        if (live && e < 3) {
            acc[0] = fq_add(acc[0], p0);
            acc[1] = fq_add(acc[1], p1);
            acc[2] = fq_add(acc[2], p2);
        }
    }
    // sums over the 16 lanes of each e, then over the waves
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        fq a = acc[t];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) a = fq_add(a, fq_shfl_down(a, off));
        if (jj == 0 && e < 3) red[wave][3 * e + t] = a;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        fq t = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t = fq_add(t, red[w][threadIdx.x]);
        fq_store(&partials[((size_t)term * gridDim.x + blockIdx.x) * 9 + threadIdx.x], t);
    }
}

// ---- three rounds per pass on SMALL tables (GKR layers) ---------------------------------------------------------------------
// A pass on a table of <= 2^16 entries costs its host round trip (~20 us) and each lane's serial chain, so it should serve as many
// rounds as one chain allows.  For a product term F G and the three top variables, write the value of F "at" a in {0, 1, oo} along
// a variable as F(0), F(1), F(1) - F(0) (the coefficient of the variable).  The 27 sums
//   T[9a + 3b + c] = sum over both terms and all j of F[a,b,c; j] * G[a,b,c; j]
// determine the next THREE round polynomials: per variable P(X) = P[0] + X (P[1] - P[0] - P[oo]) + X^2 P[oo], applied axis by axis
// (host_round_triple_from_sum27).  The kernel folds M (2 or 3) variables of all four tables and forms the 27 sums of the RESULT:
// 32 lanes per (term, octet-index j''): lane l < 16 folds entry e q'' + j'' (e = l & 7) of factor l >> 3 (2^M loads, 2^M unreduced
// products, one reduction), stores it and puts it at slot 9 b1 + 3 b2 + b3 of the octet's 27-slot value table in wave-private LDS;
// three short steps fill the table's oo slots axis by axis (4, 6, 9 differences per factor, one per lane: 19 subtractions per factor
// instead of 14 per CLASS); lane l < 27 then reads its class's two operands and multiplies ONCE.  Chain per lane: the fold, three
// subtractions with their LDS round trips, one product -- shorter than k_fold2_sum9's, for three rounds instead of two.
// partials: [(term * gridDim.x + bx) * 27 + 9a + 3b + c].
struct Val27 { u32x4 h[2]; };
__device__ __forceinline__ fq val27_get(const Val27* p) {
    const u32x4 lo = p->h[0], hi = p->h[1];
    fq x;
#pragma unroll
    for (int k = 0; k < 4; ++k) { x.v[k] = lo[k]; x.v[4 + k] = hi[k]; }
    return x;
}
__device__ __forceinline__ void val27_put(Val27* p, const fq& x) {
    p->h[0] = u32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
    p->h[1] = u32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
}
// M = 0: no fold -- the 27 sums of the tables as they are (the first pass of a prove; `outlen` is the tables' length, D unused).
// PWQ: the relation has the shape P W + Q * 1 (a product term plus a LINEAR term: both phases of a GKR layer, gkr.cpp): tables
// S.t[0] = P, S.t[1] = W, S.t[2] = Q, the fourth factor is the constant one and is neither read, folded nor written (any prefix of a
// table of ones is its own fold).  One workgroup population instead of two (gridDim.y = 1): lanes 16 .. 23 of a group fold Q's eight
// entries beside the sixteen of P and W (lanes that idle in the general form), and a class without an oo digit adds Q's corner to its
// product -- the constant one is 1 at 0 and 1 and 0 at oo, so the other classes get nothing from the linear term.  Half the wave
// instructions and three quarters of the bytes of the general form for the same 27 sums.
template <int M, bool PWQ>
__global__ void __launch_bounds__(kThreads) k_fold_sum27(Tab4 S, Tab4w D, size_t outlen, W28<M> W, fq* __restrict__ partials) {
    constexpr int NV = PWQ ? 3 : 2;
    __shared__ Val27 val[kThreads / 64][2][NV][27]; // [wave][octet of the wave][factor (2: Q)][9a + 3b + c]
    __shared__ fq red[kThreads / 64][27];
    const int term = PWQ ? 0 : (int)blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o2 = lane >> 5, l = lane & 31;
    const int role = l >> 3, e = l & 7; // fold role: the entry e of factor 0 / 1 (role 2, PWQ: of Q)
    const bool folds = l < 8 * NV;
    const int ti = PWQ ? (folds ? role : 0) : 2 * term + (role & 1);
    const fq* __restrict__ src = S.t[ti];
    fq* __restrict__ dst = D.t[ti];
    // difference roles: step 1 (third variable, l < 8): pair (b1, b2) = l & 3 of factor l >> 2; step 2 (second variable, l < 12):
    // (b1, c) = (l % 6) / 3, l % 3 of factor l / 6; step 3 (first variable, l < 18): slot 3b + c = l % 9 of factor l / 9
    Val27* const v1 = &val[wave][o2][(l >> 2) & 1][9 * ((l >> 1) & 1) + 3 * (l & 1)];
    Val27* const v2 = &val[wave][o2][l < 6 ? 0 : 1][9 * ((l % 6) / 3) + l % 3];
    Val27* const v3 = &val[wave][o2][l < 9 ? 0 : 1][l % 9];
    const int cls = l < 27 ? l : 0;
    const bool corner = cls / 9 != 2 && (cls / 3) % 3 != 2 && cls % 3 != 2; // no oo digit: the linear term's classes
    const size_t qn = outlen >> 3; // octet length of the folded tables
    fq acc = fq_zero();
    const size_t stride = (size_t)gridDim.x * (kThreads / 32); // octet-indices per grid step
    for (size_t j0 = (size_t)blockIdx.x * (kThreads / 32) + (size_t)wave * 2; j0 < qn; j0 += stride) { // wave-uniform
        const size_t j = j0 + o2;
        const bool live = j < qn;
        if (folds) {
            const size_t o = (size_t)e * qn + (live ? j : 0);
            fq v;
            if constexpr (M == 0) {
                v = fq_load(src + o);
            } else {
                u64 a28[20];
#pragma unroll
                for (int k = 0; k < 20; ++k) a28[k] = 0;
#pragma unroll
                for (int b = 0; b < (1 << M); ++b) fq28_mac(a28, fq_load(src + (size_t)b * outlen + o), W.w[b]);
                v = fq28_redc(a28);
                if (live) fq_store(dst + o, v);
            }
            val27_put(&val[wave][o2][role][9 * (e >> 2) + 3 * ((e >> 1) & 1) + (e & 1)], v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private table: program order, no barrier
        if (l < 8) val27_put(v1 + 2, fq_sub(val27_get(v1 + 1), val27_get(v1)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (l < 12) val27_put(v2 + 6, fq_sub(val27_get(v2 + 3), val27_get(v2)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (l < 18) val27_put(v3 + 18, fq_sub(val27_get(v3 + 9), val27_get(v3)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fq pr = fq_mul(val27_get(&val[wave][o2][0][cls]), val27_get(&val[wave][o2][1][cls]));
        if constexpr (PWQ) {
            const fq qv = val27_get(&val[wave][o2][NV - 1][corner ? cls : 0]); // slot 0 is a corner: always written
            if (corner) pr = fq_add(pr, qv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // operands in registers before the next trip overwrites the table
        if (live && l < 27) acc = fq_add(acc, pr);
    }
    acc = fq_add(acc, fq_shfl_down(acc, 32)); // the wave's two octets hold the same classes
    if (lane < 27) red[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 27) {
        fq t = red[0][threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < kThreads / 64; ++wv) t = fq_add(t, red[wv][threadIdx.x]);
        fq_store(&partials[((size_t)term * gridDim.x + blockIdx.x) * 27 + threadIdx.x], t);
    }
}
// second stage: block s adds sum s of `groups` workgroups (fixed order)
__global__ void __launch_bounds__(kThreads) k_reduce27(const fq* __restrict__ partials, int groups, fq* __restrict__ out) {
    fq acc[1] = {fq_zero()};
    for (int i = threadIdx.x; i < groups; i += kThreads) acc[0] = fq_add(acc[0], fq_load(partials + (size_t)i * 27 + blockIdx.x));
    block_sum_store<1>(acc, out);
}

// the host tail of the degree-2 prover: four small tables (n entries each) into one contiguous buffer (table-major),
// normally the device-mapped pinned staging buffer -- one launch instead of four copy commands
__global__ void __launch_bounds__(kThreads) k_gather4(Tab4 T, size_t n, fq* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < 4 * n) fq_store(out + i, fq_load(T.t[i / n] + i % n));
}

// ---- launchers ---------------------------------------------------------------------------
int launch_eval3(const void* const t[4], size_t half, void* partials, hipStream_t st) {
    Tab4 T;
    for (int k = 0; k < 4; ++k) T.t[k] = (const fq*)t[k];
    int g = grid_for(half);
    hipLaunchKernelGGL(k_eval3, dim3(g), dim3(kThreads), 0, st, T, half, (fq*)partials);
    return g;
}
int launch_fold4_eval3(const void* const s[4], void* const d[4], size_t q, const uint64_t r[4], void* partials, hipStream_t st) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    int g = grid_for(q);
    hipLaunchKernelGGL(k_fold4_eval3, dim3(g), dim3(kThreads), 0, st, S, D, q, to_fq(r), (fq*)partials);
    return g;
}
void launch_fold4(const void* const s[4], void* const d[4], size_t half, const uint64_t r[4], hipStream_t st) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    hipLaunchKernelGGL(k_fold4, dim3(grid_for(half)), dim3(kThreads), 0, st, S, D, half, to_fq(r));
}
int launch_sum9(const void* const t[4], size_t q, const uint64_t fix24[4], int lazy_from_log, void* partials, void* out, hipStream_t st,
                const DmaGates& gates, bool allow_u, bool pwq) {
    Tab4 T;
    for (int k = 0; k < 4; ++k) T.t[k] = (const fq*)t[k];
    if (gates.sum9_mfma_from_log < 63 && q % 64 == 0 && q >= ((size_t)1 << std::max(gates.sum9_mfma_from_log, 0)) &&
        q <= ((size_t)kMaxBlocks9 << 16)) { // the matrix cores
        size_t gx = q / 64;
        if (gx > (size_t)kMaxBlocks9) gx = kMaxBlocks9;
        // i32 accumulators hold 2^17 quarter-indices; kMaxBlocks9 workgroups cover q <= 2^26 with 2^16 each (beyond: launch_sum9's caller
        // would need a larger partials buffer -- the reduced-accumulation kernel takes over)
        fq* red = (fq*)partials + 2 * gx * 24; // 48 reduced values behind the partials (kSum9PartialElems)
        fq qn{};
        qn.v[0] = (u32)q;
        qn.v[1] = (u32)((u64)q >> 32);
        g_forms |= kFormSum9Mfma;
        hipLaunchKernelGGL(k_sum16_mfma, dim3((unsigned)gx, 2), dim3(kThreads), 0, st, T, q, (fq*)partials);
        hipLaunchKernelGGL(k_reduce48, dim3(48), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, red);
        hipLaunchKernelGGL(k_sum9_from_p16, dim3(1), dim3(64), 0, st, (const fq*)red, qn, (fq*)out);
        return 0;
    }
    if (lazy_from_log >= 0 && lazy_from_log < 63 && q >= 64 && q >= ((size_t)1 << lazy_from_log)) { // large: unreduced accumulation, one wave per Y
        size_t gx = (q + 63) / 64;
        // 256 workgroups per term (two per CU): this kernel serves the L2-resident sizes (GKR layers: 2^18 quarter-entries), where fewer,
        // longer-running workgroups beat 1024 short ones (GKR config-5 layers 13.4 -> 12.6 ms, 4 x 2^20 prove 0.389 -> 0.358 ms)
        // (the same cap for the product + linear form: 512 workgroups per term measured 10.3 instead of 10.0 ms of GKR layers)
        if (gx > 256) gx = 256;
        fq* u = (fq*)partials + (size_t)18 * gx; // nine U sums, behind the partials
        g_forms |= kFormSum9Lazy;
        if (pwq) {
            g_forms |= kFormPwq;
            hipLaunchKernelGGL(k_sum9_lazy<true>, dim3((unsigned)gx, 2), dim3(64, 9), 0, st, T, q, to_fq(fix24), (fq*)partials);
        } else {
            hipLaunchKernelGGL(k_sum9_lazy<false>, dim3((unsigned)gx, 2), dim3(64, 9), 0, st, T, q, to_fq(fix24), (fq*)partials);
        }
        if (allow_u) { // the caller maps U to the nine sums on the host (a few additions): one launch less on the critical path
            hipLaunchKernelGGL(k_reduce9, dim3(9), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 1, (fq*)out);
            return 1;
        }
        hipLaunchKernelGGL(k_reduce9, dim3(9), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 1, u);
        hipLaunchKernelGGL(k_sum9_from_u, dim3(1), dim3(64), 0, st, (const fq*)u, (fq*)out);
        return 0;
    }
    size_t gx = (q + kThreads - 1) / kThreads;
    if (gx < 1) gx = 1;
    if (gx > (size_t)kMaxBlocks9) gx = kMaxBlocks9;
    g_forms |= kFormSum9;
    const bool ysplit = q < ((size_t)1 << 21); // small tables: three times the lanes, tables read twice (mostly cache-resident)
    if (ysplit) hipLaunchKernelGGL((k_sum9<true>), dim3((unsigned)gx, 6), dim3(kThreads), 0, st, T, q, (fq*)partials);
    else hipLaunchKernelGGL((k_sum9<false>), dim3((unsigned)gx, 2), dim3(kThreads), 0, st, T, q, (fq*)partials);
    hipLaunchKernelGGL(k_reduce9, dim3(9), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, ysplit ? 1 : 0, (fq*)out);
    return 0;
}
// a workgroup's i32 accumulators take 2^16 quarter-indices: |byte - 128| <= 128, so a product is <= 2^14 and 2^16 of them stay below
// 2^31 with a factor of two to spare (2^17 all-zero entries would reach exactly 2^31)
bool cross64_fits(size_t q) { return q >= 64 && q % 64 == 0 && q <= ((size_t)kMaxBlocks9 << 16); }
// one workgroup per CU and term half: 128 x 2 persistent workgroups; more only to keep a workgroup's share within the i32
// accumulators (2^16 quarter-indices)
static size_t cross64_grid(size_t q) {
    size_t gx = std::min<size_t>(q / 64, 128);
    const size_t need = (q + ((size_t)1 << 16) - 1) >> 16;
    return gx < need ? need : gx;
}
size_t cross64_raw_bytes(size_t q) { return 2 * cross64_grid(q) * 64 * 64 * sizeof(long long); }
void launch_cross64(const void* const t[4], size_t q, int ring, void* partials, void* raw, void* out, hipStream_t st) {
    Tab4 T;
    for (int k = 0; k < 4; ++k) T.t[k] = (const fq*)t[k];
    const size_t gx = cross64_grid(q);
    fq* red = (fq*)partials + 2 * gx * 80; // 160 reduced values behind the partials (kSum9PartialElems)
    fq qn{};
    qn.v[0] = (u32)q;
    qn.v[1] = (u32)((u64)q >> 32);
    g_forms |= kFormCross64;
    if (ring == 5) hipLaunchKernelGGL((k_cross64_mfma<5>), dim3((unsigned)gx, 2), dim3(512), 0, st, T, q, (fq*)partials, (long long*)raw);
    else if (ring == 4) hipLaunchKernelGGL((k_cross64_mfma<4>), dim3((unsigned)gx, 2), dim3(512), 0, st, T, q, (fq*)partials, (long long*)raw);
    else hipLaunchKernelGGL((k_cross64_mfma<3>), dim3((unsigned)gx, 2), dim3(512), 0, st, T, q, (fq*)partials, (long long*)raw);
    if (raw) hipLaunchKernelGGL(k_reduce_cross_raw, dim3(160), dim3(kThreads), 0, st, (const long long*)raw, (const fq*)partials, (int)gx, red);
    else hipLaunchKernelGGL(k_reduce_cross, dim3(160), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 80, red);
    hipLaunchKernelGGL(k_cross64_fix, dim3(1), dim3(128), 0, st, (const fq*)red, qn, (fq*)out);
}
// launch_fold3_tables followed by launch_cross64 on its result, as one sweep (needs cross64_fits(outlen / 8)); ring = per-wave DMA slots
void launch_fold3_cross64(const void* const s[4], void* const d[4], size_t outlen, const uint64_t* r3, const uint64_t two24[4], void* wbuf,
                          int ring, void* partials, void* raw, void* out, hipStream_t st, const DmaGates& gates) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    const size_t q = outlen / 8;
    const size_t gx = cross64_grid(q);
    fq* red = (fq*)partials + 2 * gx * 80;
    fq qn{};
    qn.v[0] = (u32)q;
    qn.v[1] = (u32)((u64)q >> 32);
    const bool stream_stores = gates.wide_stream_from_log < 63 && outlen >= ((size_t)1 << std::max(gates.wide_stream_from_log, 0));
    if (ring == 8 || (ring >= 12 && ring <= 14)) { // the fold on the matrix cores too (k_fold3_cross64_mfma); 12..14: ring of 2..4 steps (8 = 4)
        launch_eq_weights_mfma(r3, 3, two24, wbuf, st);
        const fq kfix = ZK_MFMA_FOLD_FIX;
        g_forms |= kFormFold3Cross64Mfma;
        const int np = ring == 8 ? 4 : ring - 10;
#define ZK_F3M(BV, NV) hipLaunchKernelGGL((k_fold3_cross64_mfma<BV, NV>), dim3((unsigned)gx, 2), dim3(256), 0, st, S, D, outlen, (const u32x4*)wbuf, kfix, (fq*)partials, (long long*)raw)
        if (stream_stores) { if (np == 2) ZK_F3M(true, 2); else if (np == 3) ZK_F3M(true, 3); else ZK_F3M(true, 4); }
        else { if (np == 2) ZK_F3M(false, 2); else if (np == 3) ZK_F3M(false, 3); else ZK_F3M(false, 4); }
#undef ZK_F3M
        if (raw) hipLaunchKernelGGL(k_reduce_cross_raw, dim3(160), dim3(kThreads), 0, st, (const long long*)raw, (const fq*)partials, (int)gx, red);
        else hipLaunchKernelGGL(k_reduce_cross, dim3(160), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 80, red);
        hipLaunchKernelGGL(k_cross64_fix, dim3(1), dim3(128), 0, st, (const fq*)red, qn, (fq*)out);
        return;
    }
    launch_eq_weights28(r3, 3, two24, wbuf, st);
    g_forms |= kFormFold3Cross64;
#define ZK_F3X(BV, DV) hipLaunchKernelGGL((k_fold3_cross64<BV, DV>), dim3((unsigned)gx, 2), dim3(512), 0, st, S, D, outlen, (const u32*)wbuf, (fq*)partials)
    if (stream_stores) { if (ring == 5) ZK_F3X(true, 5); else if (ring == 3) ZK_F3X(true, 3); else ZK_F3X(true, 4); }
    else { if (ring == 5) ZK_F3X(false, 5); else if (ring == 3) ZK_F3X(false, 3); else ZK_F3X(false, 4); }
#undef ZK_F3X
    hipLaunchKernelGGL(k_reduce_cross, dim3(160), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 80, red);
    hipLaunchKernelGGL(k_cross64_fix, dim3(1), dim3(128), 0, st, (const fq*)red, qn, (fq*)out);
}
// bind the THREE top variables of four tables of 8 * outlen entries (outlen a multiple of 64): eq weights on the device, then one
// m = 3 pass of the LDS-DMA fold per table
void launch_fold3_tables(const void* const s[4], void* const d[4], size_t outlen, const uint64_t* r3, const uint64_t two24[4], void* wbuf,
                         void* partials, hipStream_t st, const DmaGates& gates) {
    launch_eq_weights28(r3, 3, two24, wbuf, st);
    const bool stream_stores = gates.wide_stream_from_log < 63 && outlen >= ((size_t)1 << std::max(gates.wide_stream_from_log, 0));
    size_t per = (outlen / 64 + 3) / 4;
    if (per > 1024) per = 1024;
    g_forms |= kFormFold3Dma;
    for (int k = 0; k < 4; ++k) {
        if (stream_stores)
            hipLaunchKernelGGL((k_multifold_wide_dma<true, 4>), dim3((unsigned)per), dim3(kThreads), 0, st, (const fq*)s[k], (fq*)d[k], outlen,
                               (const u32*)wbuf, 3, 0, (fq*)partials);
        else
            hipLaunchKernelGGL((k_multifold_wide_dma<false, 4>), dim3((unsigned)per), dim3(kThreads), 0, st, (const fq*)s[k], (fq*)d[k], outlen,
                               (const u32*)wbuf, 3, 0, (fq*)partials);
    }
}
void launch_fold3_tables_small(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* gather, hipStream_t st) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    W28<3> W;
    for (int b = 0; b < 8; ++b)
        for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    size_t gx = (outlen + kThreads - 1) / kThreads;
    if (gx < 1) gx = 1;
    if (gx > (size_t)kMaxBlocks) gx = kMaxBlocks;
    g_forms |= kFormFold3Reg;
    hipLaunchKernelGGL(k_fold3_tables, dim3((unsigned)gx, 4), dim3(kThreads), 0, st, S, D, outlen, W, (fq*)gather);
}
void launch_gather4(const void* const t[4], size_t n, void* out, hipStream_t st) {
    Tab4 T;
    for (int k = 0; k < 4; ++k) T.t[k] = (const fq*)t[k];
    hipLaunchKernelGGL(k_gather4, dim3((unsigned)((4 * n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, T, n, (fq*)out);
}
// the four weights of a two-variable fold into device memory (for the LDS-DMA fold, which takes them as scalar loads)
__global__ void __launch_bounds__(64) k_store_w28x4(W28<2> W, u32* __restrict__ out) {
    if (threadIdx.x < 40) out[threadIdx.x] = W.w[threadIdx.x / 10][threadIdx.x % 10];
}
void launch_fold2_tables(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* wbuf, void* partials, hipStream_t st,
                         const DmaGates& gates) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    W28<2> W;
    for (int b = 0; b < 4; ++b)
        for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    if (wbuf && partials && outlen % 64 == 0 && gates.fold2_from_log < 63 && outlen >= ((size_t)1 << std::max(gates.fold2_from_log, 0))) {
        g_forms |= kFormFold2Dma;
        const bool stream_stores = gates.wide_stream_from_log < 63 && outlen >= ((size_t)1 << std::max(gates.wide_stream_from_log, 0));
        // the largest tables: each table is one m = 2 pass of the LDS-DMA fold (its block sum is not used).  The pass stays
        // bound by its store stream (1 store per 4 reads): 4.67 -> 4.85 TB/s only, and slower than the lane-per-output
        // kernel below 2^22 outputs per table
        hipLaunchKernelGGL(k_store_w28x4, dim3(1), dim3(64), 0, st, W, (u32*)wbuf);
        size_t per = (outlen / 64 + 3) / 4;
        if (per > 1024) per = 1024;
        for (int k = 0; k < 4; ++k) {
            if (stream_stores)
                hipLaunchKernelGGL((k_multifold_wide_dma<true, 4>), dim3((unsigned)per), dim3(kThreads), 0, st, S.t[k], D.t[k], outlen, (const u32*)wbuf, 2,
                                   0, (fq*)partials);
            else
                hipLaunchKernelGGL((k_multifold_wide_dma<false, 4>), dim3((unsigned)per), dim3(kThreads), 0, st, S.t[k], D.t[k], outlen,
                                   (const u32*)wbuf, 2, 0, (fq*)partials);
        }
        return;
    }
    size_t gx = (outlen + kThreads - 1) / kThreads;
    if (gx < 1) gx = 1;
    if (gx > (size_t)kMaxBlocks) gx = kMaxBlocks;
    g_forms |= kFormFold2Reg;
    hipLaunchKernelGGL(k_fold2_tables, dim3((unsigned)gx, 4), dim3(kThreads), 0, st, S, D, outlen, W);
}
// the fold of launch_fold2_tables and the nine sums of its result (launch_sum9 on d[]) in one launch + the second stage: small tables
int launch_fold2_sum9(const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* partials, void* out, hipStream_t st,
                      int host_groups_max) {
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = (fq*)d[k]; }
    W28<2> W;
    for (int b = 0; b < 4; ++b)
        for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    size_t gx = (outlen / 4 + 63) / 64;
    if (gx < 1) gx = 1;
    if (gx > (size_t)kMaxBlocks9) gx = kMaxBlocks9; // 512 and 256 measure the same on GKR config 5
    g_forms |= kFormFold2Sum9;
    if (2 * gx <= (size_t)host_groups_max) { // few workgroups: their partial sums go straight to `out` and the caller adds them -- no second stage
        hipLaunchKernelGGL(k_fold2_sum9, dim3((unsigned)gx, 2), dim3(kThreads), 0, st, S, D, outlen, W, (fq*)out);
        return (int)(2 * gx);
    }
    hipLaunchKernelGGL(k_fold2_sum9, dim3((unsigned)gx, 2), dim3(kThreads), 0, st, S, D, outlen, W, (fq*)partials);
    hipLaunchKernelGGL(k_reduce9, dim3(9), dim3(kThreads), 0, st, (const fq*)partials, (int)gx, 0, (fq*)out);
    return 0;
}

// launch_fold2_tables / launch_fold3_tables (m = 2 / 3) and the 27 sums of the result for the next THREE rounds in one launch (small
// tables; outlen a multiple of 16); m = 0: the 27 sums of the tables s as they are (outlen = their length, d unused).  pwq: the
// relation is s[0] s[1] + s[2] * 1 (s[3], d[3] are not touched).  Returns G > 0 when `out` holds G workgroups' shares
// out[g * 27 + i] for the caller to add, 0 when it holds the 27 sums.
template <int M, bool PWQ>
static int launch_fold_sum27_m(const Tab4& S, const Tab4w& D, size_t outlen, const uint32_t* w28, void* partials, void* out, hipStream_t st,
                               int host_groups_max) {
    W28<M> W{};
    if (M > 0)
        for (int b = 0; b < (1 << M); ++b)
            for (int j = 0; j < 10; ++j) W.w[b][j] = w28[b * 10 + j];
    size_t gx = (outlen / 8 + kThreads / 32 - 1) / (kThreads / 32);
    if (gx < 1) gx = 1;
    if (gx > (size_t)kMaxBlocks9) gx = kMaxBlocks9;
    g_forms |= kFormFoldSum27;
    const unsigned gy = PWQ ? 1 : 2;
    if (gy * gx <= (size_t)host_groups_max) {
        hipLaunchKernelGGL((k_fold_sum27<M, PWQ>), dim3((unsigned)gx, gy), dim3(kThreads), 0, st, S, D, outlen, W, (fq*)out);
        return (int)(gy * gx);
    }
    hipLaunchKernelGGL((k_fold_sum27<M, PWQ>), dim3((unsigned)gx, gy), dim3(kThreads), 0, st, S, D, outlen, W, (fq*)partials);
    hipLaunchKernelGGL(k_reduce27, dim3(27), dim3(kThreads), 0, st, (const fq*)partials, (int)(gy * gx), (fq*)out);
    return 0;
}
int launch_fold_sum27(int m, const void* const s[4], void* const d[4], size_t outlen, const uint32_t* w28, void* partials, void* out,
                      hipStream_t st, int host_groups_max, bool pwq) {
    if ((m != 0 && m != 2 && m != 3) || outlen < 16 || outlen % 16 || (m > 0 && (!w28 || !d))) return -1;
    Tab4 S;
    Tab4w D;
    for (int k = 0; k < 4; ++k) { S.t[k] = (const fq*)s[k]; D.t[k] = d ? (fq*)d[k] : nullptr; }
    if (pwq) {
        g_forms |= kFormPwq;
        return m == 0 ? launch_fold_sum27_m<0, true>(S, D, outlen, w28, partials, out, st, host_groups_max)
             : m == 2 ? launch_fold_sum27_m<2, true>(S, D, outlen, w28, partials, out, st, host_groups_max)
                      : launch_fold_sum27_m<3, true>(S, D, outlen, w28, partials, out, st, host_groups_max);
    }
    return m == 0 ? launch_fold_sum27_m<0, false>(S, D, outlen, w28, partials, out, st, host_groups_max)
         : m == 2 ? launch_fold_sum27_m<2, false>(S, D, outlen, w28, partials, out, st, host_groups_max)
                  : launch_fold_sum27_m<3, false>(S, D, outlen, w28, partials, out, st, host_groups_max);
}

} // namespace zkk
