// zkmi_internal.hpp -- structs and helpers shared by the translation units of libzkmi.so
// (zkmi.cpp: contexts, tables, sumcheck provers; gkr.cpp: circuit + GKR layer prover).
#pragma once
#include "../../include/zkmi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "fq_host.hpp"
#include "keccak_host.hpp"
#include "kernels.h"

using zkh::Fq;

namespace zki {
constexpr size_t kElem = 32;
constexpr size_t kChunkElems = (size_t)1 << 21; // 64 MiB staging chunks for byte conversion
constexpr size_t kLeChunkElems = (size_t)1 << 19; // to_bytes: 16 MiB halves of the pinned staging buffer
constexpr size_t kLeFirstElems = (size_t)1 << 15; // ... and a 1 MiB first chunk (stream_le32)
constexpr size_t kSums = (size_t)1 << zkk::kMaxFoldBatch; // block sums per pass
constexpr size_t kHostTail = 64;                // degree-2 prover: tables of <= this many entries are finished on the host
constexpr size_t kFold3SmallBelow = (size_t)1 << 12; // degree-2 prover: three-variable folds with fewer outputs per table run as one launch (k_fold3_tables)
constexpr int kHostSumGroups = 32;              // fold + nine-sum passes of <= this many workgroups leave their partial sums to the host
constexpr size_t kHostStage = 27 * kHostSumGroups; // elements of the pinned staging buffer (>= 4 * kHostTail): 27 sums per workgroup of k_fold_sum27
static_assert(kHostStage >= 4 * kHostTail, "the staging buffer receives the four tail tables");
static_assert(kHostStage >= kSums, "the staging buffer also receives the block sums of a pass");

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline unsigned ilog2(size_t n) {
    unsigned k = 0;
    while (n > 1) { n >>= 1; ++k; }
    return k;
}
} // namespace zki
using namespace zki;

struct zk_transcript {
    zkh::Keccak256 hasher;
    void append(const uint8_t* d, size_t n) { hasher.update(d, n); }
    void append_fq(const Fq& v) {
        uint8_t b[32];
        zkh::to_le32(v, b);
        hasher.update(b, 32);
    }
    Fq sample() { // transcript.rs:28-36
        uint8_t h[32];
        hasher.peek(h);
        hasher.update(h, 32);
        return zkh::from_le32_mod_order(h);
    }
};

struct zk_comm; // sharded.cpp: the exchange between the ranks of a sharded prove

struct ProfSlot { int kernel; hipEvent_t a, b; uint64_t bytes; };

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;     // false: the caller's stream (zk_ctx_create_on_stream)
    void* partials = nullptr;   // kMaxBlocks * 3 elements
    void* sums_dev = nullptr;   // kSums elements
    void* wide_w = nullptr;     // kWideWeightWords u32: eq weights of a wide fold pass (k_eq_weights28 -> k_multifold_wide)
    uint64_t* sums_host = nullptr; // pinned + device-mapped, kHostStage elements: kernels write small results here directly
    void* sums_host_dev = nullptr; // the device address of sums_host
    int fold_batch = 4;          // rounds bound per device pass (1..kMaxFoldBatch)
    int wide_batch = 8;          // rounds bound by a pass over a LARGE table (0: never; kernels.hip, k_multifold_wide) ...
    int wide_lds_below_log = 15; // wide passes with fewer than 2^k outputs derive their weights per workgroup in LDS (one launch)
    int tail_max_log = 16;       // tables of <= 2^k entries: eight rounds per pass with one wave per output (k_multifold_coop); 0: off
    int tail_out_below_log = 13; // ... used by passes whose output has fewer than 2^k entries
    int wide_min_log_out = 16;   // ... taken while its output keeps at least 2^wide_min_log_out entries
    bool lazy = true;            // inner-product form with one reduction per output (k_multifold_lazy); false: the fold tree
    int deg2_lazy_from_log = 18; // nine-sum sweeps over tables of >= 2^k quarter-entries accumulate unreduced (k_sum9_lazy); < 0: never
    uint64_t two24[4];           // 2^24 in Montgomery form (radix-2^28 reductions divide by 2^280, not 2^256)
    bool deg2_pairs = true;      // degree-2 prover: two rounds per device pass (k_sum9 / k_fold2x4_sum9)
    zkk::DmaGates gates;         // size thresholds of the LDS-DMA forms (zk_ctx_set_dma_gates)
    bool gkr_strict_claims = false; // zk_gkr_verify also binds every layer's claim (zk_ctx_set_gkr_strict_claims)
    bool validate_uploads = true;   // zk_mle_upload_mont* reject limbs >= p (zk_ctx_set_validate)
    zk_comm* comm = nullptr;     // zk_comm_init_*: this context is rank `comm->rank` of a sharded prove
    void* cross_raw = nullptr;   // raw digits of the cross sums (launch_cross64 / launch_fold3_cross64), grow-only, made on first use
    size_t cross_raw_cap = 0;    // bytes
    bool cross_raw_on = true;    // zk_ctx_set_deg2_raw_cross
    void* work[2] = {nullptr, nullptr};
    size_t work_cap[2] = {0, 0}; // bytes
    void* stage = nullptr;       // kChunkElems elements
    uint8_t* le_host = nullptr;  // to_bytes staging: pinned + device-mapped, 2 x kLeChunkElems elements (stream_le32), made on first use
    void* le_host_dev = nullptr; // its device address
    hipEvent_t le_ev[2] = {nullptr, nullptr};
    void* gkr_scratch = nullptr; // layer prover's tables (gkr.cpp), kept between proofs: grow-only
    size_t gkr_scratch_cap = 0;  // bytes
    std::string err;
    bool prof_on = false;
    std::vector<ProfSlot> prof_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    uint64_t prof_launches[ZK_K_COUNT] = {0};
    double prof_ms[ZK_K_COUNT] = {0};
    uint64_t prof_bytes[ZK_K_COUNT] = {0};
};

struct zk_mle {
    zk_ctx* ctx;
    void* d;
    size_t len;
    bool owned;
    int device; // copy of ctx->device: freeing a table must not touch a context that is already gone
};

// a table being folded from the top: the single-GPU prover, a rank's shard, evaluate()
struct zk_shard {
    zk_ctx* ctx;
    const void* cur; // current local table
    unsigned ell;    // variables left in it
    int pass;
};

// four factor tables being folded from the top (degree-2 branch in steps)
struct zk_shard4 {
    zk_ctx* ctx;
    const void* cur[4];
    size_t len; // entries per table
    int pass;
};

#define HIPCHK(ctx, call)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return ZK_E_HIP;                                                                        \
        }                                                                                           \
    } while (0)

namespace zki {

inline int set_device(zk_ctx* c) { HIPCHK(c, hipSetDevice(c->device)); return ZK_OK; }

// rounds the next pass over a local table of `ell` variables should bind: as many as the wide batch allows while
// the folded table keeps 2^wide_min_log_out entries (below that the chip is short of independent outputs and the
// pass turns latency-bound) if that beats the default batch by two rounds or more (measured: tools/wide_sweep.sh),
// the default batch otherwise
inline unsigned next_batch(const zk_ctx* c, unsigned ell) {
    if (c->wide_batch > c->fold_batch && ell > (unsigned)c->wide_min_log_out) {
        const unsigned w = std::min((unsigned)c->wide_batch, ell - (unsigned)c->wide_min_log_out);
        if (w >= (unsigned)c->fold_batch + 2 || w == (unsigned)c->wide_batch) return w;
    }
    // the tail: eight rounds per pass (one wave per output), and a last batch of up to eight rounds straight from the
    // block sums (no fold pass follows it)
    if (c->tail_max_log > 0 && ell <= (unsigned)c->tail_max_log && ell > (unsigned)c->fold_batch)
        return std::min((unsigned)zkk::kMaxFoldBatch, ell);
    return std::min((unsigned)c->fold_batch, ell);
}

// degree-2 prover: does a pass over four tables of n entries serve three rounds (8 x 8 cross sums per term) rather than two?
// linear: the relation is a product term plus a LINEAR term (prove_sum2x2_dev's ones3): one size class later
inline bool deg2_triple_ok(const zk_ctx* c, size_t n, bool linear = false) {
    const int tf = c->gates.deg2_triple_from_log + (linear ? 1 : 0);
    return tf < 63 && n >= 512 && n >= ((size_t)1 << std::max(tf, 0)) && zkk::cross64_fits(n / 8);
}
// ... and does the fold that FOLLOWS a three-round pass leave its result's cross sums too (k_fold3_cross64*)?  One size class further
// down than a first pass: the sweep is already running in that form, and the alternative is four fold launches plus a nine-sum pass
inline bool deg2_triple_cont_ok(const zk_ctx* c, size_t n, bool linear = false) {
    const int tf = c->gates.deg2_triple_from_log + (linear ? 1 : 0);
    return tf < 63 && n >= 512 && n >= ((size_t)1 << std::max(tf - 1, 0)) && zkk::cross64_fits(n / 8);
}

inline int ensure_work(zk_ctx* c, int which, size_t bytes) {
    if (c->work_cap[which] >= bytes) return ZK_OK;
    if (c->work[which]) { HIPCHK(c, hipFree(c->work[which])); c->work[which] = nullptr; c->work_cap[which] = 0; }
    HIPCHK(c, hipMalloc(&c->work[which], bytes));
    c->work_cap[which] = bytes;
    return ZK_OK;
}
inline int ensure_stage(zk_ctx* c) {
    if (!c->stage) HIPCHK(c, hipMalloc(&c->stage, kChunkElems * kElem));
    return ZK_OK;
}

// ---- profiling brackets -------------------------------------------------------------------
struct ProfScope {
    zk_ctx* c;
    bool active;
    ProfSlot slot;
    ProfScope(zk_ctx* ctx, int kernel, uint64_t bytes) : c(ctx), active(ctx->prof_on) {
        if (!active) return;
        slot.kernel = kernel;
        slot.bytes = bytes;
        if (!c->prof_free.empty()) {
            slot.a = c->prof_free.back().first;
            slot.b = c->prof_free.back().second;
            c->prof_free.pop_back();
        } else if (hipEventCreate(&slot.a) != hipSuccess || hipEventCreate(&slot.b) != hipSuccess) {
            active = false;
            return;
        }
        (void)hipEventRecord(slot.a, c->stream);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(slot.b, c->stream);
        c->prof_pending.push_back(slot);
    }
};
// call only after the stream has been synchronized
inline void prof_collect(zk_ctx* c) {
    for (auto& s : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->prof_launches[s.kernel] += 1;
            c->prof_ms[s.kernel] += ms;
            c->prof_bytes[s.kernel] += s.bytes;
        }
        c->prof_free.emplace_back(s.a, s.b);
    }
    c->prof_pending.clear();
}

// second stage + device->host of ns sums; blocks until they are on the host
inline int fetch_sums(zk_ctx* c, int nblocks, int ns, Fq* out) {
    // the second stage stores straight into the device-mapped pinned buffer: no copy command, one sync
    zkk::launch_reduce_partials(c->partials, nblocks, ns, c->sums_host_dev, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < ns; ++s) out[s] = zkh::load(c->sums_host + 4 * s);
    return ZK_OK;
}

// device->host of the ns block sums a launch_block_sums / launch_multifold left in sums_dev
inline int fetch_block_sums(zk_ctx* c, int ns, Fq* out) {
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->sums_host, c->sums_dev, (size_t)ns * kElem, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < ns; ++s) out[s] = zkh::load(c->sums_host + 4 * s);
    return ZK_OK;
}

// eq(r, b) * 2^280 mod p for all b < 2^m as 10 x 28-bit limbs each; r[l] (Montgomery, 4 limbs each)
// belongs to bit (m-1-l) of b: variable 0 is the most significant index bit (evaluation_form.rs:40-42)
inline void eq_weights28(const uint64_t* rs, unsigned m, uint32_t* w28) {
    Fq eq[1 << zkk::kMaxLazyBatch];
    eq[0] = zkh::one();
    unsigned n = 1;
    for (unsigned l = 0; l < m; ++l) { // append variable l as the new least significant bit
        const Fq r = zkh::load(rs + 4 * l), one_minus_r = zkh::sub(zkh::one(), r);
        for (unsigned b = n; b-- > 0;) {
            eq[2 * b + 1] = zkh::mul(eq[b], r);
            eq[2 * b] = zkh::mul(eq[b], one_minus_r);
        }
        n *= 2;
    }
    // eq[b] holds eq*2^256 mod p; one more Montgomery product by (2^24 in Montgomery form) gives the
    // plain integer eq * 2^280 mod p
    const Fq two24 = zkh::from_u64((uint64_t)1 << 24);
    for (unsigned b = 0; b < n; ++b) {
        const Fq v = zkh::mul(eq[b], two24);
        for (int j = 0; j < 10; ++j) {
            const unsigned bit = 28 * j, w = bit / 64, sh = bit % 64;
            uint64_t x = v.l[w] >> sh;
            if (sh > 36 && w + 1 < 4) x |= v.l[w + 1] << (64 - sh);
            w28[b * 10 + j] = (uint32_t)(x & 0x0fffffffu);
        }
    }
}

inline int new_mle(zk_ctx* c, size_t len, zk_mle** out) {
    if (!is_pow2(len)) return ZK_E_NOT_POW2; // evaluation_form.rs:23-27
    zk_mle* m = new (std::nothrow) zk_mle{c, nullptr, len, true, c->device};
    if (!m) return ZK_E_ALLOC;
    hipError_t e = hipMalloc(&m->d, len * kElem);
    if (e != hipSuccess) {
        c->err = std::string("hipMalloc table: ") + hipGetErrorString(e);
        delete m;
        return ZK_E_ALLOC;
    }
    *out = m;
    return ZK_OK;
}

// to_bytes of a table fed to `sink` in chunks.  The conversion kernel stores the LE32 bytes straight into device-mapped pinned host
// memory (no copy command: a 32 MiB device-to-host copy into pageable memory was 64 staged blit kernels), in two halves so that chunk
// i + 1 is converted while the host consumes chunk i (the sink is a Keccak absorb at ~0.6 GB/s: the device side hides under it).
template <class Sink>
int stream_le32(zk_ctx* c, const void* table, size_t len, Sink&& sink) {
    if (!c->le_host) {
        HIPCHK(c, hipHostMalloc((void**)&c->le_host, 2 * kLeChunkElems * kElem, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer(&c->le_host_dev, c->le_host, 0));
        HIPCHK(c, hipEventCreateWithFlags(&c->le_ev[0], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->le_ev[1], hipEventDisableTiming));
    }
    // chunk 0 is SHORT (1 MiB): nothing can be consumed before the first chunk has crossed PCIe (16 MiB: 0.3 ms of a 45 ms absorb with
    // the host idle); from then on the device side hides under the sink.  Chunk i > 0 covers [kLeFirstElems + (i - 1) kLeChunkElems, ...)
    const size_t first = std::min(len, kLeFirstElems);
    const size_t nchunks = len ? 1 + (len - first + kLeChunkElems - 1) / kLeChunkElems : 0;
    auto span = [&](size_t i, size_t* off, size_t* n) {
        *off = i == 0 ? 0 : first + (i - 1) * kLeChunkElems;
        *n = i == 0 ? first : std::min(kLeChunkElems, len - *off);
    };
    auto convert = [&](size_t i) -> int {
        size_t off, n;
        span(i, &off, &n);
        zkk::launch_to_le32((const uint8_t*)table + off * kElem, (uint8_t*)c->le_host_dev + (i & 1) * kLeChunkElems * kElem, n, c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->le_ev[i & 1], c->stream));
        return ZK_OK;
    };
    int rc;
    if (nchunks && (rc = convert(0))) return rc;
    for (size_t i = 0; i < nchunks; ++i) {
        if (i + 1 < nchunks && (rc = convert(i + 1))) return rc; // its half was consumed by the sink one iteration ago
        HIPCHK(c, hipEventSynchronize(c->le_ev[i & 1]));
        size_t off, n;
        span(i, &off, &n);
        sink(c->le_host + (i & 1) * kLeChunkElems * kElem, n * kElem, off);
    }
    return ZK_OK;
}

// defined in zkmi.cpp
// m rounds of the MLE branch on the gathered [world][2^m] block sums (see zkmi.cpp)
void host_rounds(zk_transcript* tr, const uint64_t* sums, int world, unsigned m, uint64_t* coeffs_out, uint64_t* ch_out, uint64_t* folded_out,
                 uint64_t* total_out);
int fold_pass(const zk_ctx& k, const uint64_t two24[4], const void* src, void* dst, size_t lout, int m, const uint64_t* r_mont, int m_next,
              void* partials, void* wide_w, void* sums, hipStream_t stream);
int prove_mle_dev(zk_ctx* c, const void* table, unsigned n, zk_transcript* tr, uint64_t claim_out[4], uint64_t* coeffs_out,
                  uint64_t* challenges_out);
int prove_sum2x2_dev(zk_ctx* c, const void* const tables[4], size_t len, zk_transcript* tr, uint64_t* coeffs_out,
                     uint64_t* challenges_out, uint64_t* final_out, bool ones3 = false);
// the host part of three rounds from the ranks' shares of the 2 x 64 cross sums (zkmi.cpp)
void host_rounds_from_cross64(zk_transcript* tr, const uint64_t* cross, int world, uint64_t* coeffs_out, uint64_t* rs);
// defined in sharded.cpp: the same for tables sharded over the ranks of the context's communicator (len_local entries per rank)
int prove_sum2x2_sharded_dev(zk_ctx* c, const void* const tabs[4], size_t len_local, zk_transcript* tr, int gather_log, uint64_t* coeffs_out,
                             uint64_t* challenges_out, uint64_t* final_out);

} // namespace zki
using namespace zki;
