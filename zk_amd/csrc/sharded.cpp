// sharded.cpp -- the sumcheck provers over a table sharded across the GPUs of one node, inside the library
// (include/zkmi.h "sharded provers").  One process per GPU; rank q of world = 2^g holds the entries
// whose global index is q mod world, i.e. the table is sharded on the g variables bound LAST, so every fold pair
// (j, j + N/2) of evaluation_form.rs:61-79 is rank-local for the first n - g rounds.  Per device pass the ranks
// all-gather their small sums (2^m block sums / nine sums: <= 8 KiB per rank) and every rank runs the identical host
// part (Keccak transcript, coefficients, challenges: nothing is broadcast back); once the local tables are small the
// ranks all-gather the TABLES and every rank finishes alone.  What must stay identical to the reference:
// generate_round_polys, sumcheck/src/prover.rs:45-155 -- the coefficients and the transcript stream.
//
// The exchange is RCCL over xGMI (ncclAllGather on the context stream; librccl is loaded on first use, so single-GPU
// users never map it) or a host all-gather callback the caller brings (MPI_Allgather shaped; the multi-rank parity
// tests bring a gloo one, because RCCL refuses two ranks on one GPU and a test box has one).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "zkmi_internal.hpp"

struct zk_comm {
    int rank = 0, world = 1;
    // RCCL (resolved from librccl.so.1 on first use)
    void* dl = nullptr;
    ncclComm_t nccl = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    // host all-gather callback
    zk_allgather_host_fn fn = nullptr;
    void* user = nullptr;
    // staging, grow-only: send (pinned), recv (pinned; for RCCL the gathered block is copied here), device receive buffer,
    // and the gathered + interleaved tables of the switch to replicated passes
    void* send_host = nullptr;
    size_t send_cap = 0;
    void* recv_host = nullptr;
    size_t recv_cap = 0;
    void* recv_dev = nullptr;
    size_t recv_dev_cap = 0;
    void* tables_dev = nullptr;
    size_t tables_cap = 0;
    void* send_dev = nullptr; // contiguous copy of the local tables (four tables are four buffers)
    size_t send_dev_cap = 0;
};

namespace {

struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    std::string err;
};

// librccl.so.1: the copy already in the process if the host framework loaded one (same SONAME), else ROCm's
RcclApi* rccl_api() {
    static RcclApi api;
    if (api.dl) return &api;
    void* dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) {
        const char* e = dlerror();
        api.err = std::string("dlopen librccl.so.1: ") + (e ? e : "not found");
        return nullptr;
    }
    api.get_unique_id = (decltype(api.get_unique_id))dlsym(dl, "ncclGetUniqueId");
    api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(dl, "ncclCommInitRank");
    api.all_gather = (decltype(api.all_gather))dlsym(dl, "ncclAllGather");
    api.comm_destroy = (decltype(api.comm_destroy))dlsym(dl, "ncclCommDestroy");
    api.error_string = (decltype(api.error_string))dlsym(dl, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.all_gather || !api.comm_destroy || !api.error_string) {
        api.err = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString";
        dlclose(dl);
        return nullptr;
    }
    api.dl = dl;
    return &api;
}

int comm_fail(zk_ctx* c, const std::string& what) {
    c->err = what;
    return ZK_E_COMM;
}

int grow_dev(zk_ctx* c, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return ZK_OK;
    if (*p) { HIPCHK(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipMalloc(p, bytes));
    *cap = bytes;
    return ZK_OK;
}
int grow_host(zk_ctx* c, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return ZK_OK;
    if (*p) { HIPCHK(c, hipHostFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(c, hipHostMalloc(p, bytes, hipHostMallocDefault));
    *cap = bytes;
    return ZK_OK;
}

int new_comm(zk_ctx* c, int rank, int world, zk_comm** out) {
    if (!c || world < 1 || rank < 0 || rank >= world || !is_pow2((size_t)world)) return ZK_E_ARG;
    if (c->comm) return comm_fail(c, "this context already has a communicator (zk_comm_destroy first)");
    zk_comm* k = new (std::nothrow) zk_comm();
    if (!k) return ZK_E_ALLOC;
    k->rank = rank;
    k->world = world;
    *out = k;
    return ZK_OK;
}

// every rank's `bytes` at src_dev -> the host: [world][bytes] in comm->recv_host.  Blocks until the block is on the host.
int gather_to_host(zk_ctx* c, const void* src_dev, size_t bytes, const uint8_t** out) {
    zk_comm* k = c->comm;
    const size_t all = bytes * (size_t)k->world;
    int rc;
    if ((rc = grow_host(c, &k->recv_host, &k->recv_cap, all))) return rc;
    if (k->nccl) {
        if ((rc = grow_dev(c, &k->recv_dev, &k->recv_dev_cap, all))) return rc;
        const ncclResult_t r = k->all_gather(src_dev, k->recv_dev, bytes, ncclUint8, k->nccl, c->stream);
        if (r != ncclSuccess) return comm_fail(c, std::string("ncclAllGather: ") + k->error_string(r));
        HIPCHK(c, hipMemcpyAsync(k->recv_host, k->recv_dev, all, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else if (!k->fn) { // a single rank without a transport: nothing to exchange
        HIPCHK(c, hipMemcpyAsync(k->recv_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        if ((rc = grow_host(c, &k->send_host, &k->send_cap, bytes))) return rc;
        HIPCHK(c, hipMemcpyAsync(k->send_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int r = k->fn(k->user, k->send_host, k->recv_host, bytes);
        if (r != 0) return comm_fail(c, "host all-gather callback returned " + std::to_string(r));
    }
    *out = (const uint8_t*)k->recv_host;
    return ZK_OK;
}

// The switch to replicated passes: every rank's ntab local tables of L entries (src_dev: [ntab][L], contiguous) -> the ntab
// remaining tables of L * world entries in comm->tables_dev ([ntab][L * world]; entry j * world + q = rank q's entry j).
// Asynchronous on the context stream for RCCL; the host-callback form blocks.
int gather_tables(zk_ctx* c, const void* src_dev, int ntab, size_t L, void** out) {
    zk_comm* k = c->comm;
    const size_t bytes = (size_t)ntab * L * kElem, all = bytes * (size_t)k->world;
    int rc;
    if ((rc = grow_dev(c, &k->tables_dev, &k->tables_cap, all))) return rc;
    if (!k->nccl && !k->fn) { // a single rank without a transport: the local tables are the remaining tables.  Copied, because
        // the caller goes on to fold them into the context's work buffers, which src_dev may be one of
        HIPCHK(c, hipMemcpyAsync(k->tables_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
        *out = k->tables_dev;
        return ZK_OK;
    }
    if ((rc = grow_dev(c, &k->recv_dev, &k->recv_dev_cap, all))) return rc;
    if (k->nccl) {
        const ncclResult_t r = k->all_gather(src_dev, k->recv_dev, bytes, ncclUint8, k->nccl, c->stream);
        if (r != ncclSuccess) return comm_fail(c, std::string("ncclAllGather: ") + k->error_string(r));
    } else {
        if ((rc = grow_host(c, &k->send_host, &k->send_cap, bytes))) return rc;
        if ((rc = grow_host(c, &k->recv_host, &k->recv_cap, all))) return rc;
        HIPCHK(c, hipMemcpyAsync(k->send_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int r = k->fn(k->user, k->send_host, k->recv_host, bytes);
        if (r != 0) return comm_fail(c, "host all-gather callback returned " + std::to_string(r));
        HIPCHK(c, hipMemcpyAsync(k->recv_dev, k->recv_host, all, hipMemcpyHostToDevice, c->stream));
    }
    zkk::launch_interleave(k->recv_dev, k->tables_dev, k->world, ntab, L, c->stream);
    HIPCHK(c, hipGetLastError());
    *out = k->tables_dev;
    return ZK_OK;
}

// transcript.append(&init_polynomial.to_bytes()) (prover.rs:162) for a sharded table: the byte stream is in GLOBAL
// index order, so chunks of every shard are gathered and interleaved.  Serial Keccak over 32 * N bytes on every rank
// (SURVEY.md H1).
int absorb_sharded(zk_ctx* c, const zk_mle* local, zk_transcript* tr) {
    zk_comm* k = c->comm;
    const size_t chunk = std::min(local->len, (size_t)1 << 16);
    int rc;
    if ((rc = ensure_stage(c))) return rc;
    std::vector<uint8_t> host(chunk * (size_t)k->world * kElem);
    for (size_t off = 0; off < local->len; off += chunk) {
        zkk::launch_to_le32((const uint8_t*)local->d + off * kElem, c->stage, chunk, c->stream);
        HIPCHK(c, hipGetLastError());
        void* all = nullptr;
        if ((rc = gather_tables(c, c->stage, 1, chunk, &all))) return rc;
        HIPCHK(c, hipMemcpyAsync(host.data(), all, host.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        tr->append(host.data(), host.size());
    }
    return ZK_OK;
}

} // namespace

extern "C" {

int zk_comm_unique_id(uint8_t out[ZK_COMM_ID_BYTES]) {
    static_assert(ZK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "zk_comm_unique_id hands out an ncclUniqueId");
    if (!out) return ZK_E_ARG;
    RcclApi* api = rccl_api();
    if (!api) return ZK_E_COMM;
    ncclUniqueId id;
    if (api->get_unique_id(&id) != ncclSuccess) return ZK_E_COMM;
    std::memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ZK_OK;
}

int zk_comm_init_rccl(zk_ctx* c, const uint8_t id_bytes[ZK_COMM_ID_BYTES], int rank, int world) {
    if (!c || !id_bytes) return ZK_E_ARG;
    int rc = set_device(c);
    if (rc) return rc;
    zk_comm* k = nullptr;
    if ((rc = new_comm(c, rank, world, &k))) return rc;
    { // world = 1 gets a communicator too: the one-rank case then runs the same calls as the N-rank one
        RcclApi* api = rccl_api();
        if (!api) {
            delete k;
            return comm_fail(c, "RCCL is not available: librccl.so.1 could not be loaded");
        }
        ncclUniqueId id;
        std::memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
        const ncclResult_t r = api->comm_init_rank(&k->nccl, world, id, rank);
        if (r != ncclSuccess) {
            const std::string msg = std::string("ncclCommInitRank: ") + api->error_string(r);
            delete k;
            return comm_fail(c, msg);
        }
        k->all_gather = api->all_gather;
        k->comm_destroy = api->comm_destroy;
        k->error_string = api->error_string;
    }
    c->comm = k;
    return ZK_OK;
}

int zk_comm_init_host(zk_ctx* c, int rank, int world, zk_allgather_host_fn fn, void* user) {
    if (!c || (world > 1 && !fn)) return ZK_E_ARG; // world = 1 and no callback: a single rank that exchanges nothing
    zk_comm* k = nullptr;
    int rc = new_comm(c, rank, world, &k);
    if (rc) return rc;
    k->fn = fn;
    k->user = user;
    c->comm = k;
    return ZK_OK;
}

void zk_comm_destroy(zk_ctx* c) {
    if (!c || !c->comm) return;
    zk_comm* k = c->comm;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (k->nccl && k->comm_destroy) (void)k->comm_destroy(k->nccl);
    if (k->send_host) (void)hipHostFree(k->send_host);
    if (k->recv_host) (void)hipHostFree(k->recv_host);
    if (k->recv_dev) (void)hipFree(k->recv_dev);
    if (k->tables_dev) (void)hipFree(k->tables_dev);
    if (k->send_dev) (void)hipFree(k->send_dev);
    delete k;
    c->comm = nullptr;
}

int zk_comm_rank(const zk_ctx* c) { return c && c->comm ? c->comm->rank : -1; }
int zk_comm_world(const zk_ctx* c) { return c && c->comm ? c->comm->world : 0; }

// generate_round_polys / generate_sumcheck_proof (sumcheck/src/prover.rs:45-173), MLE branch, for a table sharded over the
// ranks of the context's communicator.  Every rank passes its own shard and gets the same proof.
int zk_sumcheck_prove_mle_sharded(zk_ctx* c, const zk_mle* local, int absorb_table, zk_transcript* transcript, int gather_log,
                                  uint64_t claim_out[4], uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!c || !local || !claim_out || !coeffs_out || local->ctx != c) return ZK_E_ARG;
    if (!c->comm) return comm_fail(c, "no communicator on this context (zk_comm_init_rccl / zk_comm_init_host)");
    zk_comm* k = c->comm;
    int rc = set_device(c);
    if (rc) return rc;
    const unsigned g = ilog2((size_t)k->world);
    zk_transcript fresh;
    zk_transcript* tr = transcript ? transcript : &fresh;
    if (!transcript || absorb_table)
        if ((rc = absorb_sharded(c, local, tr))) return rc;

    zk_shard st{c, local->d, ilog2(local->len), 0};
    unsigned done = 0; // rounds so far
    // the ranks' local tables -> the remaining table on every rank, which finishes alone on the running transcript
    auto finish_gathered = [&]() -> int {
        void* all = nullptr;
        int r2 = gather_tables(c, st.cur, 1, (size_t)1 << st.ell, &all);
        if (r2) return r2;
        uint64_t total[4];
        r2 = prove_mle_dev(c, all, st.ell + g, tr, total, coeffs_out + 8 * done, challenges_out ? challenges_out + 4 * done : nullptr);
        if (r2) return r2;
        if (done == 0) std::memcpy(claim_out, total, kElem);
        return ZK_OK;
    };
    if (gather_log >= 0 && st.ell <= (unsigned)gather_log) return finish_gathered();
    if (st.ell < 1) { // one entry per rank: they ARE the remaining g-variable table
        const uint8_t* entries = nullptr;
        if ((rc = gather_to_host(c, st.cur, kElem, &entries))) return rc;
        if (g) host_rounds(tr, (const uint64_t*)entries, 1, g, coeffs_out, challenges_out, nullptr, claim_out);
        else std::memcpy(claim_out, entries, kElem);
        return ZK_OK;
    }

    std::vector<uint64_t> folded(4 * (size_t)k->world);
    unsigned m = next_batch(c, st.ell);
    if ((rc = zk_shard_block_sums(&st, (int)m, c->sums_dev))) return rc;
    for (;;) {
        const uint8_t* sums = nullptr;
        if ((rc = gather_to_host(c, c->sums_dev, ((size_t)1 << m) * kElem, &sums))) return rc;
        const bool last = st.ell == m;
        uint64_t rs[4 * zkk::kMaxFoldBatch];
        host_rounds(tr, (const uint64_t*)sums, k->world, m, coeffs_out + 8 * done, rs, last ? folded.data() : nullptr,
                    done == 0 ? claim_out : nullptr);
        if (challenges_out) std::memcpy(challenges_out + 4 * done, rs, (size_t)m * kElem);
        done += m;
        if (last) break;
        if (gather_log >= 0 && st.ell - m <= (unsigned)gather_log) {
            if ((rc = zk_shard_fold(&st, (int)m, rs, 0, nullptr))) return rc;
            rc = finish_gathered();
            prof_collect(c);
            return rc;
        }
        const unsigned m_next = next_batch(c, st.ell - m);
        if ((rc = zk_shard_fold(&st, (int)m, rs, (int)m_next, c->sums_dev))) return rc;
        m = m_next;
    }
    // the remaining g-variable table (entry q = rank q's fully folded local table) is on every host already
    if (g) host_rounds(tr, folded.data(), 1, g, coeffs_out + 8 * done, challenges_out ? challenges_out + 4 * done : nullptr, nullptr, nullptr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZK_OK;
}

// generate_proof_for_partial_verify (prover.rs:175-190), SumPolynomial branch, for four factor tables sharded the same way.
int zk_sumcheck_prove_sum2x2_sharded(zk_ctx* c, zk_mle* const tables[4], const uint64_t claim_mont[4], zk_transcript* tr, int gather_log,
                                     uint64_t* coeffs_out, uint64_t* challenges_out) {
    if (!c || !tables || !claim_mont || !tr || !coeffs_out) return ZK_E_ARG;
    for (int t = 0; t < 4; ++t)
        if (!tables[t] || tables[t]->ctx != c) return ZK_E_ARG;
    for (int t = 1; t < 4; ++t)
        if (tables[t]->len != tables[0]->len) return ZK_E_LEN_MISMATCH;
    const void* t[4] = {tables[0]->d, tables[1]->d, tables[2]->d, tables[3]->d};
    return prove_sum2x2_sharded_dev(c, t, tables[0]->len, tr, gather_log, coeffs_out, challenges_out, nullptr);
}

} // extern "C"

// the same on raw device tables (the GKR layer prover's); final_out (optional, 4 elements: every table folded at all challenges)
// needs the table gather (gather_log >= 0)
int zki::prove_sum2x2_sharded_dev(zk_ctx* c, const void* const tabs[4], size_t len_local, zk_transcript* tr, int gather_log, uint64_t* coeffs_out,
                                  uint64_t* challenges_out, uint64_t* final_out) {
    if (!c->comm) return comm_fail(c, "no communicator on this context (zk_comm_init_rccl / zk_comm_init_host)");
    if (final_out && gather_log < 0) return ZK_E_ARG;
    zk_comm* k = c->comm;
    int rc = set_device(c);
    if (rc) return rc;
    const unsigned g = ilog2((size_t)k->world);
    zk_shard4 s{c, {tabs[0], tabs[1], tabs[2], tabs[3]}, len_local, 0};
    unsigned ell = ilog2(s.len), done = 0;
    uint64_t rs[8];

    if (gather_log >= 0) { // two rounds per exchange while the local tables are large, then gather the tables
        const unsigned T = std::max(1, gather_log);
        if (ell > T) {
            // three rounds per exchange (the ranks' shares of the 2 x 64 cross sums: 4 KiB each) while the local tables are large,
            // two (nine sums) below; the cross sums are sums over the quarter-index, which the low-bit sharding partitions
            int K = zk_shard4_can_triple(&s) ? 3 : 2;
            if ((rc = K == 3 ? zk_shard4_cross64(&s, c->sums_dev) : zk_shard4_sum9(&s, c->sums_dev))) return rc;
            uint64_t r3[12];
            while (ell > T) { // ell >= 2; a three-round pass only runs on tables of >= 2^9 entries
                const uint8_t* sums = nullptr;
                if ((rc = gather_to_host(c, c->sums_dev, (K == 3 ? 128 : 9) * kElem, &sums))) return rc;
                if (K == 3) rc = zk_sumcheck_rounds_from_cross64(tr, (const uint64_t*)sums, k->world, coeffs_out + 12 * done, r3);
                else rc = zk_sumcheck_round_pair_from_sum9(tr, (const uint64_t*)sums, k->world, coeffs_out + 12 * done, r3);
                if (rc) return rc;
                if (challenges_out) std::memcpy(challenges_out + 4 * done, r3, (size_t)K * kElem);
                ell -= (unsigned)K;
                done += (unsigned)K;
                if (K == 3) {
                    int next = 0;
                    if ((rc = zk_shard4_fold3(&s, r3, ell > T ? c->sums_dev : nullptr, &next))) return rc;
                    if (ell > T) K = next; // 3 or 2: the tables keep >= 2^(T+1) >= 4 entries
                } else if ((rc = zk_shard4_fold2(&s, r3, ell > T ? c->sums_dev : nullptr))) {
                    return rc;
                }
            }
        }
        // the four local tables as one contiguous block (they are folded into the context's work buffers next, which the
        // current tables may live in), gathered into the four remaining tables
        const void* cur[4];
        if ((rc = grow_dev(c, &k->send_dev, &k->send_dev_cap, 4 * s.len * kElem))) return rc;
        if ((rc = zk_shard4_copy_tables(&s, k->send_dev))) return rc;
        void* all = k->send_dev;
        if (k->nccl || k->fn)
            if ((rc = gather_tables(c, k->send_dev, 4, s.len, &all))) return rc;
        const size_t len = s.len * (size_t)k->world;
        for (int t = 0; t < 4; ++t) cur[t] = (const uint8_t*)all + (size_t)t * len * kElem;
        rc = prove_sum2x2_dev(c, cur, len, tr, coeffs_out + 12 * done, challenges_out ? challenges_out + 4 * done : nullptr, final_out);
        prof_collect(c);
        return rc;
    }

    // one round per exchange (three evaluation shares), sharded to the end; the rank bits are finished on the host
    if (ell < 1) { // one entry per rank and table already
        if ((rc = zk_shard4_copy_tables(&s, c->sums_dev))) return rc;
    } else if ((rc = zk_shard4_eval(&s, c->sums_dev))) {
        return rc;
    }
    for (unsigned r = 0; r < ell; ++r) {
        const uint8_t* ev = nullptr;
        if ((rc = gather_to_host(c, c->sums_dev, 3 * kElem, &ev))) return rc;
        if ((rc = zk_sumcheck_round_from_evals(tr, (const uint64_t*)ev, k->world, coeffs_out + 12 * done, rs))) return rc;
        if (challenges_out) std::memcpy(challenges_out + 4 * done, rs, kElem);
        if ((rc = zk_shard4_fold(&s, rs, c->sums_dev))) return rc;
        ++done;
    }
    if (g) { // every rank's four remaining entries: rows [world][4] -> tables [4][world]
        const uint8_t* rows = nullptr;
        if ((rc = gather_to_host(c, c->sums_dev, 4 * kElem, &rows))) return rc;
        std::vector<uint64_t> tabs(16 * (size_t)k->world);
        for (int q = 0; q < k->world; ++q)
            for (int t = 0; t < 4; ++t) std::memcpy(&tabs[4 * ((size_t)t * k->world + q)], rows + ((size_t)q * 4 + t) * kElem, kElem);
        rc = zk_sumcheck_prove_sum2x2_host(tr, tabs.data(), (size_t)k->world, coeffs_out + 12 * done, challenges_out ? challenges_out + 4 * done : nullptr);
        if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    return ZK_OK;
}
