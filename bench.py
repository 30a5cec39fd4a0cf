#!/usr/bin/env python3
"""bench.py -- sumcheck prove time / field elements per second on MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (single GPU)
  python bench.py --gpus N ...                           (starts its own N ranks, one per GPU, before touching a GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, the driver's form)

A "step" is one complete run of the prover's round loop (sumcheck/src/prover.rs:45-155) over a
synthetic, device-resident 2^log_n-entry BN254-Fq table: n rounds of
{reduce -> host Keccak/challenge -> fold}.  The table absorb into Keccak (32*N bytes through a
serial sponge) happens once, OUTSIDE the timed region, and is reported separately
(`absorb_ms`); `value` is table entries / round-loop time.  Prints ONE JSON line on rank 0; with one
GPU the same line also carries the other workloads north_star names (`extra`: a 2^20..2^30 size sweep,
the degree-2 4 x 2^26 prove and the GKR config-5 prove).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = "sumcheck prove ms + field elems/s at 2^28-entry MLE, 1/8 GPUs; % HBM roofline"
DTYPE = "u32x8 (254-bit Montgomery, BN254 Fq)"
KERNEL_NAMES = ("block_sums", "multifold_wide", "multifold", "multifold_tail", "eval3", "fold4_eval3", "sum9", "fold2x4_sum9")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=28, help="table entries = 2^log_n (BASELINE metric: 28)")
    ap.add_argument("--workload", choices=["mle", "sum2x2", "gkr"], default="mle")
    ap.add_argument("--gkr-depth", type=int, default=16)
    ap.add_argument("--gkr-log-width", type=int, default=20, help="gates per layer = 2^k (BASELINE config 5: depth 16, k = 20)")
    ap.add_argument("--no-absorb", action="store_true", help="skip the one-off 32*N-byte Keccak table absorb")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workloads (size sweep, degree 2, GKR) of the single-GPU line")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 counter passes (child runs of this command, one step each) that fill roofline.traffic")
    ap.add_argument("--cpu-log-n", type=int, default=24, help="bounded CPU sample size (2^k entries)")
    ap.add_argument("--cpu-full-log", type=int, default=None, help="one more CPU sample at 2^k entries (SURVEY 8d: 2^28 once; tens of seconds, ~26 GiB)")
    ap.add_argument("--fold-batch", type=int, default=0, help="rounds bound per device pass (0 = library default)")
    ap.add_argument("--wide-batch", type=int, default=-1, help="rounds bound by a pass over a large table (5..8, 0 = off, -1 = library default 8)")
    ap.add_argument("--wide-min-log-out", type=int, default=16, help="wide passes are taken while the folded table keeps >= 2^k entries")
    ap.add_argument("--wide-lds-below", type=int, default=15, help="wide passes with < 2^k outputs make their weights per workgroup in LDS")
    ap.add_argument("--tail-max-log", type=int, default=-1, help="tables of <= 2^k entries bind eight rounds per pass (0 = off, -1 = library default 16)")
    ap.add_argument("--tail-out-below", type=int, default=13, help="passes with < 2^k outputs use one wave per output")
    ap.add_argument("--fold-variant", type=int, default=0, help="0 eq-weight inner product, 1 fold tree (cross-check form)")
    ap.add_argument("--deg2-rounds", type=int, default=0, help="degree-2 prover: rounds per device pass (1 or 2; 0 = library default, 2)")
    ap.add_argument("--gather-log", type=int, default=13, help="multi-GPU: local tables of <= 2^k entries are all-gathered and finished on every rank (-1 = sharded to the end)")
    ap.add_argument("--deg2-lazy-from", type=int, default=None, help="degree-2 prover: unreduced nine-sum accumulation from 2^k quarter-entries (negative = never)")
    ap.add_argument("--wide-mfma", type=int, default=None, help="wide fold passes with >= 2^k outputs contract on the matrix cores (64 = never; default: library default)")
    ap.add_argument("--wide-mfma-wg", type=int, default=1, help="workgroups per CU of the matrix-core wide fold")
    ap.add_argument("--wide-mfma-depth", type=int, default=2, help="steps in flight per wave of the matrix-core wide fold (2..4)")
    ap.add_argument("--deg2-mfma-from", type=int, default=None, help="degree-2 prover: nine sums on the matrix cores from 2^k quarter-entries (64 = never)")
    ap.add_argument("--deg2-triple-from", type=int, default=None, help="degree-2 prover: three rounds per pass on tables of >= 2^k entries (64 = never)")
    ap.add_argument("--deg2-small-triple", type=int, default=None, help="degree-2 prover: three rounds per pass (fold + 27 sums in one kernel) for folded tables below 2^k entries (0 = never)")
    ap.add_argument("--deg2-raw-cross", type=int, default=None, help="1 (library default): the three-round sweeps leave raw digits for the second stage; 0: field elements per workgroup")
    ap.add_argument("--dma-gates", type=int, nargs=3, default=None, metavar=("SUMS_FROM", "WIDE_STREAM_FROM", "FOLD2_FROM"),
                    help="size thresholds (log2 entries) of the LDS-DMA block sums, the streaming stores of the wide fold and the LDS-DMA two-variable fold (library defaults 27 22 22)")
    ap.add_argument("--deg2-first-triple", type=int, nargs=2, default=None, metavar=("BELOW", "LINEAR_BELOW"),
                    help="degree-2 prover: first pass as 27 sums below 2^BELOW entries (0 = never); LINEAR_BELOW: the same for the GKR phases' product-plus-linear form (0 = that form off)")
    ap.add_argument("--deg2-fused-below", type=int, default=None, help="degree-2 prover: fold + nine sums in one kernel for folded tables below 2^k entries (0 = never)")
    ap.add_argument("--deg2-fold3-fused", type=int, default=None, help="degree-2 prover: three-variable fold fused with the next cross sums (0 = off, 8 = fold on the matrix cores (default), 3..5 = DMA slots per wave of the VALU-fold form)")
    ap.add_argument("--deg2-triple-ring", type=int, default=None, help="LDS tiles of the three-round sweep (3 or 4; default: the library's, 4)")
    ap.add_argument("--weak", action="store_true", help="multi-GPU: 2^log_n entries PER GPU instead of in total")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the ranks' rendezvous (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--shared-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path even with one rank")
    ap.add_argument("--sharded-impl", choices=["lib", "py"], default="lib",
                    help="multi-GPU loop: inside libzkmi.so (RCCL or host all-gather, default) or zk_amd/sharded.py over torch.distributed")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args) -> int:
    """`--gpus N` without a launcher: start N fresh ranks (one per GPU) BEFORE this process touches a GPU, pass rank 0's
    line through, and refuse to print a line whose n_gpus is not the world size that ran."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        err = proc.stderr.decode(errors="replace")
        sys.stderr.write(err)
        line = None
        for raw in proc.stdout.decode(errors="replace").splitlines():
            if raw.startswith("{"):
                line = raw
        # the port was free when picked and taken before the ranks bound it: nothing ran, pick another (no GPU work is repeated)
        if proc.returncode != 0 and line is None and ("EADDRINUSE" in err or "address already in use" in err.lower()) and attempt < 2:
            sys.stderr.write(f"bench.py: rendezvous port {port} was taken before the ranks bound it; trying another\n")
            continue
        break
    if proc.returncode != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit {proc.returncode}); no result line\n")
        return proc.returncode or 1
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} GPUs but the run reports n_gpus = {got.get('n_gpus')}: refusing to print it\n")
        return 1
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(log_n: int, extra_logs=(16, 20), full_log=None):
    """The CPU oracle (reference-faithful C restatement, single thread like the reference) timed
    on this host: a reported baseline, never the thing measured above.  A second figure times an honestly
    optimised CPU variant (oracle/fast.c: in-place fused fold, OpenMP) so the ratio is not inflated by the
    reference's per-round copies.  SURVEY.md 8(d): samples at 2^16, 2^20, 2^24 always (`samples`); 2^28 once
    (--cpu-full-log 28: tens of seconds and ~26 GiB, so not in the default run; profiles/ keeps that run's line)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    claim = np.empty(4, dtype=np.uint64)
    threads = min(16, os.cpu_count() or 1, O.lib().orc_max_threads())  # the box's CPU share for one GPU

    def sample(lg, with_absorb=True, with_fast=True):
        n = 1 << lg
        table = O.gen_table(SEED, 0, n)
        reps = max(1, min(32, (1 << 20) >> lg))   # small tables: best of a few runs, they finish in microseconds
        out = {"log_n": lg}
        for absorb, key in ((0, "round_loop_ms"), (1, "with_table_absorb_ms")):
            if absorb and not with_absorb:
                continue
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                rc = O.lib().orc_sumcheck_prove_timed(O.ptr(table), C.c_size_t(n), C.c_int(absorb), O.ptr(claim))
                dt = time.perf_counter() - t0
                assert rc == 0
                best = dt if best is None else min(best, dt)
            out[key] = best * 1e3
        out["elems_per_s"] = n / (out["round_loop_ms"] * 1e-3)
        if with_fast:
            O.sumcheck_prove_fast(table[: 1 << min(lg, 16)], absorb=False, threads=threads)  # spin the thread pool up
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                O.sumcheck_prove_fast(table, absorb=False, threads=threads)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out["optimised_ms"] = best * 1e3
            out["optimised_elems_per_s"] = n / best
        return out

    samples = [sample(lg) for lg in sorted(set(extra_logs) | {log_n})]
    main = next(x for x in samples if x["log_n"] == log_n)
    if full_log is not None and full_log > log_n:
        samples.append(sample(full_log, with_absorb=False))
    n = 1 << log_n
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": main["elems_per_s"], "unit": "field elems/s", "cores": 1, "kind": "port",
        "sample": f"2^{log_n}-entry single-MLE sumcheck, round loop only (generate_round_polys): "
                  f"{main['round_loop_ms']:.0f} ms; reference entry point incl. table absorb "
                  f"(generate_sumcheck_proof): {main['with_table_absorb_ms']:.0f} ms "
                  f"= {n / (main['with_table_absorb_ms'] * 1e-3):.3g} elems/s; C restatement of the Rust prover, 1 thread",
        "optimised_variant": {"value": main["optimised_elems_per_s"], "unit": "field elems/s", "cores": threads, "kind": "port",
                              "sample": f"same 2^{log_n} round loop, in-place fused fold + OpenMP ({threads} threads): {main['optimised_ms']:.0f} ms"},
        "samples": samples,
        "samples_note": "per size: faithful restatement on 1 thread (round loop; with the reference's Keccak table absorb), and the optimised "
                        f"OpenMP variant on {threads} threads; a restatement, not the Rust/ark-ff binary (SURVEY.md 8d)",
        "cpu_model": cpu_model,
    }



# ------------------------------------------------------------------------------------------------ proof checks
def _fq(fn, a, b, np):
    from zk_amd._lib import lib, u64p

    out = np.empty(4, dtype=np.uint64)
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    getattr(lib(), fn)(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), out.ctypes.data_as(u64p))
    return out


def check_mle_proof(zk_amd, np, poly, base, last):
    """The verifier on the last timed proof, after the clock stopped (sumcheck/src/verifier.rs:29-62 partial_verify on a
    transcript in the prover's starting state + :15-26 perform_oracle_check: the table evaluated at all challenges on the
    device must equal g_last(r_last)).  Costs one more pass over the table."""
    claim, coeffs, ch = last
    t0 = time.perf_counter()
    ok, final, vch = zk_amd.SumcheckVerifier.partial_verify(zk_amd.SumCheckProof(claim, coeffs), base.clone())
    same_ch = bool(np.array_equal(vch, ch))
    at_r = poly.evaluate(list(ch)).get_evaluation_points()[0]
    oracle = bool(np.array_equal(at_r, final))
    total = bool(np.array_equal(poly.evaluation_sum(), claim))
    return {"verified": bool(ok and same_ch and oracle and total), "partial_verify": bool(ok), "challenges_match": same_ch,
            "oracle_check": oracle, "claim_is_table_sum": total, "check_ms": (time.perf_counter() - t0) * 1e3,
            "what": "SumcheckVerifier::partial_verify + perform_oracle_check (verifier.rs:15-62) on the last timed proof, outside the timed region"}


def check_sum2x2_proof(zk_amd, np, tabs, base, claim, last):
    """partial_verify (verifier.rs:29-62) on the last timed degree-2 proof, then the oracle check the caller of
    partial_verify owes (gkr/src/verifier.rs:106-113 in the reference's only use): f(r)g(r) + h(r)k(r) from four device
    evaluations must equal the final claim."""
    proof, ch = last
    t0 = time.perf_counter()
    ok, final, vch = zk_amd.SumcheckVerifier.partial_verify(zk_amd.SumCheckProof(claim, proof.round_polys), base.clone())
    same_ch = bool(np.array_equal(vch, ch))
    e = [t.evaluate(list(ch)).get_evaluation_points()[0] for t in tabs]
    at_r = _fq("zk_fq_add", _fq("zk_fq_mul", e[0], e[1], np), _fq("zk_fq_mul", e[2], e[3], np), np)
    oracle = bool(np.array_equal(at_r, final))
    return {"verified": bool(ok and same_ch and oracle), "partial_verify": bool(ok), "challenges_match": same_ch, "oracle_check": oracle,
            "check_ms": (time.perf_counter() - t0) * 1e3,
            "what": "SumcheckVerifier::partial_verify (verifier.rs:29-62) + f(r)g(r)+h(r)k(r) == final claim from four device evaluations, on the last "
                    "timed proof, outside the timed region; the claim handed to the prover is g_0(0)+g_0(1) of an untimed first prove "
                    "(generate_proof_for_partial_verify trusts its caller's claim, prover.rs:175-190)"}


# ------------------------------------------------------------------------------------------------ one rank
def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args))  # this process never initialises a GPU
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would be mislabelled")

    # roofline.traffic: the HBM counter passes run as CHILD processes before this one touches the GPU
    live_traffic = None
    if world == 1 and not args.force_sharded and args.workload in ("mle", "sum2x2") and not args.no_live_traffic:
        import torch  # noqa: F401  pages the libraries in (minutes on a fresh box) so the children's time limit is not spent on it; importing torch does not initialise the GPU
        live_traffic = live_pmc_traffic(args)

    # stdout carries exactly ONE JSON line: libraries (RCCL prints a version banner on fd 1) write to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch

    import zk_amd

    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist_mod

        if world == 1 and "MASTER_ADDR" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        dist = dist_mod
        if args.shared_device:
            if args.backend == "nccl" and world > 1:
                raise SystemExit("bench.py: --shared-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    if dist is not None:
        result = bench_sharded(args, zk_amd, np, dist, rank, local_rank, world)
    else:
        ctx = zk_amd.Context(local_rank)
        apply_knobs(args, ctx)
        if args.workload == "gkr":
            emit(bench_gkr(args, ctx, zk_amd, np, args.gkr_depth, args.gkr_log_width, args.steps, args.warmup))
            return
        result = bench_single(args, ctx, zk_amd, np, args.workload, args.log_n, args.steps, args.warmup, absorb=not args.no_absorb)
        if live_traffic is not None:
            attach_live_traffic(result["roofline"], result.get("kernels", {}), live_traffic)
        if not args.no_extra:
            result["extra"] = extra_workloads(args, ctx, zk_amd, np, torch)

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_ranks = dist.get_world_size() if dist is not None else 1
    rccl = dist is not None and (result.get("transport") == "rccl" or (args.sharded_impl == "py" and dist.get_backend() == "nccl"))
    out = {
        "metric": result.get("metric", METRIC),
        "value": result["elems_per_s"],
        "unit": result.get("unit", "field elems/s"),
        "n_gpus": n_ranks,  # the world size that actually ran
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": result["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak" if (n_ranks == 1 or args.weak) else "strong",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic (seeded splitmix64, full-range field elements)",
        "config": result["config"],
        "roofline": result["roofline"],
        "prove_ms": result["ms_per_step"],
        "backend": dist.get_backend() if dist is not None else None,
        "transport": result.get("transport"),
        "rccl_ranks": n_ranks if rccl else 0,
    }
    if dist is not None and not rccl:
        out["rccl_note"] = ("NO RCCL in this run: the ranks exchanged through a host all-gather over " + str(dist.get_backend()) +
                            " (a rehearsal of the multi-GPU path, not an xGMI measurement)")
    elif dist is not None and n_ranks == 1:
        out["rccl_note"] = "RCCL communicator of ONE rank: the exchange path ran, nothing crossed xGMI"
    for k in ("absorb_ms", "inclusive_ms", "kernels", "kernel_forms", "int32_alu", "survey_8d_equivalent", "proof_check", "sharded_impl_fallback",
              "gkr", "extra"):
        if k in result:
            out[k] = result[k]
    if not args.no_cpu_baseline and n_ranks == 1 and args.workload == "mle":
        out["cpu_baseline"] = cpu_baseline(args.cpu_log_n, full_log=args.cpu_full_log)
    else:
        out["cpu_baseline"] = None
    emit(out)
    if dist is not None:
        dist.destroy_process_group()


def apply_knobs(args, ctx):
    if args.fold_batch:
        ctx.set_fold_batch(args.fold_batch, args.fold_variant)
    if args.wide_batch >= 0:
        ctx.set_wide_fold(args.wide_batch, args.wide_min_log_out, args.wide_lds_below)
    if args.tail_max_log >= 0:
        ctx.set_tail_fold(args.tail_max_log, args.tail_out_below)
    if args.wide_mfma is not None:
        ctx.set_wide_mfma(args.wide_mfma, args.wide_mfma_wg, args.wide_mfma_depth)
    if args.deg2_rounds:
        ctx.set_deg2_rounds_per_pass(args.deg2_rounds)
    if args.deg2_lazy_from is not None:
        ctx.set_deg2_lazy_sums(args.deg2_lazy_from)
    if args.deg2_mfma_from is not None:
        ctx.set_deg2_mfma_sums(args.deg2_mfma_from)
    if args.deg2_fold3_fused is not None:
        ctx.set_deg2_fold3_fused(args.deg2_fold3_fused)
    if args.deg2_fused_below is not None:
        ctx.set_deg2_fused(args.deg2_fused_below)
    if args.deg2_small_triple is not None:
        ctx.set_deg2_small_triple(args.deg2_small_triple)
    if args.deg2_raw_cross is not None:
        ctx.set_deg2_raw_cross(bool(args.deg2_raw_cross))
    if args.dma_gates is not None:
        ctx.set_dma_gates(*args.dma_gates)
    if args.deg2_first_triple is not None:
        ctx.set_deg2_first_triple(args.deg2_first_triple[0], args.deg2_first_triple[1])
    if args.deg2_triple_from is not None or args.deg2_triple_ring is not None:
        ctx.set_deg2_triple(args.deg2_triple_from if args.deg2_triple_from is not None else 20, args.deg2_triple_ring or 4)


def read_kernels(ctx, names=KERNEL_NAMES):
    kernels = {}
    for name in names:
        p = ctx.prof_read(name)
        if p["launches"]:
            kernels[name] = {
                "launches": p["launches"], "total_ms": p["ms"], "avg_ms": p["ms"] / p["launches"],
                "alg_bytes_per_launch": p["alg_bytes"] / p["launches"],
                "achieved_GBs": p["alg_bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else None,
            }
    return kernels


_mad_peak = {}


def mad_peak(ctx) -> float:
    """lane multiply-adds per second of the v_mad_u64_u32 issue probe (measured live, once per context)."""
    key = id(ctx)
    if key not in _mad_peak:
        blocks, iters = 2048, 200  # 8 waves per SIMD
        ms = min(ctx.bench_mad(blocks, iters) for _ in range(2))
        _mad_peak[key] = blocks * 256 * iters * 256 / (ms * 1e-3)
    return _mad_peak[key]


def int32_alu(ctx, kernels, forms):
    """The int32-ALU roofline beside the HBM one: limb multiply-adds per second of the two product-heavy kernels against the
    measured v_mad_u64_u32 issue ceiling of the chip, and which of the two rooflines binds each."""
    peak = mad_peak(ctx)
    blocks, iters = 2048, 2000
    modmul_per_s = blocks * 256 * iters / (ctx.bench_modmul(blocks, iters) * 1e-3)
    hbm = {k: v["achieved_GBs"] / HBM_PEAK_GBS for k, v in kernels.items() if v["achieved_GBs"]}
    alu = {"peak_lane_mads_per_s": peak,
           "peak_note": "k_mad_probe: 2048 x 256 lanes, four independent v_mad_u64_u32 chains per lane, measured in this run",
           "modmul_per_s_dependent_chain": modmul_per_s}
    if "multifold_wide" in kernels:
        k = kernels["multifold_wide"]
        entries = k["alg_bytes_per_launch"] / 32.0 * 256 / 257
        alu["fold_pass_hbm_frac"] = hbm.get("multifold_wide")
        if "wide_mfma" in forms:
            # the contraction runs on the matrix cores: 64 product columns x 32 table bytes = 2048 i8 multiply-adds per entry (half of
            # them against the zeros of the banded weight matrix), against the dense i8 MFMA peak (2 x the bf16 rate: 5 POP/s)
            alu["fold_pass_pipe"] = "mfma_i8 (v_mfma_i32_16x16x64_i8)"
            alu["fold_pass_i8_macs_per_s"] = 2048 * entries / (k["avg_ms"] * 1e-3)
            alu["fold_pass_frac"] = alu["fold_pass_i8_macs_per_s"] / 2.5e15
            alu["fold_pass_peak_note"] = "2.5e15 dense i8 multiply-adds per second (MI355X_MICROARCH.md: i8 = 2 x bf16 per clock)"
        else:  # one 9x9-limb inner product (81 v_mad_u64_u32) per input entry
            alu["fold_pass_pipe"] = "valu (v_mad_u64_u32)"
            alu["fold_pass_limb_mads_per_s"] = 81 * entries / (k["avg_ms"] * 1e-3)
            alu["fold_pass_frac"] = alu["fold_pass_limb_mads_per_s"] / peak
        alu["fold_pass_binds"] = "hbm" if hbm.get("multifold_wide", 0) >= alu["fold_pass_frac"] else "int32"
    if "sum9" in kernels:            # nine-sum sweeps: 18 field products per quarter-index (16 entries of 32 B), 82 limb products each
        k = kernels["sum9"]
        prod = 18 * k["alg_bytes_per_launch"] * k["launches"] / 512.0
        alu["sum9_products_per_s"] = prod / (k["total_ms"] * 1e-3)
        alu["sum9_limb_mads_per_s"] = 82 * alu["sum9_products_per_s"]
        alu["sum9_frac"] = alu["sum9_limb_mads_per_s"] / peak
        alu["sum9_hbm_frac"] = hbm.get("sum9")
        alu["sum9_binds"] = "hbm" if hbm.get("sum9", 0) >= alu["sum9_frac"] else "int32"
    return alu


def bench_single(args, ctx, zk_amd, np, workload, log_n, steps, warmup, absorb=False, want_alu=True, check=True):
    import torch

    n = 1 << log_n
    absorb_ms = None
    if workload == "mle":
        poly = zk_amd.MultiLinearPolynomial.generate(SEED, n, ctx=ctx)
        base = zk_amd.Transcript()
        if absorb:
            # transcript.append(&init_polynomial.to_bytes()) (prover.rs:162): once, outside the timed region
            t0 = time.perf_counter()
            absorb_table(zk_amd, ctx, poly, base)
            absorb_ms = (time.perf_counter() - t0) * 1e3
        else:
            base.append(b"bench: table absorb skipped")

        def step():
            return zk_amd.SumcheckProver.generate_round_polys(poly, base.clone())

        kernel, what = "multifold", f"single-MLE sumcheck, 2^{log_n} entries, 1 GPU"
    else:
        tabs = [zk_amd.MultiLinearPolynomial.generate(SEED + k, n, ctx=ctx) for k in range(4)]
        f = zk_amd.SumPolynomial([zk_amd.ProductPolynomial(tabs[:2]), zk_amd.ProductPolynomial(tabs[2:])])
        base = zk_amd.Transcript()
        base.append(b"bench: degree-2 sum of two products")
        claim = np.zeros(4, dtype=np.uint64)

        def step():
            return zk_amd.SumcheckProver.generate_proof_for_partial_verify(claim, f, base.clone())

        # the true claim = g_0(0) + g_0(1) = 2 c0 + c1 + c2 of an untimed prove (round 0 does not depend on the claim passed in)
        c = step()[0].round_polys[0]
        claim = _fq("zk_fq_add", _fq("zk_fq_add", c[0], c[0], np), _fq("zk_fq_add", c[1], c[2], np), np)

        kernel = "fold4_eval3" if args.deg2_rounds == 1 else "fold2x4_sum9"
        what = f"degree-2 sumcheck Sum[Prod[f,g],Prod[h,k]], 4 x 2^{log_n} entries, 1 GPU"

    for _ in range(warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.forms_used()
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    ms_per_step = dt * 1e3 / steps
    proof_check = None
    if check and last is not None:
        proof_check = check_mle_proof(zk_amd, np, poly, base, last) if workload == "mle" else check_sum2x2_proof(zk_amd, np, tabs, base, claim, last)
    forms = sorted(ctx.forms_used())

    kernels = read_kernels(ctx)
    if kernel == "multifold":  # the fold pass the time goes to: the wide pass over the full table when there is one
        kernel = next(k for k in ("multifold_wide", "multifold", "multifold_tail", "block_sums") if k in kernels)
    if kernel not in kernels:
        kernel = max(kernels, key=lambda k: kernels[k]["total_ms"])
    dom = kernels[kernel]
    roofline = {
        "bound": "hbm", "kernel": kernel, "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": dom["achieved_GBs"] / HBM_PEAK_GBS,
        # filled by attach_live_traffic from this run's own counter passes (child processes); null with the committed passes named beside it otherwise
        "traffic": None, "traffic_source": pmc_traffic(kernel, log_n) if workload == "mle" else None,
        "note": "achieved = algorithmic bytes of all launches of the kernel in the timed steps / their summed "
                "HIP-event durations on the context's stream (multifold / multifold_wide: 32 B per input entry read + 32 B per output entry written; "
                "fold4_eval3: 48 B per input entry per table; fold2x4_sum9: 32 B per input + 32 B per output entry per table; sum9 / block_sums: 32 B per entry)",
    }
    res = {
        "elems_per_s": n / (ms_per_step * 1e-3), "ms_per_step": ms_per_step, "roofline": roofline, "kernels": kernels, "kernel_forms": forms,
        "config": {"workload": what, "log_n": log_n, "field": "BN254 Fq",
                   "parallelism": "1 GPU", "transcript": "Keccak-256 on host, one challenge per round"},
    }
    if proof_check is not None:
        res["proof_check"] = proof_check
    if want_alu:
        res["int32_alu"] = int32_alu(ctx, kernels, forms)
    if workload == "mle":
        # SURVEY.md 8(d) prices a whole prove at 128*N bytes (one fused fold + sum per round).  Batching rounds REMOVES
        # traffic (this build moves ~64*N), so that figure over the prove time exceeds what HBM delivers: it is reported
        # as an equivalent rate only; roofline.achieved above counts the bytes the dominant kernel really has to move.
        res["survey_8d_equivalent"] = {"bytes_per_prove": 128.0 * n, "GBs": 128.0 * n / (ms_per_step * 1e-3) / 1e9,
                                       "note": "128*N bytes of the one-round-per-pass schedule / prove time; > HBM peak is possible because rounds are batched"}
    if absorb_ms is not None:
        res["absorb_ms"] = absorb_ms
        res["inclusive_ms"] = absorb_ms + ms_per_step
    return res


def extra_workloads(args, ctx, zk_amd, np, torch):
    """The other workloads north_star asks the benchmark to report, in the same driver-run line: a size sweep of the
    single-MLE prove, the degree-2 prove at 4 x 2^26 (config 4's shape on one GPU) and the GKR config-5 prove.  Three timed
    steps each; every entry carries its prove time, rate and the HBM fraction of its dominant kernel."""
    import gc

    out = {"note": "steps 3, warmup 1 each (size sweep at 2^24 and below: 20 and 3); single GPU; same library calls and timing brackets as the headline workload"}
    gc.collect()
    sweep = []
    sizes = [20, 24, 28]
    free, _total = torch.cuda.mem_get_info()
    if free > (45 << 30):
        sizes.append(30)  # 32 GiB table
    for log_n in sizes:
        k_steps, k_warm = (20, 3) if log_n <= 24 else (3, 1)  # sub-millisecond proves: three steps are dominated by first-touch effects
        r = bench_single(args, ctx, zk_amd, np, "mle", log_n, k_steps, k_warm, absorb=False, want_alu=False)
        sweep.append({"log_n": log_n, "steps": k_steps, "prove_ms": r["ms_per_step"], "elems_per_s": r["elems_per_s"], "dominant_kernel": r["roofline"]["kernel"],
                      "hbm_frac": r["roofline"]["frac"], "verified": r["proof_check"]["verified"],
                      "kernels": {k: {"avg_ms": v["avg_ms"], "launches_per_prove": v["launches"] / k_steps, "GBs": v["achieved_GBs"]} for k, v in r["kernels"].items()}})
        gc.collect()
    out["mle_sweep"] = sweep
    r = bench_single(args, ctx, zk_amd, np, "sum2x2", 26, 3, 1)
    out["sum2x2_log26"] = {"workload": r["config"]["workload"], "prove_ms": r["ms_per_step"], "entries_per_s": 4 * (1 << 26) / (r["ms_per_step"] * 1e-3),
                           "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "peak", "unit", "frac")},
                           "kernels": r["kernels"], "int32_alu": r.get("int32_alu"), "proof_check": r["proof_check"]}
    gc.collect()
    g = bench_gkr(args, ctx, zk_amd, np, 16, 20, 3, 1)
    out["gkr_config5"] = {k: g[k] for k in ("config", "prove_ms", "value", "unit", "layers_ms_total", "ms_per_round", "circuit_evaluate_ms",
                                            "output_absorb_ms", "host_other_ms", "proof_check", "note")}
    return out


def setup_ranks(args, zk_amd, dist, local_rank):
    """This rank's context and its communicator.  --sharded-impl lib: the library's own communicator (RCCL with the nccl backend, the
    host all-gather callback otherwise).  zk_amd.sharded.comm_init votes before any id exchange, so the ranks fail together; if they
    do (ZK_E_COMM: RCCL missing or refusing), every rank falls back to the torch.distributed loop of zk_amd/sharded.py and the line
    says so (`sharded_impl_fallback`).  -> (ctx, use_lib, transport, fallback)"""
    import torch

    from zk_amd import sharded

    use_lib = args.sharded_impl == "lib"
    ctx = zk_amd.Context(local_rank) if use_lib else sharded.torch_stream_context(local_rank)
    apply_knobs(args, ctx)
    fallback, transport = None, None
    if use_lib:
        try:
            transport, err = sharded.comm_init(ctx, dist), None
        except Exception as e:  # noqa: BLE001 -- reported in the JSON line
            transport, err = None, repr(e)
        # ncclCommInitRank itself can still fail on one rank only (after the vote inside comm_init): agree on the outcome
        flag = torch.tensor([0 if err else 1], dtype=torch.int32,
                            device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if err is None:
                sharded.comm_destroy(ctx)
            fallback = err or "another rank could not set up the library communicator"
            use_lib = False
            ctx = sharded.torch_stream_context(local_rank)
            apply_knobs(args, ctx)
    if not use_lib:
        transport = "rccl" if dist.get_backend() == "nccl" else "host"
    return ctx, use_lib, transport, fallback


def digests_agree(np, torch, dist, arrays, dev) -> bool:
    """every rank must hold the same bytes: MIN and MAX over ranks of a SHA-256 of them coincide"""
    import hashlib

    digest = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in arrays)).digest()
    dg = torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(torch.int64).to(dev)
    lo, hi = dg.clone(), dg.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def bench_gkr_sharded(args, ctx, zk_amd, np, dist, rank, world, transport):
    """GKR config 5 over the ranks (zk_gkr_prove_sharded; gkr/src/prover.rs:24-154): circuit and inputs replicated, every layer's
    two sumcheck phases sharded on the low index bits.  The same circuit on every rank ("strong": total work fixed)."""
    import torch

    from zk_amd import sharded
    from zk_amd.synthetic import gkr_layers

    depth, k = args.gkr_depth, args.gkr_log_width
    circuit = zk_amd.Circuit.from_arrays(gkr_layers(depth, k), ctx=ctx)
    inputs = zk_amd.MultiLinearPolynomial.generate(SEED, 1 << k, ctx=ctx)
    gather_log = args.gather_log if args.gather_log >= 0 else 0

    def step(timing=False):
        return zk_amd.GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs, timing=timing, sharded_gather_log=gather_log)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    proof = None
    for _ in range(args.steps):
        proof = step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    dev = torch.device("cuda", ctx.device_id) if dist.get_backend() == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ms = float(tmax.item()) * 1e3 / args.steps
    same = digests_agree(np, torch, dist, [proof.elements()], dev)
    timed = step(timing=True)
    verified = None
    if rank == 0:   # the verifier on rank 0's last timed proof (gkr/src/verifier.rs:21-119), single GPU, outside the timed region
        verified = bool(zk_amd.GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof))
    dist.barrier()
    sharded.comm_destroy(ctx)
    gates = depth << k
    rounds = sum(p.round_polys.shape[0] for p in proof.sumcheck_proofs)
    layers_ms = sum(timed.layer_ms[:depth])
    return {
        "elems_per_s": gates / (ms * 1e-3), "unit": "gates/s", "ms_per_step": ms,
        "metric": "GKR prove ms + gates/s, layered circuit, wiring predicates built on device",
        "roofline": None, "transport": transport, "sharded_impl_fallback": None,
        "proof_check": {"all_ranks_identical": same, "verified": verified,
                        "what": "SHA-256 of every rank's GKRProof elements agree; GKRVerifier::verify_proof accepts rank 0's last timed proof"},
        "gkr": {"layers_ms_total": layers_ms, "ms_per_round": layers_ms / rounds, "circuit_evaluate_ms": timed.layer_ms[depth],
                "output_absorb_ms": timed.layer_ms[depth + 1], "sumcheck_rounds": rounds,
                "note": "latency-bound at this size (32 MiB of tables per layer): sharding adds an exchange per pass and removes no time (DESIGN.md 6); "
                        "the line exists so an N-GPU box reports the GKR prove north_star asks for; the W_0 Keccak absorb is per rank and serial"},
        "config": {"workload": f"GKR prove, depth {depth}, 2^{k} gates per layer, every layer's sumcheck sharded over {world} GPUs (circuit replicated)",
                   "depth": depth, "log_width": k, "field": "BN254 Fq", "parallelism": f"layer tables sharded x{world}",
                   "exchange": f"all-gather of nine sums per two rounds; tables of <= 2^{gather_log} entries per rank gathered and finished on every rank"},
    }


def bench_sharded(args, zk_amd, np, dist, rank, local_rank, world):
    """One rank per GPU; the 2^log_n-entry table is sharded on the variables bound last.  Strong scaling by default (the
    metric's 2^28 entries in total).  --sharded-impl lib (default): the loop and the exchange run inside libzkmi.so
    (zk_sumcheck_prove_*_sharded: RCCL ncclAllGather on the context stream with the nccl backend, host all-gather through
    torch.distributed otherwise); py: zk_amd/sharded.py over torch.distributed."""
    import torch

    from zk_amd import sharded

    g = world.bit_length() - 1
    if world != 1 << g:
        raise SystemExit("world size must be a power of two")
    ctx, use_lib, transport, fallback = setup_ranks(args, zk_amd, dist, local_rank)
    if args.workload == "gkr":
        if not use_lib:
            raise SystemExit(f"multi-GPU GKR runs inside the library only (zk_gkr_prove_sharded); its communicator could not be set up: {fallback}")
        return bench_gkr_sharded(args, ctx, zk_amd, np, dist, rank, world, transport)
    log_total = args.log_n + (g if args.weak else 0)
    n_local = 1 << (log_total - g)
    gather_log = args.gather_log if args.gather_log >= 0 else None
    base = zk_amd.Transcript()
    base.append(b"bench: table absorb skipped (sharded run; SURVEY.md H1)")
    if use_lib:
        via = "in-library loop, " + ("ncclAllGather on the context stream" if transport == "rccl" else f"host all-gather through {dist.get_backend()}")
    else:
        via = f"zk_amd/sharded.py over torch.distributed ({dist.get_backend()})"
    if args.workload == "mle":
        poly = zk_amd.MultiLinearPolynomial.generate(SEED, n_local, first=rank, step=world, ctx=ctx)

        def step():
            if use_lib:
                return sharded.prove_mle_sharded(poly, base.clone(), False, gather_log)
            sh = sharded.DeviceShard(poly)
            try:
                return sharded.generate_round_polys_sharded(sh, base.clone(), world, dist, None, None, gather_log)
            finally:
                sh.close()

        dom_kernel, names = "multifold", ("block_sums", "multifold_wide", "multifold", "multifold_tail")
        what = f"single-MLE sumcheck, 2^{log_total} entries"
        exchange = f"all-gather of 2^m block sums per pass ({via}), library pass schedule; tables of <= 2^{args.gather_log} entries per rank all-gathered and finished on every rank"
    else:
        tabs = [zk_amd.MultiLinearPolynomial.generate(SEED + k, n_local, first=rank, step=world, ctx=ctx) for k in range(4)]

        def step():
            if use_lib:
                return sharded.prove_sum2x2_sharded(tabs, base.clone(), None, gather_log)
            sh = sharded.DeviceShard4(tabs)
            try:
                return sharded.generate_round_polys_sum2x2_sharded(sh, base.clone(), world, dist, None, gather_log)
            finally:
                sh.close()

        dom_kernel, names = ("fold4_eval3", ("eval3", "fold4_eval3")) if gather_log is None else ("fold2x4_sum9", ("sum9", "fold2x4_sum9"))
        what = f"degree-2 sumcheck Sum[Prod[f,g],Prod[h,k]], 4 x 2^{log_total} entries"
        exchange = (f"all-gather of 3 evaluation shares per round ({via})" if gather_log is None else
                    f"all-gather of nine sums per two rounds ({via}); tables of <= 2^{gather_log} entries per rank all-gathered and finished on every rank")

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_collect()
    ctx.prof_reset()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    ctx.prof_collect()
    ctx.prof_enable(False)
    dev = torch.device("cuda", ctx.device_id) if dist.get_backend() == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    same = digests_agree(np, torch, dist, last, dev)   # every rank must hold the same (claim, round polys, challenges)
    ms_per_step = float(tmax.item()) * 1e3 / args.steps

    kernels = read_kernels(ctx, names)
    if dom_kernel == "multifold":
        dom_kernel = next(k for k in ("multifold_wide", "multifold", "multifold_tail", "block_sums") if k in kernels)
    dom = kernels[dom_kernel]
    roofline = {
        "bound": "hbm", "kernel": dom_kernel, "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None,
        "note": "rank 0's launches (per GPU): algorithmic bytes of all launches of the kernel in the timed steps / "
                "their summed HIP-event durations (multifold: 32 B per input entry read + 32 B per output entry written; "
                "fold4_eval3: 48 B per input entry per table)",
    }
    if use_lib:
        sharded.comm_destroy(ctx)
    return {
        "elems_per_s": (1 << log_total) / (ms_per_step * 1e-3), "ms_per_step": ms_per_step, "roofline": roofline,
        "kernels": kernels, "proof_check": {"all_ranks_identical": same}, "transport": transport, "sharded_impl_fallback": fallback,
        "config": {"workload": f"{what} sharded over {world} GPUs (2^{log_total - g} per GPU, low-bit sharding)",
                   "log_n": log_total, "field": "BN254 Fq", "parallelism": f"table-sharded x{world}",
                   "exchange": exchange,
                   "transcript": "Keccak-256 on every host, one challenge per round"},
    }


def bench_gkr(args, ctx, zk_amd, np, depth, k, steps, warmup):
    """Full GKR prove (gkr/src/prover.rs:24-154) of a layered circuit: circuit evaluation, predicate
    folding from the gate lists and every layer's degree-2 sumcheck on the device."""
    import torch

    from zk_amd.synthetic import gkr_layers

    t0 = time.perf_counter()
    circuit = zk_amd.Circuit.from_arrays(gkr_layers(depth, k), ctx=ctx)
    build_ms = (time.perf_counter() - t0) * 1e3
    inputs = zk_amd.MultiLinearPolynomial.generate(SEED, 1 << k, ctx=ctx)

    def step(timing=False):
        return zk_amd.GKRProver.generate_proof(circuit, zk_amd.Transcript(), inputs, timing=timing)

    for _ in range(warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof = step()
    ctx.synchronize()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    t0 = time.perf_counter()
    timed = step(timing=True)
    timed_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    verified = bool(zk_amd.GKRVerifier.verify_proof(inputs, circuit, zk_amd.Transcript(), proof))   # gkr/src/verifier.rs:21-119, outside the timed region
    verify_ms = (time.perf_counter() - t0) * 1e3
    gates = depth << k
    rounds = sum(p.round_polys.shape[0] for p in proof.sumcheck_proofs)
    return {
        "metric": "GKR prove ms + gates/s, layered circuit, wiring predicates built on device", "value": gates / (ms * 1e-3),
        "unit": "gates/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
        "data": "synthetic (SURVEY.md 8d config 5 wiring, seeded inputs)",
        "config": {"workload": f"GKR prove, depth {depth}, 2^{k} gates per layer", "depth": depth, "log_width": k,
                   "sumcheck_rounds": rounds, "field": "BN254 Fq"},
        "prove_ms": ms, "layer_ms": timed.layer_ms[:depth], "layers_ms_total": sum(timed.layer_ms[:depth]),
        "ms_per_round": sum(timed.layer_ms[:depth]) / rounds, "circuit_evaluate_ms": timed.layer_ms[depth],
        "output_absorb_ms": timed.layer_ms[depth + 1], "circuit_build_ms_host": build_ms,
        # what the three timers inside zk_gkr_prove do not cover, on the SAME (timed) call: argument marshalling in Python, the proof
        # arrays' allocation, the join of the output download thread, the final transcript bookkeeping
        "host_other_ms": timed_ms - sum(timed.layer_ms[:depth + 2]), "timed_call_ms": timed_ms,
        "proof_check": {"verified": verified, "verify_ms": verify_ms,
                        "what": "GKRVerifier::verify_proof (zk_gkr_verify) accepts the last timed proof; outside the timed region"},
        "note": "prove_ms includes the reference's serial Keccak absorb of the 2^k-entry output layer (prover.rs:46, "
                "output_absorb_ms); the layer sumchecks are latency-bound: 2*k_in sequential rounds per layer, each kernel + 96-byte D2H + host Keccak; tables are "
                "2^k_in entries (32 MiB at k = 20), far below the HBM roofline regime",
    }


def live_pmc_traffic(args):
    """HBM bytes of this very command from the PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: two
    SEPARATE rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE; no tracing options beside them), each a child run of
    `python3 bench.py <same flags> --steps 1 --warmup 1` without the CPU legs, started before this process initialises the GPU.
    Returns tools/prof_summary.py's per-kernel dictionary, or {"error": reason}: the line then keeps traffic = null and names the
    committed counter file instead.  Every failure mode ends the attempt (time limit with SIGKILL, non-zero exit, no CSV)."""
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    if not os.path.exists("/dev/kfd"):
        return {"error": "no GPU device node (/dev/kfd)"}
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "this run is itself under a profiler (its preloaded library has initialised the GPU: no child may be started from here)"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import prof_summary
    except ImportError as e:
        return {"error": f"tools/prof_summary.py: {e}"}
    child = ["python3", os.path.abspath(__file__)] + sys.argv[1:] + ["--no-absorb", "--no-cpu-baseline", "--no-extra", "--no-live-traffic", "--steps", "1", "--warmup", "1"]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="zk_pmc_", dir="/tmp") as tmp:
        dirs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["timeout", "-s", "KILL", "180", prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
            except (OSError, subprocess.TimeoutExpired) as e:
                return {"error": f"{counter} pass: {type(e).__name__}: {e}"}
            if r.returncode != 0:
                return {"error": f"{counter} pass: exit {r.returncode}: " + r.stderr.decode(errors="replace")[-300:]}
            dirs[counter] = d
        try:
            out = prof_summary.traffic_dict(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], args.log_n)
        except (SystemExit, Exception) as e:  # no CSV under the output directory, or a column missing
            return {"error": f"counter CSV: {e}"}
    out["seconds"] = time.perf_counter() - t0
    out["command"] = "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> -- " + " ".join(child[:1] + ["bench.py"] + child[2:])
    return out


def attach_live_traffic(roofline, kernels, live):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from THIS run's counter passes (read = 2 x FETCH_SIZE KiB,
    write = WRITE_SIZE KiB: the guide's gfx950 correction), next to the algorithmic bytes per launch `achieved` is priced on."""
    if "error" in live:
        roofline["traffic_note"] = "live counter passes failed (" + live["error"] + "); traffic_source names the committed passes of this command"
        return
    k = live["kernels"].get(roofline["kernel"])
    if not k or not k["launches"]:
        roofline["traffic_note"] = "live counter passes hold no launch of " + roofline["kernel"]
        return
    roofline["traffic"] = k["bytes_per_launch"]
    alg = kernels.get(roofline["kernel"], {}).get("alg_bytes_per_launch")
    roofline["traffic_source"] = {
        "live": True, "command": live["command"], "seconds": round(live["seconds"], 1), "correction": live["correction"],
        "launches_counted": k["launches"], "read_bytes_per_launch": k["read_bytes"] / k["launches"], "write_bytes_per_launch": k["write_bytes"] / k["launches"],
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (k["bytes_per_launch"] / alg) if alg else None,
        "all_kernels": {n: v["bytes_per_launch"] for n, v in live["kernels"].items()},
    }


def pmc_traffic(kernel: str, log_n: int):
    """The committed rocprofv3 counter passes of this same command (profiles/*hbm_traffic.json): HBM bytes per launch of
    `kernel` and the file they come from.  Not this run's counters -- PMC cannot be read from inside the process."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("log_n") == log_n and kernel in d.get("kernels", {}):
            return {"file": os.path.relpath(path, ROOT) + " (committed)", "bytes_per_launch": d["kernels"][kernel]["bytes_per_launch"]}
    return None


def absorb_table(zk_amd, ctx, poly, transcript):
    """to_bytes() of the device table streamed through the host Keccak sponge (done inside the library)."""
    from zk_amd._lib import lib

    n = len(poly)
    chunk = 1 << 22
    if n <= chunk:
        transcript.append(poly.to_bytes())
        return
    for off in range(0, n, chunk):
        part = zk_amd.MultiLinearPolynomial.wrap_device(lib().zk_mle_device_ptr(poly.h) + off * 32, chunk, ctx=ctx)
        transcript.append(part.to_bytes())


if __name__ == "__main__":
    main()
