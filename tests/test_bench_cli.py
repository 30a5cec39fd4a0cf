"""bench.py's launcher logic: `--gpus N` must never print a line labelled N GPUs unless N ranks ran."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr.decode()


def test_gpus_flag_disagreeing_with_the_launcher_is_refused():
    rc, lines, err = _run(["--gpus", "2", "--no-extra"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "mislabelled" in err
    rc, lines, err = _run(["--gpus", "1", "--no-extra"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "mislabelled" in err


def test_gpus_n_without_gpus_prints_no_line():
    """here (no GPU) the ranks bench.py starts cannot create a context: the parent must fail, not print a 1-GPU line labelled 2"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    rc, lines, err = _run(["--gpus", "2", "--backend", "gloo", "--shared-device", "--log-n", "12", "--steps", "1", "--warmup", "0", "--no-extra"])
    assert rc != 0 and not lines


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_and_labels_the_line_with_the_world_size():
    rc, lines, err = _run(["--gpus", "2", "--backend", "gloo", "--shared-device", "--log-n", "20", "--steps", "2", "--warmup", "1", "--no-extra"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["proof_check"]["all_ranks_identical"] is True
    assert d["backend"] == "gloo" and d["transport"] == "host" and d["rccl_ranks"] == 0
    assert d["config"]["log_n"] == 20 and "sharded over 2 GPUs" in d["config"]["workload"]


@pytest.mark.gpu
def test_gpus_1_line_has_the_contract_keys():
    rc, lines, err = _run(["--gpus", "1", "--log-n", "20", "--steps", "2", "--warmup", "1", "--no-extra", "--no-absorb", "--cpu-log-n", "16"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None
    # roofline.traffic: this run's own two rocprofv3 counter passes (child processes), or null with the reason beside it
    rf = d["roofline"]
    if rf["traffic"] is None:
        assert "traffic_note" in rf, rf
    else:
        src = rf["traffic_source"]
        assert src["live"] is True and rf["traffic"] > 0 and src["launches_counted"] > 0, rf
        assert abs(src["read_bytes_per_launch"] + src["write_bytes_per_launch"] - rf["traffic"]) <= 1e-6 * rf["traffic"]
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    alu = d["int32_alu"]
    assert alu["peak_lane_mads_per_s"] > 1e13
    for k in ("fold_pass_frac", "sum9_frac"):
        if k in alu:
            assert 0 < alu[k] <= 1.0, (k, alu[k])


@pytest.mark.gpu
def test_gpus_2_gkr_line_through_the_sharded_layer_prover():
    """bench.py --workload gkr --gpus N runs zk_gkr_prove_sharded on every rank (here: two gloo ranks sharing the GPU, a small
    circuit), all ranks end with the same proof and rank 0's is accepted by the verifier."""
    rc, lines, err = _run(["--gpus", "2", "--backend", "gloo", "--shared-device", "--workload", "gkr", "--gkr-depth", "3", "--gkr-log-width", "10",
                           "--gather-log", "5", "--steps", "2", "--warmup", "1", "--no-extra"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["unit"] == "gates/s" and d["metric"].startswith("GKR prove")
    assert d["proof_check"]["all_ranks_identical"] is True and d["proof_check"]["verified"] is True
    assert d["rccl_ranks"] == 0 and "NO RCCL" in d["rccl_note"] and d["cpu_baseline"] is None
    assert d["gkr"]["sumcheck_rounds"] == 3 * 20 and "sharded over 2 GPUs" in d["config"]["workload"]


@pytest.mark.gpu
def test_single_gpu_lines_carry_a_proof_check():
    rc, lines, err = _run(["--gpus", "1", "--log-n", "18", "--steps", "2", "--warmup", "1", "--no-extra", "--no-absorb", "--no-cpu-baseline"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    pc = json.loads(lines[0])["proof_check"]
    assert pc["verified"] is True and pc["oracle_check"] is True and pc["challenges_match"] is True and pc["claim_is_table_sum"] is True
    rc, lines, err = _run(["--gpus", "1", "--workload", "sum2x2", "--log-n", "16", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    pc = json.loads(lines[0])["proof_check"]
    assert pc["verified"] is True and pc["oracle_check"] is True
    rc, lines, err = _run(["--gpus", "1", "--workload", "gkr", "--gkr-depth", "3", "--gkr-log-width", "10", "--steps", "2", "--warmup", "1"])
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = json.loads(lines[0])
    assert d["proof_check"]["verified"] is True and "host_other_ms" in d
