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
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["roofline"]["traffic"] is None
    assert 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    alu = d["int32_alu"]
    assert alu["peak_lane_mads_per_s"] > 1e13
    for k in ("fold_pass_frac", "sum9_frac"):
        if k in alu:
            assert 0 < alu[k] <= 1.0, (k, alu[k])
