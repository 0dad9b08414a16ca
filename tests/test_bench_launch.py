"""`python bench.py --gpus N` the way the driver invokes it (no launcher, WORLD_SIZE unset): bench.py itself starts the
N ranks as child processes and rank 0's JSON line is the last line of stdout. Run here in the rehearsal mode (CPU
tensors, gloo, the oracle injected as the compute backend of csmpn_hip.sharded): what is checked is the launch path,
the partition bookkeeping and the line's contract, not a number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
           "--rehearse-backend", "oracle_backend:OracleBackend", "--rehearse-size", "48,400"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    return json.loads(lines[-1])      # the JSON line must be the LAST line of stdout


@pytest.mark.parametrize("partition,scaling", [("auto", "weak"), ("auto", "strong"), ("A", "weak")])
def test_bench_spawns_its_own_ranks(partition, scaling):
    line = _run(["--gpus", "2", "--dist-backend", "gloo", "--partition", partition, "--scaling", scaling])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == scaling and line["data"] == "rehearsal"
    cfg = line["config"]
    total = 800 if scaling == "weak" else 400
    assert sum(cfg["edges_per_rank"]) == total          # every adjacency on exactly one rank
    if partition == "auto":                             # the default is the destination partition
        assert cfg["partition"] == "B"
        assert 1.0 <= cfg["pad_ratio"] <= 1.5 and 0.0 <= cfg["local_share"] <= 1.0
    else:
        assert cfg["partition"] == "A" and cfg["pad_ratio"] is None
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_bench_rejects_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-backend",
                          "oracle_backend:OracleBackend"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
