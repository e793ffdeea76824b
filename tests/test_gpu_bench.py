"""bench.py keeps the driver's contract: one JSON line with the agreed keys (checked on a small, fast configuration)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--po2", "12",
                          "--circuit", "small", "--cpu-po2", "10", "--contexts", "2"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "segments/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "u32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.02  # value = segments / time
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_the_rccl_path_of_the_driver_runs_with_one_rank():
    """The multi-GPU runs go through torch.distributed with backend "nccl" (= RCCL): process group bound to the rank's device, barrier
    around the timed region, MAX / SUM all-reduces of device tensors.  A one-GPU box cannot hold two RCCL ranks, but it can hold one:
    R0H_FORCE_PROCESS_GROUP=1 builds the group for a single rank, so the very calls the 8-GPU run makes are executed here."""
    env = dict(os.environ, R0H_FORCE_PROCESS_GROUP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--po2", "12", "--circuit", "small",
                          "--cpu-po2", "0", "--contexts", "2", "--backend", "nccl"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["steps"] == 2
