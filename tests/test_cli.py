"""The compiled host (hyperfridge-r0_amd/r0h_prove, C++ over the C ABI, no Python in the loop)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, circuit_path

CLI = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_prove")


def test_cli_builds_and_reports_usage():
    out = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "usage: r0h_prove" in out.stdout and "gfx950" in out.stdout


def test_cli_without_a_gpu_fails_loudly():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    out = subprocess.run([CLI, circuit_path("tiny"), "--po2", "9"], capture_output=True, text=True)
    assert out.returncode == 2 and "r0h_ctx_create" in out.stderr


@pytest.mark.gpu
def test_cli_seal_matches_the_harness_and_verifies(hal, orc, tmp_path):
    seal_file = str(tmp_path / "seal.bin")
    out = subprocess.run([CLI, circuit_path("small"), "--po2", "11", "--seed", "9", "--seal-out", seal_file], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    seal = np.fromfile(seal_file, dtype=np.uint32)
    assert info["seal_words"] == seal.size
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    assert orc.circuit(blob).verify(seal) == (0, "ok")
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, 11, 9)
    assert np.array_equal(seal, hal.prove_segment(gc, 11, code, data, glob))
