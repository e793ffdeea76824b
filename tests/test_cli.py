"""The compiled host (hyperfridge-r0_amd/r0h_prove, C++ over the C ABI, no Python in the loop)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, circuit_path

CLI = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_prove")
VERIFY = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_verify")


def test_verify_cli_accepts_the_golden_seal_and_rejects_a_flipped_word(tmp_path):
    """`verifier verify` counterpart (verifier/src/main.rs:118-126): host only, exit status carries the outcome."""
    seal = np.load(os.path.join(ROOT, "tests", "golden", "seal_tiny_po2_9_seed_1.npy"))
    good, bad = str(tmp_path / "good.bin"), str(tmp_path / "bad.bin")
    seal.tofile(good)
    flipped = seal.copy()
    flipped[-1] ^= 1
    flipped.tofile(bad)
    out = subprocess.run([VERIFY, circuit_path("tiny"), good], capture_output=True, text=True)
    assert out.returncode == 0 and json.loads(out.stdout)["accepted"] is True and json.loads(out.stdout)["po2"] == 9
    out = subprocess.run([VERIFY, circuit_path("tiny"), bad], capture_output=True, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["accepted"] is False
    out = subprocess.run([VERIFY, good, good], capture_output=True, text=True)  # a seal is not a circuit blob
    assert out.returncode == 2 and "circuit blob" in out.stderr


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
    out = subprocess.run([CLI, circuit_path("small"), "--po2", "11", "--seed", "9", "--seal-out", seal_file, "--verify", "1"], capture_output=True, text=True)
    assert out.returncode == 0 and "seals verified" in out.stderr, out.stderr
    assert subprocess.run([VERIFY, circuit_path("small"), seal_file], capture_output=True).returncode == 0
    info = json.loads(out.stdout.strip().splitlines()[-1])
    seal = np.fromfile(seal_file, dtype=np.uint32)
    assert info["seal_words"] == seal.size
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    assert orc.circuit(blob).verify(seal) == (0, "ok")
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, 11, 9)
    assert np.array_equal(seal, hal.prove_segment(gc, 11, code, data, glob))


@pytest.mark.gpu
def test_prove_to_receipt_json_then_verify_like_the_reference_verifier(tmp_path):
    """host writes a Receipt JSON (host/src/main.rs:251-316), verifier reads and verifies it (verifier/src/main.rs:114-126)."""
    receipt = str(tmp_path / "receipt.json")
    commitment = json.dumps({"hostinfo": "host:main", "iban": "CH4308307000289537312", "stmts": [{"elctrnc_seq_nb": "247"}]}, separators=(",", ":"))
    out = subprocess.run([CLI, circuit_path("small"), "--po2", "10", "--segments", "3", "--contexts", "2", "--receipt-out", receipt, "--journal", commitment],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    ids = [json.loads(ln) for ln in out.stdout.splitlines() if "control_root" in ln][0]
    doc = json.load(open(receipt))
    segs = doc["inner"]["Composite"]["segments"]
    assert [s["index"] for s in segs] == [0, 1, 2] and segs[2]["claim"]["exit_code"] == {"Halted": 0} and segs[0]["claim"]["exit_code"] == "SystemSplit"
    bind = ["--image-id", ids["image_ids"][0], "--control-root", "%d:%s" % (ids["control_root"]["po2"], ",".join(map(str, ids["control_root"]["root"])))]
    out = subprocess.run([VERIFY, "--receipt", receipt, circuit_path("small")] + bind, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    report = json.loads(out.stdout)
    assert report["accepted"] is True and report["journal_bound"] is True and report["segments"] == 3 and report["commitment"] == commitment
    # the same seals with another journal, or checked against another image id, are refused
    doc["journal"]["bytes"][-6] ^= 1
    json.dump(doc, open(receipt, "w"), separators=(",", ":"))
    out = subprocess.run([VERIFY, "--receipt", receipt, circuit_path("small")] + bind, capture_output=True, text=True)
    assert out.returncode == 1 and "journal" in json.loads(out.stdout)["reason"]
    out = subprocess.run([VERIFY, "--receipt", receipt, circuit_path("small")], capture_output=True, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["seals_valid"] is True and json.loads(out.stdout)["journal_bound"] is False


@pytest.mark.gpu
def test_batch_of_receipts_on_a_work_queue(tmp_path):
    """BASELINE.json configs[3] in small: receipts x segments units, lanes take the next unit when free; every seal kept for a
    receipt file must still be there, in order, and verify."""
    out = subprocess.run([CLI, circuit_path("small"), "--po2", "10", "--segments", "3", "--receipts", "4", "--contexts", "3", "--verify", "1"],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "seals verified" in out.stderr, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["receipts"] == 4 and info["segments"] == 3 and info["receipts_per_s"] > 0 and abs(info["segments_per_s"] / info["receipts_per_s"] - 3) < 1e-3
    bad = subprocess.run([CLI, circuit_path("small"), "--receipts", "2", "--receipt-out", str(tmp_path / "r.json")], capture_output=True, text=True)
    assert bad.returncode == 1 and "--receipt-dir" in bad.stderr
