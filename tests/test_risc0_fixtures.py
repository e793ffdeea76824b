"""Real risc0 vectors, when somebody has them (tests/fixtures/risc0/README.md).  Until then every test here skips with the
words "parity unpinned": the composed seal has never been compared with one made by risc0 3.0.5."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import ROOT

SLOT = os.path.join(ROOT, "tests", "fixtures", "risc0")


def _need(*names):
    missing = [n for n in names if not os.path.exists(os.path.join(SLOT, n))]
    if missing:
        pytest.skip("parity unpinned: no risc0 vectors in tests/fixtures/risc0 (%s missing)" % ", ".join(missing))
    return [os.path.join(SLOT, n) for n in names]


def test_poseidon2_table_equals_risc0s():
    (path,) = _need("poseidon2_consts.json")
    theirs = json.load(open(path))
    ours = json.load(open(os.path.join(ROOT, "tests", "golden", "poseidon2_babybear_t24.json")))
    assert theirs["round_constants"] == ours["round_constants"] and theirs["int_diag_m1"] == ours["int_diag_m1"]


def _imported_blob(tmp_path):
    taps, poly, info = _need("taps.rs", "poly_ext.rs", "circuit_info.txt")
    out = str(tmp_path / "risc0.r0c")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "import_risc0_circuit.py"), taps, poly, out, "--info", open(info).read().strip("\n")])
    return np.fromfile(out, dtype=np.uint32)


def test_risc0_circuit_tables_import_and_load(orc, tmp_path):
    blob = _imported_blob(tmp_path)
    c = orc.circuit(blob)
    assert c.n_taps > 0
    verdict, reason, _ = r0.verify_seal(blob, np.zeros(4, np.uint32))  # parses the blob; an empty seal is merely truncated
    assert reason == "seal truncated"


def test_a_real_segment_seal_is_accepted(orc, tmp_path):
    (seal_path,) = _need("segment_seal.bin")
    blob = _imported_blob(tmp_path)
    seal = np.fromfile(seal_path, dtype=np.uint32)
    ours, theirs = r0.verify_seal(blob, seal), orc.circuit(blob).verify(seal)
    assert ours[:2] == theirs, "the two verifiers disagree: %r vs %r" % (ours, theirs)
    assert ours[0] == 0, "risc0's seal is rejected at: %s -- that is where this transcript departs from risc0's" % ours[1]
