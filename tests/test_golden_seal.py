"""Frozen seal (tests/golden/seal_tiny_po2_9_seed_1.npy, written by tools/gen_golden_seals.py): the oracle and the device
must both reproduce it word for word."""
import os

import numpy as np
import pytest

from conftest import ROOT, circuit_path

GOLDEN = os.path.join(ROOT, "tests", "golden", "seal_tiny_po2_9_seed_1.npy")


def test_oracle_reproduces_the_frozen_seal(orc):
    want = np.load(GOLDEN)
    c = orc.circuit(np.fromfile(circuit_path("tiny"), dtype=np.uint32))
    code, data, glob = c.witgen(9, 1)
    assert np.array_equal(c.prove(9, code, data, glob), want)
    assert c.verify(want) == (0, "ok")


@pytest.mark.gpu
def test_device_reproduces_the_frozen_seal(hal):
    want = np.load(GOLDEN)
    gc = hal.load_circuit(np.fromfile(circuit_path("tiny"), dtype=np.uint32))
    code, data, glob = hal.witgen(gc, 9, 1)
    assert np.array_equal(hal.prove_segment(gc, 9, code, data, glob), want)
