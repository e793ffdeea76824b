"""The CODE group committed once per (circuit, po2) -- r0h_code_commit_new / r0h_prove_segment_committed (include/r0hip.h).
The seal of a proof that reads the kept commitment is word for word the seal of r0h_prove_segment (and hence the CPU oracle's);
the commitment's root is the control root; a second context of the same device may read it; mismatched sizes are refused."""
import threading

import numpy as np
import pytest

import __graft_entry__ as entry
import hyperfridge_r0_amd as r0
from conftest import circuit_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,po2", [("small", 10), ("small", 13), ("bench", 16)])
def test_committed_code_gives_the_same_seal_as_the_oracle(hal, orc, name, po2):
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path(name))
    oc = orc.circuit(blob)
    cc = hal.code_commit(gc, po2)
    assert np.array_equal(cc.root(), hal.code_root(gc, po2))
    assert np.array_equal(cc.root(), r0.control_root_host(blob, po2))  # what a verifier derives from the blob alone, on the host
    for seed in (7, 8):
        code, data, glob_ = hal.witgen(gc, po2, seed)
        plain = hal.prove_segment(gc, po2, code, data, glob_)
        kept = hal.prove_segment(gc, po2, cc, data, glob_)
        assert np.array_equal(plain, kept)
        ocode, odata, oglob = oc.witgen(po2, seed=seed)
        assert np.array_equal(kept, oc.prove(po2, ocode, odata, oglob))
        assert oc.verify(kept, code_root=cc.root()) == (0, "ok")
        # the split sequencer takes the commitment as well
        proof, mix = hal.proof_begin(gc, po2, cc, data, glob_)
        accum = hal.accum(gc, po2, code, data, mix)
        assert np.array_equal(hal.proof_finish(proof, accum), kept)
        for b in (code, data, accum):
            b.free()
    cc.free()
    gc.free()


def test_one_commitment_serves_several_contexts_of_the_device(hal, orc):
    """bench.py's arrangement: lane 0 commits, every lane (own context, own stream, own host thread) proves against it."""
    name, po2 = "small", 12
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path(name))
    cc = hal.code_commit(gc, po2)
    oc = orc.circuit(blob)
    lanes = []
    for k in range(3):
        h = r0.Hal(0)
        c = h.load_circuit(blob, entry.code_object_path(name))
        code, data, glob_ = h.witgen(c, po2, 100 + k)
        lanes.append(dict(hal=h, circuit=c, code=code, data=data, glob=glob_, seals=[]))

    def run(lane):
        for _ in range(3):
            lane["seals"].append(lane["hal"].prove_segment(lane["circuit"], po2, cc, lane["data"], lane["glob"]))

    ts = [threading.Thread(target=run, args=(ln,)) for ln in lanes]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k, ln in enumerate(lanes):
        ocode, odata, oglob = oc.witgen(po2, seed=100 + k)
        want = oc.prove(po2, ocode, odata, oglob)
        assert len(ln["seals"]) == 3 and all(np.array_equal(s, want) for s in ln["seals"])
        for key in ("code", "data", "circuit"):
            ln[key].free()
        ln["hal"].close()
    cc.free()
    gc.free()


def test_a_commitment_of_another_size_is_refused(hal):
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path("small"))
    cc = hal.code_commit(gc, 10)
    code, data, glob_ = hal.witgen(gc, 11, 1)
    with pytest.raises(r0.R0HipError, match="CODE commitment"):
        hal.prove_segment(gc, 11, cc, data, glob_)
    tiny = np.fromfile(circuit_path("tiny"), dtype=np.uint32)
    gt = hal.load_circuit(tiny, None)
    tcode, tdata, tglob = hal.witgen(gt, 10, 1)
    with pytest.raises(r0.R0HipError, match="CODE commitment"):
        hal.prove_segment(gt, 10, cc, tdata, tglob)
    for b in (code, data, tcode, tdata):
        b.free()
    cc.free(); gt.free(); gc.free()
