"""tools/import_risc0_circuit.py: risc0-style generated Rust tables (taps.rs / poly_ext.rs text) -> circuit blob.
No real risc0 file is available offline, so the round trip is exercised on text emitted from our own
blobs in the same syntax."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, circuit_path

TOOL = os.path.join(ROOT, "tools", "import_risc0_circuit.py")


def sections(words):
    pos, out = 3, {}
    for _ in range(int(words[2])):
        out[int(words[pos])] = words[pos + 2:pos + 2 + int(words[pos + 1])]
        pos += 2 + int(words[pos + 1])
    return out


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_rust_text_round_trip_preserves_taps_and_program(tmp_path, name, orc):
    import hyperfridge_r0_amd as r0
    subprocess.check_call([sys.executable, TOOL, "--emit-rust", circuit_path(name), str(tmp_path)])
    text = open(tmp_path / "poly_ext.rs").read()
    assert "PolyExtStep::AndEqz(" in text and "PolyExtStep::Get(" in text and "TapData {" in open(tmp_path / "taps.rs").read()
    out = tmp_path / "imported.r0c"
    subprocess.check_call([sys.executable, TOOL, str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(out), "--info", "R0HIP_SYNTH:v1__"])
    a = sections(np.fromfile(circuit_path(name), dtype=np.uint32))
    b = sections(np.fromfile(out, dtype=np.uint32))
    for sec in (1, 2, 4, 7):  # GROUPS, TAPS, POLY, INFO survive verbatim
        assert np.array_equal(a[sec], b[sec]), sec
    assert list(b[3][:2]) == list(a[3][:2])  # n_global, n_mix recovered from the GetGlobal steps
    assert 5 not in b and 6 not in b        # no synthetic column program
    # the product accepts the imported blob (host-only path: parse + plan + codegen) and generates the same kernels
    blob_a, blob_b = np.fromfile(circuit_path(name), dtype=np.uint32), np.fromfile(out, dtype=np.uint32)
    assert r0.emit_eval_check_source(blob_b) == r0.emit_eval_check_source(blob_a)
    # and the oracle evaluates the same constraint polynomial from it
    ca, cb = orc.circuit(blob_a), orc.circuit(blob_b)
    rng = np.random.default_rng(1)
    P = 2013265921
    u = rng.integers(0, P, 4 * ca.n_taps, dtype=np.uint32)
    g, m, pm = rng.integers(0, P, max(ca.n_global, 1), dtype=np.uint32), rng.integers(0, P, max(ca.n_mix, 1), dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)

    def poly_ext(c):
        import ctypes
        tot = np.zeros(4, np.uint32)
        orc.L.orc_poly_ext(c.h, pm.ctypes.data_as(ctypes.c_void_p), u.ctypes.data_as(ctypes.c_void_p), g.ctypes.data_as(ctypes.c_void_p),
                           m.ctypes.data_as(ctypes.c_void_p), tot.ctypes.data_as(ctypes.c_void_p))
        return tot

    assert np.array_equal(poly_ext(ca), poly_ext(cb))


def test_importer_rejects_malformed_tables(tmp_path):
    (tmp_path / "taps.rs").write_text("TapData { offset: 1, back: 0, group: 0, combo: 0, skip: 1 }, TapData { offset: 0, back: 0, group: 0, combo: 0, skip: 1 },")
    (tmp_path / "poly_ext.rs").write_text("block: &[PolyExtStep::True], ret: 0,")
    r = subprocess.run([sys.executable, TOOL, str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(tmp_path / "o.r0c")], capture_output=True, text=True)
    assert r.returncode != 0 and "sorted" in r.stderr
    (tmp_path / "taps.rs").write_text("TapData { offset: 0, back: 0, group: 0, combo: 0, skip: 1 },")
    (tmp_path / "poly_ext.rs").write_text("block: &[PolyExtStep::Frobnicate(1)], ret: 0,")
    r = subprocess.run([sys.executable, TOOL, str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(tmp_path / "o.r0c")], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown PolyExtStep" in r.stderr


@pytest.mark.gpu
def test_imported_circuit_runs_eval_check_on_the_device_and_refuses_synthetic_steps(tmp_path, hal, orc):
    import hyperfridge_r0_amd as r0
    subprocess.check_call([sys.executable, TOOL, "--emit-rust", circuit_path("small"), str(tmp_path)])
    out = tmp_path / "imported.r0c"
    subprocess.check_call([sys.executable, TOOL, str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(out), "--info", "R0HIP_SYNTH:v1__"])
    blob = np.fromfile(out, dtype=np.uint32)
    gc, oc = hal.load_circuit(blob), orc.circuit(blob)
    po2, P = 10, 2013265921
    rng = np.random.default_rng(2)
    dom = 4 << po2
    ea, ec, ed = (rng.integers(0, P, oc.group_size[g] * dom, dtype=np.uint32) for g in range(3))
    glob, mix, pm = rng.integers(0, P, oc.n_global, dtype=np.uint32), rng.integers(0, P, oc.n_mix, dtype=np.uint32), rng.integers(0, P, 4, dtype=np.uint32)
    check = hal.eval_check(gc, po2, hal.copy_from(ea), hal.copy_from(ec), hal.copy_from(ed), glob, mix, pm)
    assert np.array_equal(check.to_host(), oc.eval_check(po2, ea, ec, ed, glob, mix, pm))
    with pytest.raises(r0.R0HipError):
        hal.witgen(gc, po2, 1)
    code, data = hal.alloc(oc.group_size[1] << po2), hal.alloc(oc.group_size[2] << po2)
    with pytest.raises(r0.R0HipError):
        hal.prove_segment(gc, po2, code, data, glob)


@pytest.mark.gpu
def test_split_sequencer_with_caller_supplied_accumulation(tmp_path, hal, orc):
    """r0h_proof_begin / r0h_proof_finish: the caller owns the accumulation step (as risc0's segment driver does).  With the
    caller computing the same accumulators the seal equals r0h_prove_segment's; and an imported circuit (no column
    program in the blob) proves through this path with witness + accumulators supplied from outside."""
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    gc, oc = hal.load_circuit(blob), orc.circuit(blob)
    po2 = 11
    code, data, glob = hal.witgen(gc, po2, 31)
    whole = hal.prove_segment(gc, po2, code, data, glob)
    proof, mix = hal.proof_begin(gc, po2, code, data, glob)
    accum_host = oc.accum(po2, code.to_host(), data.to_host(), mix)   # "caller's own step_accum": here the oracle's
    seal = hal.proof_finish(proof, hal.copy_from(accum_host))
    assert np.array_equal(seal, whole) and oc.verify(seal) == (0, "ok")
    # abandoned proofs release their buffers
    proof2, _ = hal.proof_begin(gc, po2, code, data, glob)
    hal.proof_abort(proof2)
    # imported circuit: same taps / program / info tag, but the blob carries no WITGEN/ACCUM
    subprocess.check_call([sys.executable, TOOL, "--emit-rust", circuit_path("small"), str(tmp_path)])
    out = tmp_path / "imported.r0c"
    subprocess.check_call([sys.executable, TOOL, str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(out), "--info", "R0HIP_SYNTH:v1__"])
    ic = hal.load_circuit(np.fromfile(out, dtype=np.uint32))
    proof3, mix3 = hal.proof_begin(ic, po2, code, data, glob)
    assert np.array_equal(mix3, mix)  # same transcript so far
    seal3 = hal.proof_finish(proof3, hal.copy_from(accum_host))
    assert np.array_equal(seal3, whole)
