"""GPU checks at BASELINE.json's full size (po2 = 20, the 256-column bench circuit), where running the CPU oracle's
prover would take minutes: size-independent properties instead -- the oracle's *verifier* (cheap at any size) must accept
the full-size seal, transforms must round-trip, Merkle openings must recompute to the root."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry
import hyperfridge_r0_amd as r0
from conftest import circuit_path

pytestmark = pytest.mark.gpu
P = 2013265921


def test_full_size_seal_is_accepted_by_the_oracle_verifier_and_is_deterministic(hal, orc):
    blob = np.fromfile(circuit_path("bench"), dtype=np.uint32)
    co = entry.code_object_path("bench")
    gc = hal.load_circuit(blob, co if os.path.exists(co) else None)
    oc = orc.circuit(blob)
    po2 = 20
    code, data, glob = hal.witgen(gc, po2, seed=77)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    assert oc.verify(seal) == (0, "ok")
    assert r0.verify_seal(blob, seal) == (0, "ok", po2)  # the product's own host-side verifier
    assert np.array_equal(seal, hal.prove_segment(gc, po2, code, data, glob))
    bad = seal.copy()
    bad[seal.size // 2] ^= 1
    assert oc.verify(bad)[0] != 0 and r0.verify_seal(blob, bad)[:2] == oc.verify(bad)
    # a different segment of the same circuit
    code2, data2, glob2 = hal.witgen(gc, po2, seed=78)
    seal2 = hal.prove_segment(gc, po2, code2, data2, glob2)
    assert oc.verify(seal2) == (0, "ok") and not np.array_equal(seal2[:32], seal[:32])
    # per-op spot checks of the witness the proof was made from, against the oracle's column program
    ocode, odata, oglob = oc.witgen(po2, seed=77)
    assert np.array_equal(glob, oglob)
    for col in (0, 5, 100, 191):
        assert np.array_equal(data.to_host(col << po2, 1 << po2), odata[col << po2:(col + 1) << po2])


def test_ntt_round_trip_and_linearity_at_domain_size(hal, orc):
    rng = np.random.default_rng(3)
    po2, cols = 22, 2
    n = 1 << po2
    x = rng.integers(0, P, cols * n, dtype=np.uint32)
    buf = hal.copy_from(x)
    hal.batch_interpolate_ntt(buf, cols, po2)           # evals -> bit-reversed coeffs
    out = hal.alloc(cols * n)
    hal.batch_expand_into_evaluate_ntt(out, buf, cols, po2, 0)  # and back
    assert np.array_equal(out.to_host(), x)
    # linearity: NTT(a + b) = NTT(a) + NTT(b), one column
    a, b = x[:n], x[n:]
    s = hal.alloc(n)
    hal.eltwise_add_elem(s, hal.copy_from(a), hal.copy_from(b), n)
    hal.batch_interpolate_ntt(s, 1, po2)
    coeffs = buf.to_host()
    want = (coeffs[:n].astype(np.int64) + coeffs[n:]) % P
    assert np.array_equal(s.to_host().astype(np.int64), want)
    # a few coefficients against the oracle's direct definition on a column the CPU can afford
    small = orc.batch_interpolate_ntt(x[:1 << 16], 1, 16)
    sb = hal.copy_from(x[:1 << 16])
    hal.batch_interpolate_ntt(sb, 1, 16)
    assert np.array_equal(sb.to_host(), small)


def test_merkle_openings_recompute_to_the_root_at_full_size(hal, orc):
    rng = np.random.default_rng(4)
    rows, cols = 1 << 22, 16
    m = rng.integers(0, P, rows * cols, dtype=np.uint32)
    mat = hal.copy_from(m)
    nodes = hal.alloc(rows * 2 * 8)
    hal.merkle_build(nodes, mat, rows, cols)
    root = nodes.to_host(8, 8)
    for r in (0, 12345, rows - 1):
        cur = orc.hash_elem_slice(m[r::rows])
        idx = r + rows
        assert np.array_equal(nodes.to_host(idx * 8, 8), cur)
        while idx > 1:
            sib = nodes.to_host((idx ^ 1) * 8, 8)
            cur = orc.hash_pair(sib, cur) if idx & 1 else orc.hash_pair(cur, sib)
            idx //= 2
        assert np.array_equal(cur, root)
