"""`default_prover().prove(env, elf)` end to end on the device (r0h_prove_elf; host/src/main.rs:389-423, verifier/src/main.rs:118-128):
the reference's raw EBICS response is pre-processed natively, its thirteen ExecutorEnv frames are the guest's input, a guest ELF is
executed and segmented, every segment is proved on the GPU for its claim, the receipt is serialised, parsed back and verified
against the image id -- and every seal is checked by the CPU oracle's verifier, bound to the control root.

The guest is a STAND-IN, hand-assembled here (the reference ships no ELF; its guest needs the Rust toolchain): it reads the input
stream, folds it into a checksum and commits a serde-framed JSON string carrying it, the way hyperfridge's guest commits its
statement summary.  With a synthetic circuit the witness of a segment is that circuit's column program with the claim planted
(first test: the plumbing and the bindings between the stages); with the trace circuit the witness IS the segment's execution,
expanded on the device from the compact preflight rows (second test)."""
import json
import os
import struct

import numpy as np
import pytest

import __graft_entry__ as entry
import hyperfridge_r0_amd as r0
from conftest import ROOT, circuit_path
from test_rv32im import ADDI, A0, A1, A7, B, ECALL, I, J, LI, R, S, flat

pytestmark = pytest.mark.gpu
D = os.path.join(ROOT, "tests", "golden", "camt53")
rd = lambda name, mode="rb": open(os.path.join(D, name), mode).read()
T0, T1, T2, T3, T4, T5, S0, S1, S2 = 5, 6, 7, 28, 29, 30, 8, 9, 18


def stand_in_guest_elf():
    text, data, buf = 0x10000, 0x18000, 0x20000
    template = b'{"iban":"CH4308307000289537312","input_checksum":"00000000"}'
    off = template.index(b"00000000")
    frame = struct.pack("<I", len(template)) + template + bytes(-len(template) % 4)
    prog = flat(
        LI(S0, buf), ADDI(A0, S0, 0), ADDI(A1, 0, 1), ADDI(A7, 0, 1), ECALL,   # READ_WORDS(buf, 1): the word count
        I(0, S0, 2, S1, 0x03),                                                 # s1 = n
        ADDI(A0, S0, 4), ADDI(A1, S1, 0), ADDI(A7, 0, 1), ECALL,               # READ_WORDS(buf + 4, n)
        ADDI(T1, 0, 0), ADDI(T2, S0, 4), ADDI(T3, S1, 0), ADDI(T4, 0, 31),
        # loop: t1 = t1 * 31 + w
        B(28, 0, T3, 0), I(0, T2, 2, T0, 0x03), R(1, T4, T1, 0, T1), R(0, T0, T1, 0, T1), ADDI(T2, T2, 4), ADDI(T3, T3, -1), J(-24, 0),
        LI(S2, data + 4 + off), ADDI(T3, 0, 8),
        # hexloop: top nibble of t1 -> ASCII
        I(28, T1, 5, T0, 0x13), I(4, T1, 1, T1, 0x13), ADDI(T5, T0, -10), B(8, 0, T5, 4), ADDI(T0, T0, 39), ADDI(T0, T0, 48), S(0, T0, S2, 0), ADDI(S2, S2, 1),
        ADDI(T3, T3, -1), B(-36, 0, T3, 1),
        # the journal is a window of memory (R0H_JOURNAL_BASE): the frame moves there, then COMMIT names its words
        LI(T2, data), LI(T3, r0.JOURNAL_BASE), ADDI(T4, 0, len(frame) // 4),
        I(0, T2, 2, T0, 0x03), S(0, T0, T3, 2), ADDI(T2, T2, 4), ADDI(T3, T3, 4), ADDI(T4, T4, -1), B(-20, 0, T4, 1),
        LI(A0, r0.JOURNAL_BASE), ADDI(A1, 0, len(frame) // 4), ADDI(A7, 0, 2), ECALL,     # COMMIT(frame), in words
        ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)
    code = struct.pack("<%dI" % len(prog), *prog)
    ehdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, text, 52, 0, 0, 52, 32, 2, 0, 0, 0)
    o1 = 52 + 64
    o2 = o1 + len(code)
    ph = struct.pack("<IIIIIIII", 1, o1, text, text, len(code), len(code), 5, 4) + struct.pack("<IIIIIIII", 1, o2, data, data, len(frame), len(frame), 6, 4)
    return ehdr + ph + code + frame, template, off


def checksum(words):
    c = 0
    for w in words:
        c = (c * 31 + int(w)) & 0xFFFFFFFF
    return c


def test_response_to_verified_receipt(hal, orc, tmp_path):
    eb = r0.Ebics(rd("response.xml"))
    assert eb.check_digest() and eb.verify_bank_signature(rd("pub_bank.pem"))
    tx = rd("test.xml-TransactionKeyDecrypt.bin")
    assert eb.check_transaction_key(rd("pub_client.pem"), tx)[0]
    frames = eb.env_inputs(rd("pub_bank.pem"), "-----BEGIN PRIVATE KEY-----…", tx, "CH4308307000289537312", "host:main", rd("test.xml-Witness.hex", "r"),
                           rd("pub_witness.pem"), "verbose")
    stream = np.concatenate([[frames.size], frames]).astype(np.uint32)
    elf, template, off = stand_in_guest_elf()
    # the executor alone agrees with the Python model of the guest
    vm = r0.Vm()
    vm.load_elf(elf)
    vm.set_input(stream)
    assert vm.run(segment_po2=20) == (0, 0)
    want_json = template[:off] + b"%08x" % checksum(frames) + template[off + 8:]
    assert r0.journal_commitment(vm.journal) == want_json and r0.serde_decode_str(vm.journal)[0] == want_json
    total_cycles = vm.cycles

    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path("small"))
    receipt, image_id, cycles = hal.prove_elf(gc, elf, stream, segment_po2=11)
    assert cycles == total_cycles and image_id == vm.segments()[0].pre.digest()
    assert receipt.journal == vm.journal and len(receipt.seals()) >= 8
    oc = orc.circuit(blob)
    roots = {}
    for index, seal in receipt.seals():
        po2 = r0.verify_seal(blob, seal)[2]
        if po2 not in roots:
            roots[po2] = hal.code_root(gc, po2)
        assert oc.verify(seal, code_root=roots[po2]) == (0, "ok"), index
    assert max(roots) == 11
    text = receipt.to_json()
    (tmp_path / "receipt.json").write_text(text)
    back = r0.Receipt.parse(text)
    assert back.verify(blob, roots, image_id)[:2] == (0, "ok")
    assert json.loads(r0.journal_commitment(back.journal))["input_checksum"] == "%08x" % checksum(frames)
    # the things `receipt.verify(image_id)` exists to refuse
    assert back.verify(blob, roots, bytes(32))[0] == 8
    doc = json.loads(text)
    doc["journal"]["bytes"][10] ^= 1
    assert r0.Receipt.parse(json.dumps(doc)).verify(blob, roots, image_id)[0] == 7
    # another input, another journal, the same image id; a guest that fails is an error, not a receipt
    receipt2, image2, _ = hal.prove_elf(gc, elf, np.array([2, 5, 6], np.uint32), segment_po2=11)
    assert image2 == image_id and receipt2.journal != receipt.journal and len(receipt2.seals()) == 1
    bad_elf = bytearray(elf)
    bad_elf[52 + 64 + 8] ^= 0xFF  # corrupt an instruction word
    with pytest.raises(r0.R0HipError, match="guest trap|exited with code"):
        hal.prove_elf(gc, bytes(bad_elf), stream, segment_po2=11)
    with pytest.raises(r0.R0HipError, match="did not halt"):
        hal.prove_elf(gc, elf, stream, segment_po2=11, max_cycles=100)
    tiny = hal.load_circuit(np.fromfile(circuit_path("tiny"), dtype=np.uint32))
    with pytest.raises(r0.R0HipError, match="public inputs"):
        hal.prove_elf(tiny, elf, stream, segment_po2=11)
    gc.free()
    tiny.free()


def test_prove_elf_with_the_trace_circuit_proves_the_run_it_executed(hal, orc):
    """`prove(env, elf)` with circuits/trace.r0c: every seal of the receipt is a proof over THAT segment's cycles, and the receipt
    as a whole says that THIS ELF produced THIS journal.  The run is the stand-in guest over the input stream made from the
    reference's EBICS response, cut into 2^11-row segments (each proved at 2^16 rows: the lookup tables' size).  Checked here: the
    receipt verifies with the ELF (seals bound to the control roots, claims named by the seals, first / last pc of every seal equal
    to its claim's, states chaining, journal digest, the session's challenge and balance); every seal is accepted by the CPU
    oracle's verifier; the device's witness of every segment equals the host reference word for word, and the oracle proving from
    the host reference under the seal's own public inputs gives the receipt's seal word for word -- so what r0h_prove_elf proved is
    the execution an independent run of the executor records."""
    eb = r0.Ebics(rd("response.xml"))
    tx = rd("test.xml-TransactionKeyDecrypt.bin")
    frames = eb.env_inputs(rd("pub_bank.pem"), "-----BEGIN PRIVATE KEY-----…", tx, "CH4308307000289537312", "host:main", rd("test.xml-Witness.hex", "r"),
                           rd("pub_witness.pem"), "verbose")
    stream = np.concatenate([[frames.size], frames]).astype(np.uint32)
    elf, template, off = stand_in_guest_elf()
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    po2 = 11
    receipt, image_id, cycles = hal.prove_elf(gc, elf, stream, segment_po2=po2)
    stats = hal.last_session_stats()
    # the same run, executed again with the limits r0h_prove_elf uses
    vm = r0.Vm()
    vm.load_elf(elf)
    vm.set_input(stream)
    assert vm.run(segment_po2=po2, keep_trace=True, boundary_rows=True) == (0, 0)
    segs, claims = vm.segments(), vm.claims()
    assert cycles == vm.cycles and image_id == segs[0].pre.digest() == r0.compute_image_id(elf) and receipt.journal == vm.journal
    seals = receipt.seals()
    assert len(seals) == len(segs) >= 6 and stats["segments"] == len(segs) and stats["cycles"] == cycles
    want_json = template[:off] + b"%08x" % checksum(frames) + template[off + 8:]
    assert r0.journal_commitment(receipt.journal) == want_json
    assert stats["lean_segments"] == 0
    # the same session with (almost) nothing kept between the phases: all but the first segment give their evaluations back after the
    # commitment and are evaluated again when the challenge is known -- the seals are the same words
    hal.set_session_resident_limit(r0.TRACE_COLUMNS * (16 << r0.TRACE_MIN_PO2))
    lean, _, _ = hal.prove_elf(gc, elf, stream, segment_po2=po2)
    assert hal.last_session_stats()["lean_segments"] == len(segs) - 1
    hal.set_session_resident_limit(0)
    assert all(ia == ib and np.array_equal(a, b) for (ia, a), (ib, b) in zip(lean.seals(), seals)) and lean.journal == receipt.journal
    # with the image circuit set on the context the receipt carries an image proof: the image id alone verifies it (no ELF), the proof
    # is the oracle's word for word, and it is about THIS session (another session's challenge is refused)
    iblob = np.fromfile(circuit_path("image"), dtype=np.uint32)
    ic = hal.load_circuit(iblob, entry.code_object_path("image"))
    hal.set_image_circuit(ic)
    with_proof, _, _ = hal.prove_elf(gc, elf, stream, segment_po2=po2)
    hal.set_image_circuit(None)
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(with_proof.seals(), seals)) and with_proof.image_proof is not None and receipt.image_proof is None
    oi = orc.circuit(iblob)
    ipo2 = r0.image_po2(elf)
    idata, iglob = r0.image_witness(elf, ipo2)
    iglob[r0.IMAGE_GAMMA:r0.IMAGE_GAMMA + 16] = seals[0][1][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]
    icode = oi.witgen(ipo2, 0)[0]
    want_image_seal = oi.prove(ipo2, icode, idata.reshape(-1), oi.logup_totals(ipo2, icode, idata.reshape(-1), iglob))
    assert np.array_equal(with_proof.image_proof, want_image_seal) and np.array_equal(hal.prove_image(ic, elf, iglob[r0.IMAGE_GAMMA:r0.IMAGE_GAMMA + 16]), want_image_seal)
    ic.free()
    size = r0.TRACE_MIN_PO2
    cc = hal.code_commit(gc, size)
    roots, ocode = {size: cc.root()}, c.witgen(size, 0)[0]
    assert with_proof.verify_image(blob, roots, iblob, image_id)[:2] == (0, "ok") and receipt.verify_image(blob, roots, iblob, image_id)[0] == 16
    assert with_proof.verify_image(blob, roots, iblob, bytes(32))[0] == 8
    with_proof.image_proof = None  # (no proof: back to needing the ELF)
    assert with_proof.verify_image(blob, roots, iblob, image_id)[0] == 16 and with_proof.verify(blob, roots, None, elf=elf)[:2] == (0, "ok")
    assert np.array_equal(r0.control_root_host(blob, size), roots[size])
    halting = max(k for k, s in enumerate(segs) if s.user_cycles)
    for k, (index, seal) in enumerate(seals):
        s = segs[k]
        assert index == k and r0.verify_seal(blob, seal)[2] == size and s.user_cycles + s.boundary_rows <= 1 << po2
        assert c.verify(seal, code_root=roots[size]) == (0, "ok"), k
        publics = [orc.dec(int(g)) for g in seal[8:18]]
        ends = [1, 1, 0, 0] if k == halting else [0, 0, 0, 0]  # HALT(0) ends the run, the segments before it are cut, what follows only closes the session
        assert publics == [s.pre.pc, s.post.pc, s.user_cycles] + ends + [k + 1, s.closing, 0 if s.closing else k + 1] and np.array_equal(seal[:8], claims[k].globals())
        rows, bounds = vm.preflight_arrays(k)
        data, glob = vm.trace_witness(k, size, claim_globals=claims[k].globals())
        dev, dglob = hal.trace_witgen(rows, bounds, size, claim_globals=claims[k].globals(), number=k + 1, closing=bool(s.closing), idle_pc=s.pre.pc, circuit=gc)
        assert np.array_equal(dev.to_host(), data) and np.array_equal(dglob, glob) and np.array_equal(glob[:20], seal[:20])
        dev.free()
        if k in (0, len(seals) // 2, len(seals) - 1):
            assert np.array_equal(seal, c.prove(size, ocode, data, seal[:r0.TRACE_GLOBALS])), k
    assert segs[-1].closing and all(segs[k].post.pc == segs[k + 1].pre.pc for k in range(len(segs) - 1)) and sum(s.user_cycles for s in segs) == cycles
    back = r0.Receipt.parse(receipt.to_json())
    assert back.verify(blob, roots, None, elf=elf)[:2] == (0, "ok")
    assert back.verify(blob, roots, image_id)[0] == 15 and back.verify(blob, roots, None)[0] == 12 and back.verify(blob, roots, bytes(32))[0] == 8
    other_elf = bytearray(elf)
    other_elf[-2] ^= 1  # a byte of the image the run never looks at: still another program
    assert back.verify(blob, roots, None, elf=bytes(other_elf))[0] in (8, 14)
    doc = json.loads(receipt.to_json())
    doc["journal"]["bytes"][10] ^= 1
    assert r0.Receipt.parse(json.dumps(doc)).verify(blob, roots, None, elf=elf)[0] == 7
    # a seal proved for a claim whose pc is not the one the run starts from: the seal is valid and names that claim, the states
    # chain -- and it is refused because public input 8 (the pc the circuit pins the first cycle to) is not the claim's
    gamma = seals[0][1][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]
    code_cols, synthetic, _ = hal.witgen(gc, size, 0)
    synthetic.free()

    def reprove(k, claim):
        rows, bounds = vm.preflight_arrays(k)
        dev, glob = hal.trace_witgen(rows, bounds, size, claim_globals=claim.globals(), number=k + 1, closing=bool(segs[k].closing), idle_pc=segs[k].pre.pc, circuit=gc)
        glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = gamma
        seal = hal.prove_segment(gc, size, cc, dev, hal.logup_totals(gc, size, code_cols, dev, glob))
        dev.free()
        return seal

    k = 0
    forged = r0.ReceiptClaim.make(r0.SystemState.make(segs[k].pre.pc + 4, bytes(segs[k].pre.merkle_root)), segs[k].post, claims[k].exit_system, claims[k].exit_user, None)
    seal = reprove(k, forged)
    assert c.verify(seal, code_root=roots[size]) == (0, "ok")
    rc2 = r0.Receipt.new(receipt.journal, [seal] + [s for _, s in seals[1:]], [forged] + claims[1:])
    assert rc2.verify(blob, roots, None, elf=elf)[:3] == (5, "a seal's public inputs do not name its claim", 0)
    # the same for the way the run ends: a claim that says Halted(7) over a run whose last cycle is HALT with a0 = 0
    k = halting
    forged = r0.ReceiptClaim.make(segs[k].pre, segs[k].post, 0, 7, claims[k].output_digest)
    seal = reprove(k, forged)
    assert c.verify(seal, code_root=roots[size]) == (0, "ok") and [orc.dec(int(g)) for g in seal[11:15]] == [1, 1, 0, 0]
    rc3 = r0.Receipt.new(receipt.journal, [s for _, s in seals[:k]] + [seal] + [s for _, s in seals[k + 1:]], claims[:k] + [forged] + claims[k + 1:])
    assert rc3.verify(blob, roots, None, elf=elf)[:3] == (5, "a seal's public inputs do not name its claim", k)
    # a seal that is fine by itself but was made under another challenge than the session's
    other_gamma, gamma = gamma, seals[0][1][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16][::-1].copy()
    seal = reprove(1, claims[1])
    gamma = other_gamma
    rc4 = r0.Receipt.new(receipt.journal, [seals[0][1], seal] + [s for _, s in seals[2:]], claims)
    assert c.verify(seal, code_root=roots[size]) == (0, "ok") and rc4.verify(blob, roots, None, elf=elf)[0] == 13
    code_cols.free()
    cc.free()
    # the number of prover lanes (contexts of the device that take segments as the executor cuts them) changes who proves what,
    # not what is proved: one lane and three lanes give the receipt of the default two, seal for seal
    for lanes in ("1", "3"):
        os.environ["R0H_SESSION_LANES"] = lanes
        try:
            again, image_again, _ = hal.prove_elf(gc, elf, stream, segment_po2=po2)
        finally:
            del os.environ["R0H_SESSION_LANES"]
        assert image_again == image_id and again.to_json() == receipt.to_json(), lanes
    # the page-locked row buffers stay with the context between calls: a run with another segment size in between (buffers of the wrong
    # capacity are unpinned and dropped, new ones pinned) changes nothing about the next run of this size
    other, _, _ = hal.prove_elf(gc, elf, stream, segment_po2=po2 + 1)
    assert len(other.seals()) < len(seals) and other.verify(blob, roots, None, elf=elf)[:2] == (0, "ok")
    again, _, _ = hal.prove_elf(gc, elf, stream, segment_po2=po2)
    assert again.to_json() == receipt.to_json()
    # a session proved in shares (what each GPU of a multi-GPU run does: segments part, part + parts, ...): every share commits its own
    # segments, the shares exchange their records (28 words per segment: the one exchange of the path), every share finishes under the
    # common challenge -- and merged, in any order, they are the receipt above, seal for seal
    for parts in (2, 3):
        sessions = [hal.session_begin(gc, elf, stream, segment_po2=po2, part=k, parts=parts) for k in range(parts)]
        records = np.zeros((len(seals), r0.SESSION_RECORD_WORDS), dtype=np.uint32)
        for ses in sessions:
            idx, rec = ses.records()
            records[idx] = rec
        with pytest.raises(r0.R0HipError, match="records"):
            sessions[0].finish(records[:-1])
        shares = [ses.finish(records)[0] for ses in sessions]
        assert [[i for i, _ in sh.seals()] for sh in shares] == [list(range(k, len(seals), parts)) for k in range(parts)]
        assert shares[1].verify(blob, roots, None, elf=elf)[0] != 0  # a share alone is not the session
        assert r0.Receipt.merge(shares[::-1]).to_json() == receipt.to_json(), parts
    with pytest.raises(r0.R0HipError, match="share one challenge"):
        hal.prove_elf(gc, elf, stream, segment_po2=po2, part=0, parts=2)
    # a guest that fails or never halts is an error, not a receipt
    with pytest.raises(r0.R0HipError, match="did not halt"):
        hal.prove_elf(gc, elf, stream, segment_po2=po2, max_cycles=100)
    bad_elf = bytearray(elf)
    bad_elf[52 + 64 + 8] ^= 0xFF
    with pytest.raises(r0.R0HipError, match="guest trap|exited with code"):
        hal.prove_elf(gc, bytes(bad_elf), stream, segment_po2=po2)
    gc.free()


@pytest.mark.gpu
def test_a_real_session_sharded_over_two_ranks_under_the_drivers_launcher():
    """`python -m torch.distributed.run --nproc-per-node 2 ... tools/bench_session.py --guest camt53 --backend gloo --share-device`: both
    ranks execute the camt53 guest and prove every other segment on the one GPU (r0h_prove_elf_part), rank 1's receipt travels to
    rank 0 as JSON over send / recv, rank 0 merges (r0h_receipt_merge) and verifies the receipt against the image id before it prints
    its line.  (Two GPU processes and this one: within the box's limit of six.)"""
    import subprocess
    import sys
    from conftest import ROOT
    port = 29700 + os.getpid() % 200
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tools", "bench_session.py"), "--guest", "camt53", "--backend", "gloo", "--share-device", "--repeat", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["segments"] == 12 and line["receipt_verified"] is True and line["cycles"] > 11_000_000
    assert line["journal_commitment"].startswith('{"hostinfo":"host:main","iban":"CH4308307000289537312"')
