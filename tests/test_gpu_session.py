"""`default_prover().prove(env, elf)` end to end on the device (r0h_prove_elf; host/src/main.rs:389-423, verifier/src/main.rs:118-128):
the reference's raw EBICS response is pre-processed natively, its thirteen ExecutorEnv frames are the guest's input, a guest ELF is
executed and segmented, every segment is proved on the GPU for its claim, the receipt is serialised, parsed back and verified
against the image id -- and every seal is checked by the CPU oracle's verifier, bound to the control root.

The guest is a STAND-IN, hand-assembled here (the reference ships no ELF; its guest needs the Rust toolchain): it reads the input
stream, folds it into a checksum and commits a serde-framed JSON string carrying it, the way hyperfridge's guest commits its
statement summary.  The witness of a segment is the circuit's synthetic column program with the claim planted (csrc/session.hip),
so what this test pins is the plumbing and the bindings between the stages, not the rv32im circuit."""
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
        LI(A0, data), ADDI(A1, 0, len(frame)), ADDI(A7, 0, 2), ECALL,          # COMMIT(frame)
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


def test_every_segment_of_the_guest_run_proves_as_a_trace(hal, orc):
    """The same run -- the stand-in guest over the input stream made from the reference's EBICS response -- cut into 2^11-cycle
    segments with the preflight trace kept: each segment's rows go through the trace circuit (csrc/rv32im.hip
    r0h_vm_trace_witness, circuits/trace.r0c) on the device.  The seals chain like the run does: a segment's public first pc is
    its predecessor's public last pc, the cycle counts add up, and the CPU oracle's verifier accepts every one."""
    eb = r0.Ebics(rd("response.xml"))
    tx = rd("test.xml-TransactionKeyDecrypt.bin")
    frames = eb.env_inputs(rd("pub_bank.pem"), "-----BEGIN PRIVATE KEY-----…", tx, "CH4308307000289537312", "host:main", rd("test.xml-Witness.hex", "r"),
                           rd("pub_witness.pem"), "verbose")
    stream = np.concatenate([[frames.size], frames]).astype(np.uint32)
    elf, _, _ = stand_in_guest_elf()
    vm = r0.Vm()
    vm.load_elf(elf)
    vm.set_input(stream)
    assert vm.run(segment_po2=11, keep_trace=True) == (0, 0)
    segs = vm.segments()
    assert len(segs) >= 4
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    po2 = 11
    code, synthetic, _ = hal.witgen(gc, po2, 0)
    synthetic.free()
    root = hal.code_root(gc, po2, code)
    dev = hal.alloc(r0.TRACE_COLUMNS << po2)
    publics = []
    for k, s in enumerate(segs):
        data, glob = vm.trace_witness(k, po2)
        dev.upload(data)
        seal = hal.prove_segment(gc, po2, code, dev, glob)
        assert c.verify(seal, code_root=root) == (0, "ok"), k
        publics.append([orc.dec(int(g)) for g in glob])
        assert publics[-1] == [s.pre.pc, s.post.pc, s.user_cycles]
    assert all(publics[k][1] == publics[k + 1][0] for k in range(len(segs) - 1)) and sum(p[2] for p in publics) == vm.cycles
    code.free(); dev.free(); gc.free()
