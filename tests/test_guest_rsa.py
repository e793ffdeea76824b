"""The guest-shaped program (tools/guest_rsa.py, hand-assembled RV32IM): SHA-256 and three RSA-2048 public-key operations on the
reference's own EBICS fixture -- what hyperfridge's guest spends 83 % of its cycles on (methods/guest/src/main.rs:450-485, 513-517,
663-718, 757-833; docs/hyperfridge-cycles.html).  Pinned by reference-held data: `SHA-256(test.xml-SignedInfo) = 6ce63c3d..07f5bd` is
the tail of `sig^65537 mod n_bank` (re-derived there from the fixtures; methods/guest/src/test_xmlparse.rs:70-86
is the reference's own test of the same fact), the transaction-key block raised to e is the <TransactionKey> ciphertext, the witness
signature verifies over SHA-256(decoded order data).  Python's hashlib / pow are the independent model."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import ROOT, circuit_path

sys.path.insert(0, os.path.join(ROOT, "tools"))
import guest_rsa  # noqa: E402

KAT = "6ce63c3daa3a181bf8ffe4e5187c4427d8a7acc7dcfeb8cbdb7f28211907f5bd"


@pytest.fixture(scope="module")
def guest():
    image, labels, layout = guest_rsa.build()
    return dict(elf=image, inputs=guest_rsa.reference_inputs())


def run(guest, po2=20, **changes):
    inp = dict(guest["inputs"], **changes)
    e = inp.pop("e", 65537)
    vm = r0.Vm()
    vm.load_elf(guest["elf"])
    vm.set_input(guest_rsa.input_stream(e=e, **inp))
    return vm, vm.run(segment_po2=po2)


def test_the_fixture_facts_the_guest_checks_hold_in_python(guest):
    i = guest["inputs"]
    assert hashlib.sha256(i["signed_info"]).hexdigest() == KAT
    em = pow(int.from_bytes(i["bank_sig"], "big"), 65537, i["bank_n"]).to_bytes(256, "big")
    assert em[:2] == b"\x00\x01" and em[-32:].hex() == KAT and em[2:204] == b"\xff" * 202 and em[204:224] == b"\x00" + guest_rsa.DER_SHA256
    assert pow(int.from_bytes(i["tx_plain"], "big"), 65537, i["client_n"]) == int.from_bytes(i["tx_cipher"], "big")
    wem = pow(int.from_bytes(i["witness_sig"], "big"), 65537, i["witness_n"]).to_bytes(256, "big")
    assert wem[-32:] == hashlib.sha256(i["order_data"]).digest() and len(i["order_data"]) == 2864


def test_the_guest_verifies_the_reference_fixture_and_commits_the_known_digest(guest):
    vm, (kind, code) = run(guest)
    assert (kind, code) == (0, 0)
    text, _ = r0.serde_decode_str(vm.journal)
    doc = json.loads(text)
    assert doc == {"signed_info_sha256": KAT, "order_data_sha256": hashlib.sha256(guest["inputs"]["order_data"]).hexdigest(), "bank_signature": "ok",
                   "transaction_key": "ok", "witness_signature": "ok"}
    assert r0.journal_commitment(vm.journal) == text and len(vm.journal) % 4 == 0
    assert 9_000_000 < vm.cycles < 11_000_000 and len(vm.segments()) >= 9  # three RSA operations of ~3.1 M cycles, 57 SHA-256 blocks
    # the image id is the program's, whatever the input
    other, _ = run(guest, signed_info=b"x")
    assert other.segments()[0].pre.digest() == vm.segments()[0].pre.digest()


def test_a_forged_input_makes_the_guest_exit_non_zero(guest):
    i = guest["inputs"]
    flip = lambda b, k: bytes(b[:k]) + bytes([b[k] ^ 1]) + bytes(b[k + 1:])
    cases = [(dict(bank_sig=flip(i["bank_sig"], 100)), 1),            # one signature bit
             (dict(signed_info=flip(i["signed_info"], 10)), 1),       # one message bit
             (dict(bank_n=i["bank_n"] ^ (1 << 900)), 1),              # another modulus (still odd, top bit set)
             (dict(tx_plain=flip(i["tx_plain"], 255)), 2),            # another transaction key
             (dict(witness_sig=flip(i["witness_sig"], 0)), 3),
             (dict(order_data=flip(i["order_data"], 2000)), 3),
             (dict(e=3), 4)]
    for changes, want in cases:
        vm, (kind, code) = run(guest, **changes)
        assert (kind, code) == (0, want), (list(changes), code)
        assert vm.journal == b""  # nothing is committed by a run that fails


def test_sha256_of_the_guest_on_every_padding_length(guest):
    """the hash routine alone, through the first step of the guest: messages of 0..130 bytes cover every padding case; the run exits 1
    (no valid signature for them), having left the digest in memory"""
    _, labels, layout = guest_rsa.build()
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [119, 120, 127, 128, 130]:
        msg = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        vm, (kind, code) = run(guest, signed_info=msg)
        assert (kind, code) == (0, 1)
        got = b"".join(int(w).to_bytes(4, "big") for w in vm.read(layout["DIGEST"], 8))
        assert got == hashlib.sha256(msg).digest(), n


@pytest.mark.gpu
def test_the_guest_run_is_proved_segment_by_segment_with_the_trace_circuit(hal, orc):
    """`prove(env, elf)` over the guest-shaped program on the reference's inputs: ten 2^20-row segments, each expanded on the device
    from its compact preflight rows and proved with circuits/trace.r0c; the receipt verifies with the ELF (the session sum over its
    image words and the journal included) and carries the known digest; the CPU oracle's verifier accepts the first and the last seal
    bound to their control roots."""
    import __graft_entry__ as entry
    image, stream, _ = guest_rsa.elf_and_input()
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    receipt, image_id, cycles = hal.prove_elf(gc, image, stream, segment_po2=20)
    stats = hal.last_session_stats()
    seals = receipt.seals()
    assert len(seals) >= 9 and stats["segments"] == len(seals) and cycles > 9_000_000
    assert json.loads(r0.journal_commitment(receipt.journal))["signed_info_sha256"] == KAT
    roots = {}
    for _, seal in seals:
        size = r0.verify_seal(blob, seal)[2]
        if size not in roots:
            cc = hal.code_commit(gc, size)
            roots[size] = cc.root()
            cc.free()
    assert receipt.verify(blob, roots, None, elf=image)[:2] == (0, "ok") and receipt.verify(blob, roots, image_id)[0] == 15
    oc = orc.circuit(blob)
    for _, seal in (seals[0], seals[-1]):
        assert oc.verify(seal, code_root=roots[r0.verify_seal(blob, seal)[2]]) == (0, "ok")
    # a forged signature: the guest exits 1 and `prove` is an error, not a receipt
    bad = guest_rsa.reference_inputs()
    bad["bank_sig"] = bytes([bad["bank_sig"][0] ^ 1]) + bad["bank_sig"][1:]
    with pytest.raises(r0.R0HipError, match="exited with code 1"):
        hal.prove_elf(gc, image, guest_rsa.input_stream(**bad), segment_po2=20)
    gc.free()
