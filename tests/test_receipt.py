"""Data formats either side of the proving path (SURVEY.md 8(a) a0', a0'', a18), pinned by the reference's own receipt
fixtures -- data/test/test.xml-Receipt-test.json and test.xml-Receipt-6bb958...-latest.json, committed unchanged as
tests/golden/reference_receipt_*.json: the Receipt JSON envelope, the serde framing of journal.bytes and the commitment
hyperfridge reads out of it (host/src/main.rs:251-267, verifier/src/main.rs:118-119,176-185).  Host code only."""
import json
import os
import subprocess

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import ROOT, circuit_path

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["reference_receipt_test.json", "reference_receipt_6bb95807_latest.json"]
VERIFY = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_verify")


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_receipts_parse_and_reserialise_byte_for_byte(name):
    text = open(os.path.join(GOLDEN, name)).read()
    rc = r0.Receipt.parse(text)
    assert rc.kind == "Fake" and rc.seals() == []
    assert rc.journal == bytes(json.loads(text)["journal"]["bytes"])
    assert rc.to_json() == text  # serde_json's compact form, same key order


@pytest.mark.parametrize("name,length", [(FIXTURES[0], 329), (FIXTURES[1], 1768)])
def test_journal_framing_matches_the_reference_fixtures(name, length):
    """journal = [u32 LE len][utf8 JSON][zero pad to 4] (fixture 2 starts 232,6,0,0 = 1768): decode, then re-encode to the same bytes."""
    journal = r0.Receipt.parse(open(os.path.join(GOLDEN, name)).read()).journal
    text, used = r0.serde_decode_str(journal)
    assert len(text) == length and used == len(journal) == 4 + (length + 3) // 4 * 4
    assert r0.serde_encode_str(text) == journal
    commitment = json.loads(r0.journal_commitment(journal))
    assert r0.journal_commitment(journal) == text  # the span hyperfridge cuts out is exactly the committed string
    assert commitment["hostinfo"] == "host:main" and commitment["iban"] == "CH4308307000289537312"
    assert [(s["elctrnc_seq_nb"], s["amt"], s["ccy"], s["cd"]) for s in commitment["stmts"]] == [("247", "31709.14", "CHF", "OPBD"), ("248", "31709.09", "CHF", "OPBD")]


def test_serde_string_edge_cases():
    assert r0.serde_encode_str("") == b"\0\0\0\0"
    assert r0.serde_encode_str("abcd") == b"\4\0\0\0abcd" and r0.serde_encode_str("abcde") == b"\5\0\0\0abcde\0\0\0"
    assert r0.serde_decode_str(b"\5\0\0\0abcde\0\0\0tail") == (b"abcde", 12)  # a stream of several inputs: one frame consumed
    for bad, why in [(b"\5\0\0", "length word"), (b"\5\0\0\0abc", "only 3 follow"), (b"\5\0\0\0abcde\0\1\0", "padding")]:
        with pytest.raises(r0.R0HipError, match=why):
            r0.serde_decode_str(bad)
    euro = "Zürich €"
    assert r0.serde_decode_str(r0.serde_encode_str(euro))[0].decode("utf-8") == euro


def test_malformed_receipts_are_errors():
    for bad, why in [("{", "receipt JSON"), ('{"journal":{"bytes":[1,2]}}', 'no "inner"'), ('{"inner":"Fake","journal":{"bytes":[256]}}', "journal.bytes"),
                     ('{"inner":"Groth16","journal":{"bytes":[]}}', "unsupported inner"), ('{"inner":"Fake","journal":{"bytes":[]}} x', "trailing"),
                     ('{"inner":{"Composite":{"segments":[{"seal":[4294967296],"index":0,"hashfn":"poseidon2"}]}},"journal":{"bytes":[]}}', "segment.seal")]:
        with pytest.raises(r0.R0HipError, match=why):
            r0.Receipt.parse(bad)
    assert r0.Receipt.parse('{"inner":{"Fake":{"claim":null}},"journal":{"bytes":[]}}').kind == "Fake"  # risc0 3.x spelling of the variant


def test_composite_receipt_round_trip_and_verification(tmp_path):
    """A receipt carrying real seals: write, read back, check every seal with the host-side verifier (library and CLI)."""
    seal = np.load(os.path.join(GOLDEN, "seal_tiny_po2_9_seed_1.npy"))
    journal = r0.serde_encode_str(json.dumps({"hostinfo": "host:main", "iban": "CH4308307000289537312", "stmts": []}, separators=(",", ":")))
    text = r0.Receipt.new(journal, [seal, seal]).to_json()
    back = r0.Receipt.parse(text)
    assert back.kind == "Composite" and back.journal == journal and back.to_json() == text
    blob = np.fromfile(circuit_path("tiny"), dtype=np.uint32)
    for k, (index, s) in enumerate(back.seals()):
        assert index == k and np.array_equal(s, seal) and r0.verify_seal(blob, s)[0] == 0
    parsed = json.loads(text)  # and it is plain JSON any other reader can take
    assert parsed["inner"]["Composite"]["segments"][1]["hashfn"] == "poseidon2" and parsed["journal"]["bytes"] == list(journal)
    path = tmp_path / "receipt.json"
    path.write_text(text)
    # seals alone bind neither program nor journal: the CLI checks them, says so, and does NOT accept (exit status 1)
    out = subprocess.run([VERIFY, "--receipt", str(path), circuit_path("tiny")], capture_output=True, text=True)
    report = json.loads(out.stdout)
    assert out.returncode == 1 and report["accepted"] is False and report["seals_valid"] is True and report["journal_bound"] is False
    assert report["segments"] == 2 and json.loads(report["commitment"])["iban"] == "CH4308307000289537312"
    tampered = json.loads(text)
    tampered["inner"]["Composite"]["segments"][1]["seal"][100] ^= 1
    path.write_text(json.dumps(tampered, separators=(",", ":")))
    out = subprocess.run([VERIFY, "--receipt", str(path), circuit_path("tiny")], capture_output=True, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["seals_valid"] is False
    # a Fake receipt proves nothing: the verifier says so instead of accepting it
    out = subprocess.run([VERIFY, "--receipt", os.path.join(GOLDEN, FIXTURES[0]), circuit_path("tiny")], capture_output=True, text=True)
    assert out.returncode == 1 and "not a composite receipt" in out.stdout


def test_verify_cli_binds_the_journal_and_the_program(tmp_path, orc):
    """`verifier verify --imageid-hex .. --proof-json ..` (verifier/src/main.rs:210-232, 118-128): accepted only with the image id
    and the control root; a rewritten journal beside valid seals is refused."""
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    c = orc.circuit(blob)
    po2 = 9
    journal = r0.serde_encode_str('{"iban":"CH4308307000289537312"}')
    claims, image_id = r0.session_claims(2, journal)
    seals = []
    for k, cl in enumerate(claims):
        code, data, glob = c.witgen(po2, 40 + k, globals_in=cl.globals())
        seals.append(c.prove(po2, code, data, glob))
    root = c.code_root(code, po2)
    args = ["--image-id", r0.image_id_to_hex(image_id), "--control-root", "%d:%s" % (po2, ",".join(str(int(w)) for w in root))]
    path = tmp_path / "receipt.json"
    path.write_text(r0.Receipt.new(journal, seals, claims).to_json())
    out = subprocess.run([VERIFY, "--receipt", str(path), circuit_path("small")] + args, capture_output=True, text=True)
    report = json.loads(out.stdout)
    assert out.returncode == 0 and report["accepted"] is True and report["journal_bound"] is True and report["commitment"] == '{"iban":"CH4308307000289537312"}'
    path.write_text(r0.Receipt.new(r0.serde_encode_str('{"iban":"XX00"}'), seals, claims).to_json())
    out = subprocess.run([VERIFY, "--receipt", str(path), circuit_path("small")] + args, capture_output=True, text=True)
    report = json.loads(out.stdout)
    assert out.returncode == 1 and report["accepted"] is False and report["seals_valid"] is True and "journal" in report["reason"]
    out = subprocess.run([VERIFY, "--receipt", str(path), circuit_path("small"), "--image-id", "zz"], capture_output=True, text=True)
    assert out.returncode == 2 and "64 hex digits" in out.stderr


def test_executor_env_input_stream_of_the_reference_test_inputs():
    """The 13 inputs of host/src/main.rs:389-417 in order, from the reference's own test files (tests/golden/camt53/, copied
    unchanged from data/test/): 12 String frames and the Vec<u8> transaction key; the guest-side reads
    (methods/guest/src/main.rs:159-171) are replayed on the words."""
    d = os.path.join(GOLDEN, "camt53")
    text = lambda name: open(os.path.join(d, "test.xml-" + name), encoding="utf-8").read()
    tx_key = open(os.path.join(d, "test.xml-TransactionKeyDecrypt.bin"), "rb").read()
    assert len(tx_key) == 256
    inputs = [text("SignedInfo"), text("authenticated"), text("SignatureValue"), text("OrderData"), "2519…modulus", "65537", "-----BEGIN PRIVATE KEY-----…",
              tx_key, "CH4308307000289537312", "host:main", text("Witness.hex"), "-----BEGIN PUBLIC KEY-----…", "verbose"]
    words = r0.env_input_words(inputs)
    pos = 0
    for item in inputs:  # env::read::<String>() / env::read::<Vec<u8>>()
        n = int(words[pos])
        if isinstance(item, str):
            raw = item.encode("utf-8")
            nw = (n + 3) // 4
            assert n == len(raw) and words[pos + 1:pos + 1 + nw].tobytes()[:n] == raw and not any(words[pos + 1:pos + 1 + nw].tobytes()[n:])
            assert r0.serde_encode_str(item) == words[pos:pos + 1 + nw].tobytes()  # the same frame the journal fixture pins
        else:
            nw = n
            assert n == len(item) and words[pos + 1:pos + 1 + n].tolist() == list(item)
        pos += 1 + nw
    assert pos == words.size
