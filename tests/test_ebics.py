"""The native EBICS pre-processor (csrc/ebics.cpp) against the reference's own files -- the one row of
the scope table that reference-held fixtures pin.  tests/golden/camt53/ holds, copied unchanged: the raw response
(data/response_template-generated.xml, byte-identical to data/test/test.xml-generated.xml), the three PUBLIC keys (data/pub_*.pem)
and the six pre-processed files data/test/test.xml-* that data/checkResponse.sh made from that response with xmllint / openssl /
zlib-flate.  Expected values: those files, and the known answers SURVEY.md section 4 derived from them (`BUuyFKUr...`, `6ce63c3d...`,
the 00 at offset 239, the statement values of methods/guest/src/test_xmlparse.rs:225-249).  hashlib / zlib / binascii of the Python
standard library serve as independent implementations; nothing of the reference is executed."""
import base64
import binascii
import hashlib
import os
import zlib

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import ROOT

D = os.path.join(ROOT, "tests", "golden", "camt53")
rd = lambda name, mode="rb": open(os.path.join(D, name), mode).read()


@pytest.fixture(scope="module")
def ebics():
    return r0.Ebics(rd("response.xml"))


def test_the_four_guest_inputs_are_cut_and_canonicalised_exactly_as_the_script_does(ebics):
    """data/checkResponse.sh:151-155, 192, 200, 221 -> `<xml>-authenticated`, `-SignedInfo`, `-SignatureValue`, `-OrderData`."""
    assert ebics.part(r0.Ebics.AUTHENTICATED) == rd("test.xml-authenticated")
    assert ebics.part(r0.Ebics.SIGNED_INFO) == rd("test.xml-SignedInfo")
    assert ebics.part(r0.Ebics.SIGNATURE_VALUE) == rd("test.xml-SignatureValue")
    assert ebics.part(r0.Ebics.ORDER_DATA) == rd("test.xml-OrderData")
    assert len(ebics.part(r0.Ebics.AUTHENTICATED)) == 1217 and len(ebics.part(r0.Ebics.ORDER_DATA)) == 3843


def test_digest_value_and_bank_signature(ebics):
    """methods/guest/src/test_xmlparse.rs:43-86 on the same files."""
    assert ebics.part(r0.Ebics.DIGEST_VALUE) == b"BUuyFKUrSlvHaXjTC+Jo1h9myiVZakJ8SqjseZdQLyw="
    assert base64.b64encode(hashlib.sha256(rd("test.xml-authenticated")).digest()) == ebics.part(r0.Ebics.DIGEST_VALUE)
    assert ebics.check_digest() is True
    assert hashlib.sha256(ebics.part(r0.Ebics.SIGNED_INFO)).hexdigest().startswith("6ce63c3d") and hashlib.sha256(ebics.part(r0.Ebics.SIGNED_INFO)).hexdigest().endswith("07f5bd")
    assert ebics.verify_bank_signature(rd("pub_bank.pem")) is True
    # independent check of the same identity with Python integers: sig^e mod n ends in SHA-256(SignedInfo) behind the PKCS#1 prefix
    n, e = (int(x) for x in r0.rsa_public_key_decimal(rd("pub_bank.pem")))
    em = pow(int.from_bytes(ebics.part(r0.Ebics.SIGNATURE_BIN), "big"), e, n).to_bytes(256, "big")
    assert em[:2] == b"\x00\x01" and set(em[2:204]) == {0xff} and em[-32:] == hashlib.sha256(ebics.part(r0.Ebics.SIGNED_INFO)).digest()
    assert e == 65537 and n.bit_length() == 2048
    # the wrong key, a flipped signature bit, a changed header: refused (docs/INSTRUCTIONS.md:267-292 lists these as the manual negative tests)
    assert ebics.verify_bank_signature(rd("pub_witness.pem")) is False
    xml = rd("response.xml")
    i = xml.index(b"<ds:SignatureValue>") + 30
    bad = r0.Ebics(xml[:i] + (b"B" if xml[i:i + 1] != b"B" else b"C") + xml[i + 1:])
    assert bad.check_digest() is True and bad.verify_bank_signature(rd("pub_bank.pem")) is False
    tampered = r0.Ebics(xml.replace(b"<ReturnCode>000000</ReturnCode>", b"<ReturnCode>000001</ReturnCode>"))
    assert tampered.check_digest() is False


def test_transaction_key_witness_decrypt_inflate_unzip(ebics):
    """methods/guest/src/test_xmlparse.rs:89-190 (supplied-key shortcut) and data/checkResponse.sh:231-298."""
    raw = rd("test.xml-TransactionKeyDecrypt.bin")
    assert len(raw) == 256 and raw[:2] == b"\x00\x02" and raw[239] == 0 and 0 not in raw[2:239]
    ok, key = ebics.check_transaction_key(rd("pub_client.pem"), raw)
    assert ok is True and key == raw[240:]
    n, e = (int(x) for x in r0.rsa_public_key_decimal(rd("pub_client.pem")))
    assert pow(int.from_bytes(raw, "big"), e, n).to_bytes(256, "big") == ebics.part(r0.Ebics.TRANSACTION_KEY_BIN)
    flipped = bytes([raw[0], raw[1], raw[2] ^ 1]) + raw[3:]
    assert ebics.check_transaction_key(rd("pub_client.pem"), flipped)[0] is False
    assert ebics.check_transaction_key(rd("pub_bank.pem"), raw)[0] is False
    # witness signature over SHA-256 of the decoded order data (2864 bytes, a whole number of AES blocks)
    ct = ebics.part(r0.Ebics.ORDER_DATA_BIN)
    assert len(ct) == 2864 and len(ct) % 16 == 0
    assert ebics.verify_witness(rd("pub_witness.pem"), rd("test.xml-Witness.hex")) is True
    assert ebics.verify_witness(rd("pub_bank.pem"), rd("test.xml-Witness.hex")) is False
    # AES-128-CBC zero IV, inflate, unzip
    docs = ebics.decrypt_order_data(key)
    assert len(docs) >= 1 and all(name.endswith(".xml") and data.startswith(b"<?xml") for name, data in docs)
    payload = ebics.part(r0.Ebics.PAYLOAD_ZIP)
    assert payload[:4] == b"PK\x03\x04"
    # the statement the guest commits to (methods/guest/src/test_xmlparse.rs:225-249; the reference's journal fixture)
    text = b"".join(d for _, d in docs)
    for needle in (b"<ElctrncSeqNb>247</ElctrncSeqNb>", b"CH4308307000289537312", b"31709.14", b"OPBD", b"CHF"):
        assert needle in text, needle
    with pytest.raises(r0.R0HipError, match="zlib|inflate"):
        r0.Ebics(rd("response.xml")).decrypt_order_data(bytes(16))  # the wrong key does not decrypt to a zlib stream


def test_primitives_against_published_vectors_and_the_standard_library():
    # FIPS-197 appendix C.1
    key, pt = bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff")
    ct = bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")
    assert r0.aes128_block(key, pt) == ct and r0.aes128_block(key, ct, decrypt=True) == pt
    # FIPS-197 appendix B
    assert r0.aes128_block(bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c"), bytes.fromhex("3243f6a8885a308d313198a2e0370734")) == bytes.fromhex("3925841d02dc09fbdc118597196a0b32")
    rng = np.random.default_rng(3)
    cases = [b"", b"a", b"hello hello hello hello hello", bytes(70000), rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(),
             (b"<Ntry><Amt Ccy=\"CHF\">31709.14</Amt></Ntry>" * 500)]
    for data in cases:
        for level in (0, 1, 6, 9):  # stored, fixed and dynamic Huffman blocks
            assert r0.zlib_inflate(zlib.compress(data, level)) == data
        assert r0.zlib_inflate(zlib.compress(data) + b"\x07" * 7) == data  # trailing padding bytes are ignored, as the script's zlib-flate does
    bad = bytearray(zlib.compress(cases[2]))
    bad[-1] ^= 1
    with pytest.raises(r0.R0HipError, match="Adler"):
        r0.zlib_inflate(bytes(bad))
    with pytest.raises(r0.R0HipError):
        r0.zlib_inflate(b"\x78\x9c\xff\xff")


def test_the_thirteen_executor_env_inputs(ebics):
    """host/src/main.rs:389-417: the frames in order, built from the response itself instead of the pre-processed files."""
    tx = rd("test.xml-TransactionKeyDecrypt.bin")
    witness = rd("test.xml-Witness.hex", "r")
    words = ebics.env_inputs(rd("pub_bank.pem"), "-----BEGIN PRIVATE KEY-----…", tx, "CH4308307000289537312", "host:main", witness, rd("pub_witness.pem"), "verbose")
    n, e = r0.rsa_public_key_decimal(rd("pub_bank.pem"))
    text = lambda name: rd("test.xml-" + name).decode("utf-8")
    want = r0.env_input_words([text("SignedInfo"), text("authenticated"), text("SignatureValue"), text("OrderData"), n, e, "-----BEGIN PRIVATE KEY-----…", tx,
                               "CH4308307000289537312", "host:main", witness, rd("pub_witness.pem").decode(), "verbose"])
    assert np.array_equal(words, want)


def test_malformed_responses_are_errors():
    xml = rd("response.xml")
    for bad, why in [(b"<x/>", "xmlns"), (xml.replace(b"<ds:DigestValue>", b"<ds:DigestValu>"), "DigestValue|mismatched"),
                     (xml.replace(b"</header>", b"</heade>", 1), "header|mismatched"), (xml.replace(b"<OrderData>", b"<OrderData>!"), "base64")]:
        with pytest.raises(r0.R0HipError, match=why):
            r0.Ebics(bad)
    with pytest.raises(r0.R0HipError, match="PUBLIC KEY"):
        r0.Ebics(xml).verify_bank_signature(b"-----BEGIN NOTHING-----")


def test_preprocess_cli_writes_the_files_the_host_reads(tmp_path):
    """`r0h_preprocess` in the place of data/checkResponse.sh (host/src/main.rs:143-151 runs that script as a child process, then
    reads `<xml>-*`, :206-227): same files, byte for byte; exit status carries the verdict."""
    import json
    import shutil
    import subprocess
    cli = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_preprocess")
    xml = tmp_path / "test.xml"
    shutil.copy(os.path.join(D, "response.xml"), xml)
    args = [cli, str(xml), "--pub-bank", os.path.join(D, "pub_bank.pem"), "--pub-client", os.path.join(D, "pub_client.pem"), "--pub-witness", os.path.join(D, "pub_witness.pem"),
            "--tx-key-raw", os.path.join(D, "test.xml-TransactionKeyDecrypt.bin"), "--witness-hex", os.path.join(D, "test.xml-Witness.hex")]
    out = subprocess.run(args + ["--out-dir", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    rep = json.loads(out.stdout)
    assert rep == {"ok": True, "digest": True, "bank_signature": True, "transaction_key": True, "witness_signature": True, "documents": rep["documents"]} and rep["documents"] >= 1
    for name in ("authenticated", "SignedInfo", "SignatureValue", "OrderData", "TransactionKeyDecrypt.bin", "Witness.hex"):
        assert (tmp_path / ("test.xml-" + name)).read_bytes() == rd("test.xml-" + name), name
    assert any(p.name.startswith("test.xml-camt53-") and p.read_bytes().startswith(b"<?xml") for p in tmp_path.iterdir())
    swapped = list(args)
    swapped[swapped.index("--pub-bank") + 1] = os.path.join(D, "pub_witness.pem")  # the wrong bank key (docs/INSTRUCTIONS.md:267-292)
    out = subprocess.run(swapped, capture_output=True, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["bank_signature"] is False and json.loads(out.stdout)["digest"] is True
    assert subprocess.run([cli, str(tmp_path / "missing.xml")], capture_output=True).returncode == 2


def test_private_key_steps_reproduce_the_fixture_files(tmp_path):
    """data/checkResponse.sh:231-236 and 276-279 (`openssl pkeyutl -decrypt ... rsa_padding_mode:none`, `-sign ... digest:sha256`) with
    the reference's PRIVATE test keys (data/client.pem, data/witness.pem, copied as fixtures): the raw transaction-key block is
    `<xml>-TransactionKeyDecrypt.bin` and the deterministic PKCS#1 v1.5 witness signature is `<xml>-Witness.hex`, byte for byte
    (methods/guest/src/test_xmlparse.rs:89-136 is the same decryption inside the guest)."""
    import json
    import shutil
    import subprocess
    e = r0.Ebics(rd("response.xml"))
    ok, raw, key = e.decrypt_transaction_key(rd("client.pem"))
    assert ok is True and raw == rd("test.xml-TransactionKeyDecrypt.bin") and key == raw[-16:] and len(key) == 16
    assert e.witness_sign(rd("witness.pem")) == rd("test.xml-Witness.hex")
    assert e.verify_witness(rd("pub_witness.pem"), e.witness_sign(rd("witness.pem"))) is True
    # the wrong private key does not give a padded block; a public key is not a private key
    assert e.decrypt_transaction_key(rd("witness.pem"))[0] is False
    with pytest.raises(r0.R0HipError, match="PRIVATE KEY"):
        e.decrypt_transaction_key(rd("pub_client.pem"))
    # the compiled pre-processor with the keys instead of the two files: the same six files
    cli = os.path.join(ROOT, "hyperfridge-r0_amd", "r0h_preprocess")
    xml = tmp_path / "test.xml"
    shutil.copy(os.path.join(D, "response.xml"), xml)
    out = subprocess.run([cli, str(xml), "--pub-bank", os.path.join(D, "pub_bank.pem"), "--pub-client", os.path.join(D, "pub_client.pem"), "--pub-witness", os.path.join(D, "pub_witness.pem"),
                          "--client-key", os.path.join(D, "client.pem"), "--witness-key", os.path.join(D, "witness.pem"), "--out-dir", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and json.loads(out.stdout)["ok"] is True, out.stdout + out.stderr
    for name in ("authenticated", "SignedInfo", "SignatureValue", "OrderData", "TransactionKeyDecrypt.bin", "Witness.hex"):
        assert (tmp_path / ("test.xml-" + name)).read_bytes() == rd("test.xml-" + name), name
