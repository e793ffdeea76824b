"""The whole hyperfridge guest pipeline as a hand-assembled RV32IM program (tools/guest_camt53.py): SHA-256, three RSA-2048
public-key operations, AES-128-CBC, zlib / deflate, ZIP, camt.053 field extraction, PEM re-encoding of the three keys -- run by
this library's executor on the reference's own inputs.

PINNED BY REFERENCE-HELD FIXTURES: the journal the guest commits is `journal.bytes` of the reference's two committed receipts, byte
for byte -- data/test/test.xml-Receipt-6bb958..-latest.json (the current commitment form: hostinfo, iban, the three PEM keys, the
statements; methods/guest/src/main.rs:214-262) and data/test/test.xml-Receipt-test.json (the earlier form without the keys) -- from
data/test/test.xml-* and the public keys (tests/golden/camt53/, copies of the reference's data files).  The stages in between are
checked against Python's zlib / zipfile / hashlib and the product's own AES block function; payloads the reference does not hold
(stored and fixed-Huffman deflate blocks, stored ZIP members, other accounts, corrupted streams) are signed here with a throw-away
RSA key generated in the test."""
import base64
import hashlib
import io
import json
import os
import random
import struct
import sys
import zipfile
import zlib

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import ROOT, circuit_path

sys.path.insert(0, os.path.join(ROOT, "tools"))
import guest_camt53  # noqa: E402
import guest_rsa  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
journal_of = lambda name: bytes(json.load(open(os.path.join(GOLDEN, name)))["journal"]["bytes"])


@pytest.fixture(scope="module")
def guest():
    image, labels, layout = guest_camt53.build()
    return dict(elf=image, layout=layout, ref=guest_rsa.reference_inputs())


def run(guest, iban=guest_camt53.REFERENCE_IBAN, host="host:main", form=1, inputs=None, po2=20):
    vm = r0.Vm()
    vm.load_elf(guest["elf"])
    inputs = dict(inputs or guest["ref"])
    authenticated = inputs.pop("authenticated", None)
    vm.set_input(guest_camt53.input_stream(iban, host, guest_camt53.reference_authenticated() if authenticated is None else authenticated, form=form, **inputs))
    return vm, vm.run(segment_po2=po2, max_cycles=200_000_000)


def mem(vm, layout, sym, n):
    addr = layout[sym]
    raw = vm.read(addr & ~3, (n + 7) // 4 + 1).tobytes()
    return raw[addr & 3:(addr & 3) + n]


@pytest.mark.parametrize("form,fixture", [(1, "reference_receipt_6bb95807_latest.json"), (0, "reference_receipt_test.json")])
def test_the_guest_commits_the_journal_of_the_references_receipt_byte_for_byte(guest, form, fixture):
    vm, (kind, code) = run(guest, form=form)
    assert (kind, code) == (0, 0)
    want = journal_of(fixture)
    assert vm.journal == want and len(want) in (336, 1772)
    doc = json.loads(r0.journal_commitment(vm.journal))
    assert doc["iban"] == "CH4308307000289537312" and [s["elctrnc_seq_nb"] for s in doc["stmts"]] == ["247", "248"]
    assert [(s["amt"], s["ccy"], s["cd"]) for s in doc["stmts"]] == [("31709.14", "CHF", "OPBD"), ("31709.09", "CHF", "OPBD")]  # methods/guest/src/test_xmlparse.rs:225-249
    if form:
        D = os.path.join(GOLDEN, "camt53")
        for key, pem in (("pub_bank_pem", "pub_bank.pem"), ("pub_witness_pem", "pub_witness.pem"), ("pub_client_pem", "pub_client.pem")):
            assert doc[key] == open(os.path.join(D, pem)).read()  # the guest's own DER + base64 of the modulus it verified with
    assert 11_000_000 < vm.cycles < 13_000_000 and len(vm.segments()) >= 11


def test_every_stage_of_the_pipeline_against_an_independent_implementation(guest):
    vm, (kind, code) = run(guest)
    L, ref = guest["layout"], guest["ref"]
    assert (kind, code) == (0, 0)
    key = ref["tx_plain"][-16:]
    assert mem(vm, L, "KEY", 16) == key and ref["tx_plain"][:2] == b"\x00\x02" and ref["tx_plain"][239] == 0
    # AES-128-CBC with a zero ICV through the product's single-block function (FIPS-197 vectors pin it: tests/test_ebics.py)
    ct, prev, plain = ref["order_data"], bytes(16), b""
    for k in range(0, len(ct), 16):
        block = r0.aes128_block(key, ct[k:k + 16], decrypt=True)
        plain += bytes(x ^ y for x, y in zip(block, prev))
        prev = ct[k:k + 16]
    assert mem(vm, L, "PLAIN", len(plain)) == plain and 1 <= plain[-1] <= 16
    archive = zlib.decompress(plain[:-plain[-1]])
    assert mem(vm, L, "ZIPBUF", len(archive)) == archive
    members = zipfile.ZipFile(io.BytesIO(archive))
    names = members.namelist()
    assert len(names) == 3 and mem(vm, L, "DOC", members.getinfo(names[-1]).file_size) == members.read(names[-1])  # the last member inflated
    assert mem(vm, L, "DIGEST", 32) == b"".join(struct.pack("<I", w) for w in struct.unpack(">8I", hashlib.sha256(ref["signed_info"]).digest()))


# ---- payloads of our own, signed with a throw-away key
def _is_prime(n, rng):
    if n % 2 == 0:
        return False
    for p in (3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    for _ in range(12):
        x = pow(rng.randrange(2, n - 1), d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


@pytest.fixture(scope="module")
def toy_key():
    rng = random.Random(53)
    primes = []
    while len(primes) < 2:
        c = rng.getrandbits(1024) | (3 << 1022) | 1
        if c % 65537 != 1 and _is_prime(c, rng):
            primes.append(c)
    p, q = primes
    n = p * q
    assert n.bit_length() == 2048
    return dict(n=n, d=pow(65537, -1, (p - 1) * (q - 1)))


def _sign(key, message):
    em = b"\x00\x01" + b"\xff" * 202 + b"\x00" + guest_rsa.DER_SHA256 + hashlib.sha256(message).digest()
    return pow(int.from_bytes(em, "big"), key["d"], key["n"]).to_bytes(256, "big")


AUTHENTICATED = b'<header authenticate="true"><static><HostID>TOY</HostID></static></header><DataEncryptionInfo authenticate="true"/>'


def _signed_info(authenticated):
    return (b'<ds:SignedInfo><ds:Reference URI="#xpointer(//*[@authenticate=\'true\'])"><ds:DigestValue>'
            + base64.b64encode(hashlib.sha256(authenticated).digest()) + b"</ds:DigestValue></ds:Reference></ds:SignedInfo>")


def _inputs(key, payload_zlib, aes_key=bytes(range(16)), signed_info=None, pad=None, authenticated=AUTHENTICATED):
    signed_info = _signed_info(authenticated) if signed_info is None else signed_info
    pad = 16 - len(payload_zlib) % 16 if pad is None else pad
    plain = payload_zlib + bytes(pad - 1) + bytes([pad & 0xFF]) if pad else payload_zlib
    ct, prev = b"", bytes(16)
    for k in range(0, len(plain), 16):
        prev = r0.aes128_block(aes_key, bytes(x ^ y for x, y in zip(plain[k:k + 16], prev)))
        ct += prev
    block = b"\x00\x02" + bytes([7] * 237) + b"\x00" + aes_key
    return dict(signed_info=signed_info, bank_sig=_sign(key, signed_info), bank_n=key["n"], tx_plain=block, client_n=key["n"],
                tx_cipher=pow(int.from_bytes(block, "big"), 65537, key["n"]).to_bytes(256, "big"), order_data=ct, witness_sig=_sign(key, ct),
                witness_n=key["n"], authenticated=authenticated)


def _doc(iban, seq, amount, ccy="EUR", cd="OPBD", day="2024-02-29"):
    return ('<?xml version="1.0"?><Document><BkToCstmrStmt><Stmt><ElctrncSeqNb>%d</ElctrncSeqNb><FrToDt><FrDtTm>%sT00:00:00</FrDtTm><ToDtTm>%sT23:59:59</ToDtTm></FrToDt>'
            '<Acct><Id><IBAN>%s</IBAN></Id></Acct><Bal><Tp><CdOrPrtry><Cd>%s</Cd></CdOrPrtry></Tp><Amt Ccy="%s">%s</Amt></Bal><Bal><Tp><CdOrPrtry><Cd>CLBD</Cd></CdOrPrtry></Tp>'
            '<Amt Ccy="%s">1.00</Amt></Bal><Ntry><NtryDtls><TxDtls><RltdPties><DbtrAcct><Id><IBAN>%s</IBAN></Id></DbtrAcct></RltdPties></TxDtls></NtryDtls></Ntry></Stmt></BkToCstmrStmt></Document>'
            % (seq, day, day, iban, cd, ccy, amount, ccy, iban)).encode()


def _zip(docs, method):
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", method) as z:
        for k, d in enumerate(docs):
            z.writestr("camt53/doc_%d.xml" % k, d)
    return buf.getvalue()


def _want(host, iban, stmts):
    text = '{"hostinfo":"%s","iban":"%s","stmts":[%s]}' % (host, iban, ",".join(
        '{"elctrnc_seq_nb":"%d","fr_dt_tm":"%sT00:00:00","to_dt_tm":"%sT23:59:59","amt":"%s","ccy":"%s","cd":"%s"}' % s for s in stmts))
    return r0.serde_encode_str(text)


def test_other_statements_other_containers_other_deflate_block_types(guest, toy_key):
    mine, other = "DE02120300000000202051", "FR7630006000011234567890189"
    docs = [_doc(other, 1, "5.00"), _doc(mine, 17, "1234.56"), _doc(mine, 18, "-7.10", ccy="USD", cd="PRCD", day="2024-03-01"), _doc(other, 2, "6.00")]
    stmts = [(17, "2024-02-29", "2024-02-29", "1234.56", "EUR", "OPBD"), (18, "2024-03-01", "2024-03-01", "-7.10", "USD", "PRCD")]
    fixed = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    cases = {
        "deflated members in a dynamic-Huffman stream": zlib.compress(_zip(docs, zipfile.ZIP_DEFLATED), 9),
        "stored members in a stored stream": zlib.compress(_zip(docs, zipfile.ZIP_STORED), 0),
        "stored members, fast dynamic stream": zlib.compress(_zip(docs, zipfile.ZIP_STORED), 1),
        "deflated members, fixed-Huffman stream": (lambda z: fixed.compress(z) + fixed.flush())(_zip(docs, zipfile.ZIP_DEFLATED)),
    }
    for what, stream in cases.items():
        vm, (kind, code) = run(guest, iban=mine, host="h", form=0, inputs=_inputs(toy_key, stream))
        assert (kind, code) == (0, 0), what
        assert vm.journal == _want("h", mine, stmts), what
    # 8 KiB of incompressible bytes beside the documents: stored blocks inside a level-9 stream, long matches, every length code
    rng = np.random.default_rng(5)
    noisy = _zip([bytes(rng.integers(0, 256, 6000, dtype=np.uint8)), _doc(mine, 3, "0.01") + b" " * 3000 + b"abcabcabc" * 300], zipfile.ZIP_DEFLATED)
    vm, (kind, code) = run(guest, iban=mine, host="h", form=0, inputs=_inputs(toy_key, zlib.compress(noisy, 9)))
    assert (kind, code) == (0, 0) and vm.journal == _want("h", mine, [(3, "2024-02-29", "2024-02-29", "0.01", "EUR", "OPBD")])
    assert mem(vm, guest["layout"], "ZIPBUF", len(noisy)) == noisy


def test_what_the_guest_refuses(guest, toy_key):
    mine = "DE02120300000000202051"
    good = zlib.compress(_zip([_doc(mine, 1, "1.00")], zipfile.ZIP_DEFLATED), 9)
    base = _inputs(toy_key, good)
    flip = lambda b, k: bytes(b[:k]) + bytes([b[k] ^ 1]) + bytes(b[k + 1:])
    cases = [
        (dict(iban="DE00000000000000000000"), base, 8),                                            # no statement of that account
        (dict(iban=mine[:-1]), base, 8),                                                           # a prefix of the account is not the account
        (dict(iban=mine), _inputs(toy_key, good[:-3] + b"\x00\x00\x00"), 6),                       # Adler-32 does not match
        (dict(iban=mine), _inputs(toy_key, good[:40]), 6),                                         # the stream ends early
        (dict(iban=mine), _inputs(toy_key, b"\x78\x9d" + good[2:]), 6),                            # header check bits
        (dict(iban=mine), _inputs(toy_key, good, pad=0) if len(good) % 16 == 0 else _inputs(toy_key, good + b"\x00" * (16 - len(good) % 16), pad=0), 9),  # no padding byte
        (dict(iban=mine), dict(base, tx_plain=b"\x00\x01" + base["tx_plain"][2:], tx_cipher=pow(int.from_bytes(b"\x00\x01" + base["tx_plain"][2:], "big"), 65537, toy_key["n"]).to_bytes(256, "big")), 9),
        (dict(iban=mine), dict(base, order_data=flip(base["order_data"], 5)), 3),                  # the witness signed other data
        (dict(iban=mine), dict(base, bank_sig=flip(base["bank_sig"], 200)), 1),
        (dict(iban=mine), dict(base, authenticated=flip(AUTHENTICATED, 30)), 10),                  # the signed digest is of other header data
        (dict(iban=mine), dict(base, authenticated=AUTHENTICATED + b" "), 10),
        (dict(iban=mine), _inputs(toy_key, good, signed_info=b"<ds:SignedInfo>no digest in here</ds:SignedInfo>"), 10),
        (dict(iban=mine), _inputs(toy_key, good, signed_info=_signed_info(AUTHENTICATED)[:-60]), 10),  # the value is cut short
        (dict(iban=mine), dict(base, authenticated=bytes(guest_rsa.MAX_MSG + 1)), 5),
    ]
    for kw, inputs, want in cases:
        vm, (kind, code) = run(guest, host="h", form=0, inputs=inputs, **kw)
        assert (kind, code) == (0, want), (kw, want, code)
        assert vm.journal == b""
    # a member whose header lies about its size
    z = bytearray(_zip([_doc(mine, 1, "1.00")], zipfile.ZIP_DEFLATED))
    z[22] ^= 1
    vm, (kind, code) = run(guest, iban=mine, host="h", form=0, inputs=_inputs(toy_key, zlib.compress(bytes(z), 9)))
    assert (kind, code) == (0, 7)
    # ... or about its checksum (stored member: the byte changed is the document's, the archive stays well-formed)
    for method in (zipfile.ZIP_STORED, zipfile.ZIP_DEFLATED):
        z = bytearray(_zip([_doc(mine, 1, "1.00")], method))
        z[14] ^= 0x10
        vm, (kind, code) = run(guest, iban=mine, host="h", form=0, inputs=_inputs(toy_key, zlib.compress(bytes(z), 9)))
        assert (kind, code) == (0, 11) and vm.journal == b""
    z = bytearray(_zip([_doc(mine, 1, "1.00")], zipfile.ZIP_STORED))
    z[30 + len("camt53/doc_0.xml") + 100] ^= 0x20  # a byte of the document itself
    vm, (kind, code) = run(guest, iban=mine, host="h", form=0, inputs=_inputs(toy_key, zlib.compress(bytes(z), 9)))
    assert (kind, code) == (0, 11)


def _last_segment(image, stream):
    """the run executed with the trace kept one segment at a time (earlier segments' rows released): -> (vm, index of the last)"""
    vm = r0.Vm()
    vm.load_elf(image)
    vm.set_input(stream)
    while True:
        finished, kind, code = vm.run_segment(segment_po2=20, keep_trace=True, boundary_rows=True)
        k = len(vm.segments()) - 1
        if finished:
            assert (kind, code) == (0, 0)
            return vm, k
        vm.release_trace(k)


def test_the_last_segment_of_the_real_run_proves_on_the_cpu(orc):
    """What is left of the guest's 11.8 M cycles after eleven full segments (0.3 M cycles: the end of the camt.053 scan, the PEM
    re-encoding, the COMMIT, the HALT) as a 2^19-row trace: the CPU oracle proves it with the trace circuit and both verifiers accept,
    bound to the control root; the public inputs say HALT with exit code 0."""
    image, stream, _ = guest_camt53.elf_and_input(form=1)
    vm, k = _last_segment(image, stream)
    seg = vm.segments()[k]
    assert r0.compute_image_id(image) == bytes(vm.segments()[0].pre.digest()) != r0.compute_image_id(image[:-4] + bytes(4))
    assert k == 11 and seg.user_cycles + seg.boundary_rows <= 1 << 19 and vm.journal == journal_of("reference_receipt_6bb95807_latest.json")
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    oc = orc.circuit(blob)
    data, glob = vm.trace_witness(k, 19, claim_globals=vm.claims()[k].globals())
    bl = vm.boundary(k)
    assert seg.closing == 1 and [orc.dec(int(g)) for g in glob[8:20]] == [seg.pre.pc, seg.post.pc, seg.user_cycles, 1, 1, 0, 0, k + 1, 1, 0, bl[0].addr, bl[-1].addr]
    # the rows that close the session: every word of the image among them, each with the word the ELF holds there
    image = {b.addr: b.init_value for b in bl if b.flags & 1}
    assert len(image) > 3000 and all(b.init_value == 0 for b in bl if not b.flags & 1) and any(b.prev_seg == 0 for b in bl) and max(b.prev_seg for b in bl) == k
    code = oc.witgen(19, 0)[0]
    root = oc.code_root(code, 19)
    glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(v) for v in range(7, 23)]  # one seal outside its session: any challenge will do
    glob = oc.logup_totals(19, code, data, glob)
    seal = oc.prove(19, code, data, glob)
    assert oc.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")


@pytest.mark.gpu
def test_the_proved_receipt_carries_the_references_journal(hal, orc):
    """`prove(env, elf)` over this guest with the trace circuit: twelve 2^20-row segments expanded and proved on the device; the
    receipt verifies with the ELF -- THIS program produced THIS journal: seals, claims, the session's challenge, and the balance of
    the segments' sums with the ELF's image words and the journal's words -- and its journal is the reference's committed
    receipt's journal.  Parity at the headline size on the real workload: the run's FIRST segment (a full 2^20 rows) and its last
    one, expanded on the host and proved by the CPU oracle under the seals' own public inputs, give the device's seals word for word."""
    import __graft_entry__ as entry
    image, stream, _ = guest_camt53.elf_and_input(form=1)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    receipt, image_id, cycles = hal.prove_elf(gc, image, stream, segment_po2=20)
    assert receipt.journal == journal_of("reference_receipt_6bb95807_latest.json") and cycles > 11_000_000
    assert image_id == r0.compute_image_id(image)  # what a verifier derives from the ELF alone (risc0: compute_image_id / HYPERFRIDGE_ID)
    seals = receipt.seals()
    roots = {}
    for _, seal in seals:
        size = r0.verify_seal(blob, seal)[2]
        if size not in roots:
            cc = hal.code_commit(gc, size)
            roots[size] = cc.root()
            cc.free()
    assert len(seals) >= 11 and receipt.verify(blob, roots, None, elf=image)[:2] == (0, "ok") and receipt.verify(blob, roots, image_id)[0] == 15
    # what `receipt.verify(image_id)` exists to refuse (verifier/src/main.rs:124-126): another program, another journal
    other = bytearray(image)
    other[len(other) // 2] ^= 4
    assert receipt.verify(blob, roots, None, elf=bytes(other))[0] in (8, 14)
    doc = json.loads(receipt.to_json())
    doc["journal"]["bytes"][40] ^= 1
    assert r0.Receipt.parse(json.dumps(doc)).verify(blob, roots, None, elf=image)[0] == 7  # (the output digest no longer matches; with it recomputed the session sum objects: tests/test_trace_circuit.py)
    oc = orc.circuit(blob)
    assert oc.verify(seals[-1][1], code_root=roots[r0.verify_seal(blob, seals[-1][1])[2]]) == (0, "ok")
    # parity at full size on the real workload: the run's last segment -- what is left of 11.8 M cycles after eleven full segments, and
    # the HALT -- expanded on the host and proved by the CPU oracle gives the device's seal, word for word
    vm, k = _last_segment(image, stream)
    assert k == len(seals) - 1 and vm.journal == receipt.journal
    size = r0.verify_seal(blob, seals[k][1])[2]
    seg = vm.segments()[k]
    assert seg.user_cycles + seg.boundary_rows <= 1 << size and size >= 17
    data, glob = vm.trace_witness(k, size, claim_globals=vm.claims()[k].globals())
    assert np.array_equal(glob[:20], seals[k][1][:20])
    ocode = oc.witgen(size, 0)[0]
    assert np.array_equal(oc.prove(size, ocode, data, seals[k][1][:r0.TRACE_GLOBALS]), seals[k][1])
    del data
    # ... and the first segment, a full 2^20-row trace of the real workload
    vm0 = r0.Vm()
    vm0.load_elf(image)
    vm0.set_input(stream)
    assert vm0.run_segment(segment_po2=20, keep_trace=True, boundary_rows=True)[0] is False
    s0 = vm0.segments()[0]
    assert r0.verify_seal(blob, seals[0][1])[2] == 20 and (1 << 19) < s0.user_cycles + s0.boundary_rows <= 1 << 20
    data, glob = vm0.trace_witness(0, 20, claim_globals=vm0.claims()[0].globals())
    assert np.array_equal(glob[:20], seals[0][1][:20])
    ocode = oc.witgen(20, 0)[0]
    assert np.array_equal(oc.prove(20, ocode, data, seals[0][1][:r0.TRACE_GLOBALS]), seals[0][1])
    # the reference's own envelope around it: a receipt file whose journal.bytes are the fixture's
    doc = json.loads(receipt.to_json())
    assert doc["journal"]["bytes"] == json.load(open(os.path.join(GOLDEN, "reference_receipt_6bb95807_latest.json")))["journal"]["bytes"]
    gc.free()


def test_the_library_builds_the_guests_input_words_from_the_raw_inputs():
    """r0h_camt53_guest_input (what `r0h_prove --camt53-response ...` uses): response.xml, the three public keys, the decrypted transaction
    key block and the witness signature give word for word the stream tools/guest_camt53.py builds in Python, in both commitment forms;
    keys of another size, a short block, a bad hex string are refused."""
    D = os.path.join(GOLDEN, "camt53")
    rd = lambda name: open(os.path.join(D, name), "rb").read()
    eb = r0.Ebics(rd("response.xml"))
    args = (rd("pub_bank.pem"), rd("pub_client.pem"), rd("pub_witness.pem"), rd("test.xml-TransactionKeyDecrypt.bin"), rd("test.xml-Witness.hex"))
    for form in (1, 0):
        got = eb.camt53_guest_input(*args, guest_camt53.REFERENCE_IBAN, guest_camt53.REFERENCE_HOST_INFO, form)
        assert got.tolist() == guest_camt53.elf_and_input(form=form)[1]
    with pytest.raises(r0.R0HipError, match="256 bytes"):
        eb.camt53_guest_input(args[0], args[1], args[2], args[3][:-1], args[4], "X", "h")
    with pytest.raises(r0.R0HipError, match="512 hex"):
        eb.camt53_guest_input(args[0], args[1], args[2], args[3], b"zz", "X", "h")
    with pytest.raises(r0.R0HipError, match="form"):
        eb.camt53_guest_input(*args, "X", "h", 2)
