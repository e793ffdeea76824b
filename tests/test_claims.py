"""Receipt claims and `receipt.verify(image_id)` (csrc/claim.cpp; verifier/src/main.rs:118-126).

Pinned: SHA-256 by the FIPS 180-4 example vectors and hashlib.  Recalled from the public risc0 sources, NOT pinned by anything
the reference holds (its receipts are `"inner":"Fake"`): the tagged-struct layout, the tags, the field names of the composite
receipt JSON.  The tests restate the recalled definitions independently with hashlib and check the library against them, and
drive the receipt check with seals made by the CPU oracle (no GPU needed)."""
import hashlib
import json
import struct

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import circuit_path

P = 2013265921


def test_sha256_fips_180_4_vectors():
    vec = {
        b"abc": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad",
        b"": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855",
        b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq": "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1",
        b"abcdefghbcdefghicdefghijdefghijkefghijklfghijklmghijklmnhijklmnoijklmnopjklmnopqklmnopqrlmnopqrsmnopqrstnopqrstu":
            "cf5b16a778af8380036ce59e7b0492370b249b11e8f07a51afac45037afee9d1",
        b"a" * 1000000: "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0",
    }
    for msg, want in vec.items():
        assert r0.sha256(msg).hex() == want
    rng = np.random.default_rng(7)
    for n in list(range(0, 130)) + [255, 256, 257, 4095, 70001]:  # every padding case around the 55/56/64-byte boundaries
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert r0.sha256(m) == hashlib.sha256(m).digest(), n


def _tagged(tag, down, data):
    h = hashlib.sha256(hashlib.sha256(tag.encode()).digest())
    for d in down:
        h.update(d)
    for w in data:
        h.update(struct.pack("<I", w))
    h.update(struct.pack("<H", len(down)))
    return h.digest()


def test_tagged_struct_and_claim_digests_follow_the_recalled_risc0_layout():
    a, b = hashlib.sha256(b"a").digest(), hashlib.sha256(b"b").digest()
    assert r0.tagged_struct("risc0.Test", [a, b], [1, 2013265920, 0xFFFFFFFF]) == _tagged("risc0.Test", [a, b], [1, 2013265920, 0xFFFFFFFF])
    assert r0.tagged_struct("risc0.Empty", [], []) == _tagged("risc0.Empty", [], [])
    pre, post = r0.SystemState.make(0x200800, a), r0.SystemState.make(0, b)
    assert pre.digest() == _tagged("risc0.SystemState", [a], [0x200800])
    journal = b"\x04\x00\x00\x00abcd"
    out = r0.output_digest(journal)
    assert out == _tagged("risc0.Output", [hashlib.sha256(journal).digest(), bytes(32)], [])
    for (sys_, user), outd in (((0, 0), out), ((1, 7), out), ((2, 0), None), ((2, 2), None)):
        c = r0.ReceiptClaim.make(pre, post, sys_, user, outd)
        want = _tagged("risc0.ReceiptClaim", [bytes(32), pre.digest(), post.digest(), outd or bytes(32)], [sys_, user])
        assert c.digest() == want
        g = c.globals()
        assert g.dtype == np.uint32 and g.size == 8 and (g < P).all()
    # different claims get different naming words; the words are the Poseidon2 sponge over the digest's sixteen 16-bit halves
    c1, c2 = r0.ReceiptClaim.make(pre, post, 0, 0, out), r0.ReceiptClaim.make(pre, post, 0, 1, out)
    assert not np.array_equal(c1.globals(), c2.globals())


def test_claim_globals_match_the_oracle_sponge(orc):
    d = hashlib.sha256(b"claim").digest()
    halves = [orc.enc(d[2 * i] | (d[2 * i + 1] << 8)) for i in range(16)]
    assert r0.claim_globals(d).tolist() == orc.hash_elem_slice(np.array(halves, np.uint32)).tolist()


@pytest.fixture(scope="module")
def session(orc):
    """A three-segment session proved by the CPU oracle over the `small` circuit (po2 = 9), claims planted as public inputs."""
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    c = orc.circuit(blob)
    assert c.n_global == 8
    po2 = 9
    journal = r0.serde_encode_str(json.dumps({"iban": "CH4308307000289537312", "stmts": [{"elctrnc_seq_nb": "247"}]}, separators=(",", ":")))
    claims, image_id = r0.session_claims(3, journal)
    seals, root = [], None
    for k, cl in enumerate(claims):
        code, data, glob = c.witgen(po2, 100 + k, globals_in=cl.globals())
        root = c.code_root(code, po2)
        seal = c.prove(po2, code, data, glob)
        assert seal.size and np.array_equal(seal[:8], cl.globals())
        seals.append(seal)
    return dict(blob=blob, c=c, po2=po2, journal=journal, claims=claims, image_id=image_id, seals=seals, roots={po2: root})


def test_receipt_verifies_against_the_image_id_and_round_trips_through_json(session):
    s = session
    rc = r0.Receipt.new(s["journal"], s["seals"], s["claims"])
    assert rc.verify(s["blob"], s["roots"], s["image_id"])[:2] == (0, "ok")
    text = rc.to_json()
    doc = json.loads(text)
    segs = doc["inner"]["Composite"]["segments"]
    assert [g["index"] for g in segs] == [0, 1, 2] and all(g["hashfn"] == "poseidon2" and len(g["verifier_parameters"]) == 64 for g in segs)
    assert segs[0]["claim"]["exit_code"] == "SystemSplit" and segs[0]["claim"]["output"] == {"Value": None}
    assert segs[2]["claim"]["exit_code"] == {"Halted": 0} and segs[2]["claim"]["output"] == {"Pruned": r0.output_digest(s["journal"]).hex()}
    assert segs[1]["claim"]["pre"]["Value"]["merkle_root"] == segs[0]["claim"]["post"]["Value"]["merkle_root"]
    assert doc["inner"]["Composite"]["assumption_receipts"] == [] and "verifier_parameters" in doc["metadata"]
    assert bytes(doc["journal"]["bytes"]) == s["journal"]
    back = r0.Receipt.parse(text)
    assert back.to_json() == text and back.verify(s["blob"], s["roots"], s["image_id"])[0] == 0
    assert [c.digest() for c in back.claims()] == [c.digest() for c in s["claims"]]
    # an unpruned output ({"Value":{"journal":{"Value":[..]},"assumptions":{"Value":[]}}}) folds to the same digest
    segs[2]["claim"]["output"] = {"Value": {"journal": {"Value": list(s["journal"])}, "assumptions": {"Value": []}}}
    assert r0.Receipt.parse(json.dumps(doc)).verify(s["blob"], s["roots"], s["image_id"])[0] == 0


def test_receipt_verification_rejects_what_the_reference_verifier_exists_to_reject(session):
    s = session
    blob, roots, img = s["blob"], s["roots"], s["image_id"]
    ok = lambda rc, **kw: rc.verify(blob, kw.get("roots", roots), kw.get("image_id", img))
    # a rewritten journal with perfectly valid seals
    other = r0.serde_encode_str('{"iban":"CH0000000000000000000","stmts":[]}')
    assert ok(r0.Receipt.new(other, s["seals"], s["claims"]))[:2] == (7, "the journal is not the one the last segment's claim commits to")
    # another program's image id
    assert ok(r0.Receipt.new(s["journal"], s["seals"], s["claims"]), image_id=hashlib.sha256(b"other").digest())[0] == 8
    # segments in another order (indices renumbered): the seals no longer name the claims they sit next to ... unless the claims move
    # with them, and then the states do not chain
    assert ok(r0.Receipt.new(s["journal"], [s["seals"][1], s["seals"][0], s["seals"][2]], s["claims"]))[0] == 5
    moved = r0.Receipt.new(s["journal"], [s["seals"][1], s["seals"][0], s["seals"][2]], [s["claims"][1], s["claims"][0], s["claims"][2]])
    assert ok(moved)[0] == 6
    # a segment dropped from the middle
    assert ok(r0.Receipt.new(s["journal"], [s["seals"][0], s["seals"][2]], [s["claims"][0], s["claims"][2]]))[0] == 6
    # a claim edited after proving (another post state)
    forged = [s["claims"][0], s["claims"][1], r0.ReceiptClaim.make(s["claims"][2].pre, r0.SystemState.make(4, bytes(32)), 0, 0, bytes(s["claims"][2].output_digest))]
    assert ok(r0.Receipt.new(s["journal"], s["seals"], forged))[:3] == (5, "a seal's public inputs do not name its claim", 2)
    # a last segment that did not halt
    # (the claim is part of what the seal names, so the seal has to be re-proved for it: use the session's first two segments only)
    assert ok(r0.Receipt.new(s["journal"], s["seals"][:2], s["claims"][:2]))[0] == 9
    # a flipped seal word, an unknown trace size, a receipt without claims, a Fake receipt
    bad = [x.copy() for x in s["seals"]]
    bad[1][-1] ^= 1
    v = ok(r0.Receipt.new(s["journal"], bad, s["claims"]))
    assert v[0] == 2 and v[2] == 1 and v[3] != 0
    assert ok(r0.Receipt.new(s["journal"], s["seals"], s["claims"]), roots={12: roots[s["po2"]]})[0] == 3
    assert ok(r0.Receipt.new(s["journal"], s["seals"]))[0] == 4
    assert ok(r0.Receipt.new(s["journal"]))[0] == 1
    # seals over foreign CODE columns: valid proofs, wrong program
    c = s["c"]
    code, data, glob = c.witgen(s["po2"], 100, globals_in=s["claims"][0].globals(), code_seed=0xBAD)
    foreign = c.prove(s["po2"], code, data, glob)
    v = ok(r0.Receipt.new(s["journal"], [foreign] + s["seals"][1:], s["claims"]))
    assert v[0] == 2 and v[2] == 0 and v[3] == 10
    # a segment that claims another hash suite
    other_suite = json.loads(r0.Receipt.new(s["journal"], s["seals"], s["claims"]).to_json())
    other_suite["inner"]["Composite"]["segments"][1]["hashfn"] = "sha-256"
    assert r0.Receipt.parse(json.dumps(other_suite)).verify(blob, roots, img)[:3] == (11, "a segment names a hash function other than poseidon2", 1)
    # a circuit that cannot name a claim
    tiny = np.fromfile(circuit_path("tiny"), dtype=np.uint32)
    assert r0.Receipt.new(s["journal"], s["seals"], s["claims"]).verify(tiny, roots, img)[0] == 10
    # a claim missing in the MIDDLE is reported as missing at that segment, before any claim is looked at (the chain check
    # used to read it first)
    holed = json.loads(r0.Receipt.new(s["journal"], s["seals"], s["claims"]).to_json())
    holed["inner"]["Composite"]["segments"][1]["claim"] = None
    assert r0.Receipt.parse(json.dumps(holed)).verify(blob, roots, img)[:3] == (4, "a segment carries no claim", 1)
    # no image id, no acceptance: `receipt.verify(image_id)` always names the program
    v = r0.Receipt.new(s["journal"], s["seals"], s["claims"]).verify(blob, roots, None)
    assert v[0] == 12 and "no image id" in v[1]


def test_image_id_text_follows_the_reference_fixture():
    """host/out/IMAGE_ID.hex (reference-held data): eight `{:08x}` u32 words (host/src/main.rs:445-449), read back word by word with
    `u32::from_str_radix` (verifier/src/main.rs:131-143) into `Digest::from([u32; 8])`, which keeps each word little-endian -- so every
    4-byte group of the digest is the reverse of its 8 digits."""
    import os
    text = open(os.path.join(os.path.dirname(__file__), "golden", "reference_IMAGE_ID.hex")).read().strip()
    assert len(text) == 64
    words = [int(text[8 * i:8 * i + 8], 16) for i in range(8)]
    want = b"".join(w.to_bytes(4, "little") for w in words)
    got = r0.image_id_from_hex(text)
    assert got == want and got[:4] == bytes.fromhex(text[:8])[::-1] and r0.image_id_to_hex(got) == text
    for bad in (text[:-1], text + "0", " " + text[1:], "-" + text[1:], text[:10] + "g" + text[11:]):  # strict: no signs, blanks or short reads
        with pytest.raises(r0.R0HipError, match="hex"):
            r0.image_id_from_hex(bad)


def test_the_receipts_of_several_ranks_merge_into_the_sessions_receipt():
    """r0h_receipt_merge: each rank of a multi-GPU session proves segments rank, rank + world, ... (r0h_prove_elf_part) and holds a
    composite receipt of just those; merged they are the receipt one prover would have made, whatever the order they arrive in."""
    states = [r0.SystemState.make(0x1000 + 4 * i, bytes([i + 1]) * 32) for i in range(8)]
    claims = [r0.ReceiptClaim.make(states[i], states[i + 1], 2 if i < 6 else 0, 0, None) for i in range(7)]
    seals = [np.arange(40, dtype=np.uint32) * (i + 3) for i in range(7)]
    journal = r0.serde_encode_str('{"iban":"X"}')
    whole = r0.Receipt.new(journal, seals, claims)
    for world in (1, 2, 3, 7):
        shares = [r0.Receipt.new(journal, seals[r::world], claims[r::world], indices=list(range(r, 7, world))) for r in range(world)]
        assert r0.Receipt.merge(shares).to_json() == whole.to_json() and r0.Receipt.merge(shares[::-1]).to_json() == whole.to_json(), world
    a = r0.Receipt.new(journal, seals[0::2], claims[0::2], indices=[0, 2, 4, 6])
    b = r0.Receipt.new(journal, seals[1::2], claims[1::2], indices=[1, 3, 5])
    with pytest.raises(r0.R0HipError, match="one is missing"):
        r0.Receipt.merge([a, r0.Receipt.new(journal, seals[1:4:2], claims[1:4:2], indices=[1, 3])])
    with pytest.raises(r0.R0HipError, match="there twice"):
        r0.Receipt.merge([a, b, r0.Receipt.new(journal, seals[1:2], claims[1:2], indices=[1])])
    with pytest.raises(r0.R0HipError, match="another journal"):
        r0.Receipt.merge([a, r0.Receipt.new(journal + b"\0\0\0\0", seals[1::2], claims[1::2], indices=[1, 3, 5])])
    with pytest.raises(r0.R0HipError, match="not a composite"):
        r0.Receipt.merge([a, r0.Receipt.new(journal)])
    # the session's image proof travels with whichever part carries it (rank 0 makes it), through JSON too; two different ones are refused
    proof = np.arange(100, dtype=np.uint32) * 7
    b.image_proof = proof
    merged = r0.Receipt.merge([a, b])
    assert np.array_equal(merged.image_proof, proof) and np.array_equal(r0.Receipt.parse(merged.to_json()).image_proof, proof)
    assert r0.Receipt.merge([a, r0.Receipt.new(journal, seals[1::2], claims[1::2], indices=[1, 3, 5])]).image_proof is None
    a.image_proof = proof + 1
    with pytest.raises(r0.R0HipError, match="another image proof"):
        r0.Receipt.merge([a, b])


@pytest.mark.parametrize("name", ["tiny", "small", "recursion", "trace", "bench", "image"])
def test_the_verifier_can_derive_a_circuits_control_roots_itself(orc, name):
    """r0h_control_root_host: the CODE columns of the circuit's column program committed on the host -- interpolate, shift by 3,
    evaluate on the 4N coset, hash rows, fold -- give the control root the oracle computes from the CODE group it generates (and, on
    the GPU, the root of r0h_code_commit: tests/test_gpu_code_commit.py).  A verifier with the blob needs no table of roots."""
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    c = orc.circuit(blob)
    for po2 in (9, 10, 13):
        assert np.array_equal(r0.control_root_host(blob, po2), c.code_root(c.witgen(po2, 0)[0], po2)), po2
    assert not np.array_equal(r0.control_root_host(blob, 9), r0.control_root_host(blob, 10))
    with pytest.raises(r0.R0HipError, match="po2"):
        r0.control_root_host(blob, 3)
