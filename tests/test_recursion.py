"""lift + join tree (hyperfridge-r0_amd/recursion.py; SURVEY.md 8(a) a19, 8(e), BASELINE.json configs[4]): the schedule and
the point-to-point exchange are checked on CPU (in-process and with two gloo ranks); the proofs themselves on the GPU, against
the oracle, with a recursion-shaped circuit (risc0's own recursion circuit cannot be reproduced: see recursion.py)."""
import hashlib
import os
import queue
import threading

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from hyperfridge_r0_amd import recursion
from conftest import ROOT, circuit_path

P = 2013265921


def _blob(name):
    return np.fromfile(circuit_path(name), dtype=np.uint32)


@pytest.mark.parametrize("world", range(1, 10))
def test_tree_schedule_is_a_binary_join_tree(world):
    plan = recursion.tree_schedule(world)
    assert len(plan) == world - 1  # one join per hand-over
    senders = [s for _, _, s in plan]
    assert sorted(senders) == list(range(1, world))  # every rank but 0 hands its subtree up exactly once
    alive = set(range(world))
    for level, receiver, sender in plan:
        assert receiver < sender and sender - receiver == 1 << level and receiver in alive and sender in alive
        alive.discard(sender)
    assert alive == {0}
    assert max([lvl for lvl, _, _ in plan], default=-1) + 1 == (world - 1).bit_length()


class _HashNode:
    def __init__(self, seal):
        self.seal = np.ascontiguousarray(seal, dtype=np.uint32)

    def to_words(self):
        return self.seal


class _HashRecursor:
    """Stand-in prover for the transport tests: a 'seal' is 8 words of SHA-256 over its children (no claim travels with it)."""

    @staticmethod
    def leaf(i):
        return _HashNode(np.frombuffer(hashlib.sha256(b"leaf%d" % i).digest(), dtype=np.uint32))

    def join(self, a, b):
        return _HashNode(np.frombuffer(hashlib.sha256(a.seal.tobytes() + b.seal.tobytes()).digest(), dtype=np.uint32))

    def node_from_words(self, words):
        return _HashNode(words)


def _expected_root(world):
    rec, nodes = _HashRecursor(), {r: _HashRecursor.leaf(r) for r in range(world)}
    for _, receiver, sender in recursion.tree_schedule(world):
        nodes[receiver] = rec.join(nodes[receiver], nodes.pop(sender))
    return nodes[0].seal


@pytest.mark.parametrize("world", [1, 2, 5, 8])
def test_join_across_ranks_with_an_in_memory_transport(world):
    boxes = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
    results = [None] * world

    def run(rank):
        send = lambda words, dst: boxes[(rank, dst)].put(words.copy())
        recv = lambda src: boxes[(src, rank)].get(timeout=10)
        results[rank] = recursion.join_across_ranks(_HashRecursor(), _HashRecursor.leaf(rank), rank, world, send, recv)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(20) for t in threads]
    assert all(r is None for r in results[1:]) and np.array_equal(results[0].seal, _expected_root(world))


def _gloo_rank(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        send, recv = recursion.torch_transport()
        root = recursion.join_across_ranks(_HashRecursor(), _HashRecursor.leaf(rank), rank, world, send, recv)
        if rank == 0:
            out.put(root.seal.tolist())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_exchange_seals_point_to_point():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = [ctx.Process(target=_gloo_rank, args=(r, 2, port, out)) for r in range(2)]
    [p.start() for p in procs]
    got = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got == _expected_root(2).tolist()


def _sponge_columns(blob):
    """(first CODE column, first DATA column, first public input) of the in-circuit sponge: blob section SPONGE (12)"""
    at = 3
    for _ in range(int(blob[2])):
        tag, size = int(blob[at]), int(blob[at + 1])
        if tag == 12:
            return tuple(int(x) for x in blob[at + 2:at + 5])
        at += 2 + size
    raise AssertionError("no SPONGE section")


def _node_witness(c, orc, blob, po2, seed, claim_words, consumed, digest=None):
    """the oracle's witness of a recursion node that consumed `consumed`: public inputs planted, the sponge's rows over the words"""
    public = np.concatenate([claim_words, orc.hash_elem_slice(consumed) if digest is None else digest]).astype(np.uint32)
    code, data, glob = c.witgen(po2, seed, globals_in=public)
    _, first, _ = _sponge_columns(blob)
    n = 1 << po2
    data = data.reshape(-1, n).copy()
    data[first:first + r0.SPONGE_DATA_COLUMNS] = orc.sponge_trace(consumed, po2)
    return code, data.reshape(-1), glob


def test_oracle_public_inputs_are_planted_and_committed(orc):
    """orc_witgen_public + the sponge's rows: the witness satisfies the circuit, and the seal opens with exactly those 16 words."""
    blob = _blob("recursion")
    c = orc.circuit(blob)
    claim = (np.arange(8, dtype=np.uint64) * 123456789 % P).astype(np.uint32)
    consumed = (np.arange(100, dtype=np.uint64) * 987654321 % P).astype(np.uint32)
    code, data, glob = _node_witness(c, orc, blob, 10, 5, claim, consumed)
    assert np.array_equal(glob[:8], claim) and np.array_equal(glob[8:], orc.hash_elem_slice(consumed))
    seal = c.prove(10, code, data, glob)
    assert c.verify(seal) == (0, "ok") and r0.verify_seal(blob, seal) == (0, "ok", 10)
    assert np.array_equal(seal[:16], glob)
    forged = seal.copy()
    forged[3] = (int(forged[3]) + 1) % P  # claim another digest: the transcript no longer matches
    assert c.verify(forged)[0] != 0 and r0.verify_seal(blob, forged)[0] != 0
    assert r0.seal_digest(seal).tolist() == orc.hash_elem_slice(seal % P).tolist()


def test_the_consumed_digest_is_computed_inside_the_proof(orc):
    """public inputs 8..15 of a recursion node are the output of in-circuit Poseidon2 rows over witness cells holding
    the consumed words.  The rows the library plants are the oracle's and the generator's own restatement's; a witness whose cells hold
    other words than the digest names -- or the right words hashed wrongly, or a sponge cut short -- has no satisfying trace: whatever
    the prover emits for it, both verifiers refuse."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sponge_component as sc
    blob = _blob("recursion")
    c = orc.circuit(blob)
    rng = np.random.default_rng(11)
    po2, n = 10, 1 << 10
    for count in (0, 1, 15, 16, 17, 100, 16 * 33):
        words = rng.integers(0, P, count).astype(np.uint32)
        rows = r0.sponge_trace(words, po2)
        assert np.array_equal(rows, orc.sponge_trace(words, po2)), count
        assert np.array_equal(rows[:8, 30 * max(1, -(-count // 16)) - 1], r0.seal_digest(words)), count  # the state after the last round
    # the generator's restatement (canonical integers) gives the same rows
    words = rng.integers(0, P, 40).astype(np.uint32)
    cols, digest = sc.witness([orc.dec(int(w)) for w in words], n)
    assert np.array_equal(np.array([[orc.enc(v) for v in col] for col in cols], dtype=np.uint32), r0.sponge_trace(words, po2))
    assert [orc.enc(v) for v in digest] == r0.seal_digest(words).tolist()
    with pytest.raises(r0.R0HipError, match="rows"):
        r0.sponge_trace(np.zeros(16 * 35, dtype=np.uint32), po2)  # 35 permutations of 30 rows do not fit 2^10
    with pytest.raises(r0.R0HipError, match="canonical"):
        r0.sponge_trace(np.array([P], dtype=np.uint32), po2)

    claim = rng.integers(0, P, 8).astype(np.uint32)
    consumed = rng.integers(0, P, 16 * 6 + 3).astype(np.uint32)
    _, first, _ = _sponge_columns(blob)
    code, data, glob = _node_witness(c, orc, blob, po2, 9, claim, consumed)
    assert c.verify(c.prove(po2, code, data, glob)) == (0, "ok")

    def refused(data, glob=glob):
        try:
            seal = c.prove(po2, code, np.ascontiguousarray(data).reshape(-1), glob)
        except AssertionError:  # the oracle prover found the quotient is no polynomial
            return True
        return c.verify(seal)[0] != 0 and r0.verify_seal(blob, seal)[0] != 0

    # other words in the cells than the public digest names
    other = consumed.copy()
    other[40] = (int(other[40]) + 1) % P
    assert refused(_node_witness(c, orc, blob, po2, 9, claim, other, digest=glob[8:])[1])
    table = data.reshape(-1, n)
    # the right words, one state cell of one round off
    wrong = table.copy()
    wrong[first + 3, 47] = (int(wrong[first + 3, 47]) + 1) % P
    assert refused(wrong)
    # a cube cell that is not the cube
    wrong = table.copy()
    wrong[first + 24, 8] = (int(wrong[first + 24, 8]) + 1) % P
    assert refused(wrong)
    # the sponge stopped one permutation early, its digest claimed: the cells then hold a prefix of the words
    early = r0.seal_digest(consumed[:16 * 6])
    assert not refused(_node_witness(c, orc, blob, po2, 9, claim, consumed[:16 * 6])[1], np.concatenate([claim, early]))  # (honest: that digest, those words)
    assert refused(_node_witness(c, orc, blob, po2, 9, claim, consumed[:16 * 6], digest=glob[8:])[1])
    # `act` falling in the middle of a permutation, or never raised
    wrong = table.copy()
    wrong[first + 64, 100:] = 0
    assert refused(wrong)
    wrong = table.copy()
    wrong[first:first + 65] = 0
    assert refused(wrong)


def test_contiguous_sharding_keeps_rank_order_equal_to_segment_order():
    from hyperfridge_r0_amd import driver
    for total in (0, 1, 3, 8, 13, 64):
        for world in (1, 2, 3, 6, 8):
            runs = [driver.shard_contiguous(total, world, r) for r in range(world)]
            assert [s for run in runs for s in run] == list(range(total))
            assert max(len(r) for r in runs) - min(len(r) for r in runs) <= 1


@pytest.mark.gpu
def test_lift_and_join_on_the_device(hal, orc):
    """r0h_lift / r0h_join through the C ABI: nodes carry composed claims, their seals name them, the oracle reproduces a node's seal
    word for word, two nodes that do not follow one another are refused."""
    seg_blob, rec_blob = _blob("small"), _blob("recursion")
    seg = hal.load_circuit(seg_blob)
    journal = r0.serde_encode_str('{"n":4}')
    claims, image_id = r0.session_claims(4, journal)
    seg_cc = hal.code_commit(seg, 10)
    seals = []
    for k, cl in enumerate(claims):
        code, data, glob = hal.witgen(seg, 10, 1 + k, globals_in=cl.globals())
        seals.append(hal.prove_segment(seg, 10, seg_cc, data, glob))
        code.free(); data.free()
    size = 16  # a 2^10-row `small` seal is ~21k words: 1.3k permutations of 30 rows
    import __graft_entry__ as entry
    rec = recursion.Recursor(hal, rec_blob, seg_blob, entry.code_object_path("recursion"), po2=size, segment_roots={10: seg_cc.root()})
    lifted = [rec.lift(s, cl) for s, cl in zip(seals, claims)]
    oc = orc.circuit(rec_blob)
    for node, s, cl in zip(lifted, seals, claims):
        assert np.array_equal(node.claim_words, cl.globals()) and np.array_equal(node.consumed_digest, r0.seal_digest(s))
        assert node.claim.digest() == cl.digest() and rec.verify(node)
        assert r0.verify_seal(rec_blob, node.seal, code_root=rec.control_root) == (0, "ok", size) and oc.verify(node.seal, code_root=rec.control_root) == (0, "ok")
    # bit-exact against the oracle proving the same step: the public inputs planted, the sponge's rows over the segment seal's words
    public = np.concatenate([lifted[0].claim_words, lifted[0].consumed_digest])
    seed = int(public[0]) | (int(public[8]) << 32)
    ocode, odata, oglob = _node_witness(oc, orc, rec_blob, size, seed, lifted[0].claim_words, seals[0])
    assert np.array_equal(oglob, public) and np.array_equal(oc.prove(size, ocode, odata, oglob), lifted[0].seal)
    # a node whose cells hold another seal than its public digest names: the device proves what it is given, and nobody accepts the result
    rc_circuit = hal.load_circuit(rec_blob, entry.code_object_path("recursion"))
    dcode, ddata, dglob = hal.witgen(rc_circuit, size, seed, globals_in=public)
    _, first, _ = _sponge_columns(rec_blob)
    ddata.upload(r0.sponge_trace(seals[1], size).reshape(-1), offset_words=first << size)
    try:
        stray = hal.prove_segment(rc_circuit, size, dcode, ddata, dglob)
        assert r0.verify_seal(rec_blob, stray)[0] != 0 and oc.verify(stray)[0] != 0
    except r0.R0HipError:
        pass
    ddata.upload(r0.sponge_trace(seals[0], size).reshape(-1), offset_words=first << size)  # (the right rows: the same call succeeds)
    assert np.array_equal(hal.prove_segment(rc_circuit, size, dcode, ddata, dglob), lifted[0].seal)
    dcode.free(); ddata.free(); rc_circuit.free()
    root = rec.fold(lifted)  # join(join(l0, l1), join(l2, l3))
    assert rec.verify(root) and oc.verify(root.seal, code_root=rec.control_root) == (0, "ok")
    end_to_end = r0.ReceiptClaim.make(claims[0].pre, claims[-1].post, claims[-1].exit_system, claims[-1].exit_user, bytes(claims[-1].output_digest))
    assert root.claim.digest() == end_to_end.digest() and root.claim.pre.digest() == image_id and np.array_equal(root.claim_words, end_to_end.globals())
    left, right = rec.join(lifted[0], lifted[1]), rec.join(lifted[2], lifted[3])
    assert np.array_equal(root.consumed_digest, r0.seal_digest(np.concatenate([r0.seal_digest(left.seal), r0.seal_digest(right.seal)])))
    assert left.claim.pre.digest() == claims[0].pre.digest() and left.claim.post.digest() == claims[1].post.digest() and left.claim.exit_system == 2
    # a node travels as claim + seal and arrives the same
    back = recursion.Node.from_words(left.to_words())
    assert np.array_equal(back.seal, left.seal) and back.claim.digest() == left.claim.digest() and rec.verify(back)
    # refused: a swapped pair, a gap, something after a halt, a forged claim beside a genuine seal, a broken seal
    with pytest.raises(r0.R0HipError, match="do not follow one another"):
        rec.join(lifted[1], lifted[0])
    with pytest.raises(r0.R0HipError, match="do not follow one another"):
        rec.join(lifted[0], lifted[2])
    with pytest.raises(r0.R0HipError, match="SystemSplit"):
        rec.join(lifted[3], lifted[0])
    forged = recursion.Node.from_parts(lifted[1].seal, claims[2])
    with pytest.raises(r0.R0HipError, match="does not name the claim"):
        rec.join(lifted[0], recursion.Node.from_parts(lifted[1].seal, r0.ReceiptClaim.make(claims[1].pre, claims[2].post, 2)))
    assert not rec.verify(forged)
    with pytest.raises(r0.R0HipError, match="do not name this claim"):
        rec.lift(seals[0], claims[1])
    bad = seals[0].copy()
    bad[-1] ^= 1
    with pytest.raises(r0.R0HipError, match="does not verify"):
        rec.lift(bad, claims[0])
    tampered = lifted[1].seal.copy()
    tampered[-1] ^= 1
    with pytest.raises(r0.R0HipError, match="does not verify"):
        rec.join(lifted[0], recursion.Node.from_parts(tampered, claims[1]))
    rec.close()
    seg_cc.free(); seg.free()


@pytest.mark.gpu
def test_the_segments_of_a_proved_run_join_into_one_root_with_the_receipts_claim(hal, orc):
    """the segment receipts of `prove(env, elf)` with the trace circuit -- the run the device proved from its own
    execution -- are lifted and joined into one root whose claim is the composite receipt's end-to-end claim."""
    import __graft_entry__ as entry
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_session import elf_of
    from test_rv32im import _guest
    blob, rec_blob = _blob("trace"), _blob("recursion")
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    receipt, image_id, cycles = hal.prove_elf(gc, elf_of(_guest(2000), 0x400), [7, 0x01020304], segment_po2=11)
    seals, claims = [s for _, s in receipt.seals()], receipt.claims()
    assert len(seals) >= 8
    seals, claims = seals[-8:], claims[-8:]  # the last eight segments: the one that halts carries the journal's digest; segments of closing rows follow it
    halting = max(k for k, cl in enumerate(claims) if cl.exit_system == 0)
    assert halting < 7  # (2^11-row segments: the rows that close the session do not fit beside the last cycles)
    roots = {}
    for s in seals:
        size = r0.verify_seal(blob, s)[2]
        if size not in roots:
            cc = hal.code_commit(gc, size)
            roots[size] = cc.root()
            cc.free()
    rec = recursion.Recursor(hal, rec_blob, blob, entry.code_object_path("recursion"), po2=17, segment_roots=roots)  # 2^16-row trace seals: ~50k words
    root = rec.fold([rec.lift(s, cl) for s, cl in zip(seals, claims)])
    want = r0.ReceiptClaim.make(claims[0].pre, claims[-1].post, claims[halting].exit_system, claims[halting].exit_user, bytes(claims[halting].output_digest))
    assert rec.verify(root) and root.claim.digest() == want.digest() and bytes(root.claim.output_digest) == r0.output_digest(receipt.journal) and root.claim.exit_system == 0
    assert orc.circuit(rec_blob).verify(root.seal, code_root=rec.control_root) == (0, "ok")
    lifted = [rec.lift(seals[0], claims[0]), rec.lift(seals[1], claims[1])]
    with pytest.raises(r0.R0HipError, match="do not follow one another"):
        rec.join(lifted[1], lifted[0])
    # a segment proved for a claim that says something else than its run (round 3's advisor): the seal is valid and NAMES that claim, but
    # its own public inputs -- first / last pc, the way it ends, the exit code -- are the run's: lift refuses what r0h_receipt_verify would
    vm = r0.Vm()
    vm.load_elf(elf_of(_guest(2000), 0x400))
    vm.set_input([7, 0x01020304])
    assert vm.run(segment_po2=11, keep_trace=True, boundary_rows=True) == (0, 0)
    k = max(i for i, sg in enumerate(vm.segments()) if sg.user_cycles)  # the segment that halts
    seg, honest = vm.segments()[k], vm.claims()[k]
    size = r0.TRACE_MIN_PO2
    code_cols, synthetic, _ = hal.witgen(gc, size, 0)
    synthetic.free()
    cc = hal.code_commit(gc, size, code_cols)
    rows, bounds = vm.preflight_arrays(k)
    for forged in (r0.ReceiptClaim.make(honest.pre, r0.SystemState.make(honest.post.pc + 4, bytes(honest.post.merkle_root)), 0, 0, bytes(honest.output_digest)),
                   r0.ReceiptClaim.make(honest.pre, honest.post, 0, 7, bytes(honest.output_digest)),
                   r0.ReceiptClaim.make(honest.pre, honest.post, 2, 0, None)):
        dev, glob = hal.trace_witgen(rows, bounds, size, claim_globals=forged.globals(), number=k + 1, closing=bool(seg.closing), idle_pc=seg.pre.pc, circuit=gc)
        glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = seals[0][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]
        seal = hal.prove_segment(gc, size, cc, dev, hal.logup_totals(gc, size, code_cols, dev, glob))
        dev.free()
        assert r0.verify_seal(blob, seal, code_root=roots[size])[:2] == (0, "ok") and np.array_equal(seal[:8], forged.globals())
        with pytest.raises(r0.R0HipError, match="first / last pc, way of ending or exit code"):
            rec.lift(seal, forged)
    cc.free(); code_cols.free()
    rec.close()
    gc.free()


def _run_tree(world, leaf_of, recursor_of):
    boxes = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
    results, errors = [None] * world, [None] * world

    def run(rank):
        send = lambda words, dst: boxes[(rank, dst)].put(np.array(words, dtype=np.uint32))
        recv = lambda src: boxes[(src, rank)].get(timeout=10)
        try:
            results[rank] = recursion.join_across_ranks(recursor_of(rank), leaf_of(rank), rank, world, send, recv)
        except Exception as exc:  # noqa: BLE001
            errors[rank] = exc

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(30) for t in threads]
    assert not any(t.is_alive() for t in threads), "a rank is still blocked in the tree"
    return results, errors


def test_ranks_without_a_segment_hand_up_an_empty_message():
    """segments < world: ranks 3.. own nothing; the tree still completes and the root covers the three leaves that exist."""
    world = 6
    results, errors = _run_tree(world, lambda r: _HashRecursor.leaf(r) if r < 3 else None, lambda r: _HashRecursor())
    assert errors == [None] * world
    rec = _HashRecursor()
    want = rec.join(rec.join(_HashRecursor.leaf(0), _HashRecursor.leaf(1)), _HashRecursor.leaf(2))
    assert np.array_equal(results[0].seal, want.seal) and all(r is None for r in results[1:])


def test_a_failed_join_reaches_the_root_and_no_rank_stays_blocked():
    class Failing(_HashRecursor):
        def join(self, a, b):
            raise r0.R0HipError("join (left): the seal to be consumed does not verify")

    world = 8
    results, errors = _run_tree(world, _HashRecursor.leaf, lambda r: Failing() if r == 2 else _HashRecursor())
    assert "does not verify" in str(errors[2])                       # where it happened
    assert "reported a failure" in str(errors[0])                    # carried up to the root
    assert results[0] is None and errors[1] is None and errors[3] is None  # uninvolved ranks handed up and left
