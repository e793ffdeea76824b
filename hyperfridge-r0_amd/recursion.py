"""lift + join tree over segment seals: the exchange-and-prove schedule of BASELINE.json configs[4] (SURVEY.md 8(a) a19,
8(e)): risc0-zkvm `ProverServer::lift(SegmentReceipt)` turns every segment seal into a recursion-circuit proof, `join(a, b)`
folds two of them into one, and the folds form a binary tree whose levels are the only inter-GPU exchange of the whole path.

What is reproduced here is that dataflow, with the same kernels: every lift and join is one STARK over a second,
recursion-SHAPED circuit (circuits/recursion.r0c, proved at po2 = 18) whose 16 public inputs are the two 8-word digests the
step stands for, after the step has checked the seals it consumes with the host-side verifier.  What is NOT reproduced is
risc0's recursion circuit itself (its programs are downloaded at build time upstream): the check of the children happens
beside the proof, not inside it, so the root is a verifiable tree of seals, not a succinct receipt.  Parity for this row is
therefore "the seals verify (product verifier and oracle) and bind the digests", not word parity with risc0.

Transport: `torch.distributed` point-to-point send/recv (backend "nccl" = RCCL over xGMI on a GPU node, "gloo" on CPU), one
process per GPU; partner of rank r at level l is r ^ (1 << l), the lower rank of a pair joins.  No collective is involved.
"""
import numpy as np

from . import Hal, seal_digest, verify_seal, R0HipError  # noqa: F401  (package __init__ is the ctypes harness)

RECURSION_PO2 = 18


def tree_schedule(world):
    """[(level, receiver, sender), ...] of the binary join tree over ranks 0..world-1 (any world size >= 1)."""
    plan, stride = [], 1
    while stride < world:
        for r in range(0, world, 2 * stride):
            if r + stride < world:
                plan.append((stride.bit_length() - 1, r, r + stride))
        stride *= 2
    return plan


class Node:
    """A proven tree node: its seal, the circuit it is a proof of, and the digests its public inputs carry."""

    def __init__(self, seal, left, right):
        self.seal, self.left, self.right = np.ascontiguousarray(seal, dtype=np.uint32), left, right

    @property
    def digest(self):
        return seal_digest(self.seal)


class Recursor:
    """Proves lift / join steps on one device.  `segment_blob` is the circuit the leaf seals belong to."""

    def __init__(self, hal, recursion_blob, segment_blob, code_object=None, po2=RECURSION_PO2):
        self.hal, self.po2 = hal, po2
        self.recursion_blob = np.ascontiguousarray(recursion_blob, dtype=np.uint32)
        self.segment_blob = np.ascontiguousarray(segment_blob, dtype=np.uint32)
        self.circuit = hal.load_circuit(self.recursion_blob, code_object)
        if self.circuit.n_global != 16:
            raise R0HipError("recursion circuit must expose 16 public inputs (two digests), this one has %d" % self.circuit.n_global)

    def _prove(self, left_digest, right_digest):
        public = np.concatenate([left_digest, right_digest]).astype(np.uint32)
        seed = int(public[0]) | (int(public[8]) << 32)  # the rest of the witness is synthetic: any deterministic choice
        code, data, glob = self.hal.witgen(self.circuit, self.po2, seed, globals_in=public)
        try:
            return self.hal.prove_segment(self.circuit, self.po2, code, data, glob)
        finally:
            code.free()
            data.free()

    def _checked(self, checks, prove):
        """Run `prove()` on the device while host threads verify the seals this step consumes (checks = [(seal, blob, what)]);
        the step only counts if every check passed.  The proof needs the digests, not the verdicts, so the two overlap."""
        import threading
        verdicts = [None] * len(checks)

        def run(k):
            verdicts[k] = verify_seal(checks[k][1], checks[k][0])

        threads = [threading.Thread(target=run, args=(k,)) for k in range(len(checks))]
        for t in threads:
            t.start()
        try:
            seal = prove()
        finally:
            for t in threads:
                t.join()
        for (_, _, what), v in zip(checks, verdicts):
            if v is None or v[0] != 0:
                raise R0HipError("%s: the seal to be consumed does not verify: %s" % (what, v[1] if v else "verifier did not run"))
        return seal

    def lift(self, segment_seal):
        """risc0 `lift`: one recursion-circuit proof standing for one segment seal."""
        d = seal_digest(segment_seal)
        zero = np.zeros(8, dtype=np.uint32)
        seal = self._checked([(segment_seal, self.segment_blob, "lift")], lambda: self._prove(d, zero))
        return Node(seal, d, zero)

    def join(self, a, b):
        """risc0 `join`: one recursion-circuit proof standing for two recursion proofs (seals given as arrays or Nodes)."""
        sa, sb = (a.seal if isinstance(a, Node) else a), (b.seal if isinstance(b, Node) else b)
        da, db = seal_digest(sa), seal_digest(sb)
        seal = self._checked([(sa, self.recursion_blob, "join (left)"), (sb, self.recursion_blob, "join (right)")], lambda: self._prove(da, db))
        return Node(seal, da, db)

    def fold(self, nodes):
        """Left-to-right binary fold of this rank's own nodes (log depth)."""
        nodes = list(nodes)
        while len(nodes) > 1:
            nxt = [self.join(nodes[i], nodes[i + 1]) for i in range(0, len(nodes) - 1, 2)]
            if len(nodes) & 1:
                nxt.append(nodes[-1])
            nodes = nxt
        return nodes[0]

    def close(self):
        self.circuit.free()


def public_inputs_of(recursion_blob, seal):
    """The 16 public-input words a recursion seal opens with (its transcript commits to them): (left digest, right digest)."""
    seal = np.ascontiguousarray(seal, dtype=np.uint32)
    return seal[:8].copy(), seal[8:16].copy()


FAILED = np.array([0xFFFFFFFF], dtype=np.uint32)  # handed up instead of a seal by a rank whose subtree could not be proved
EMPTY = np.zeros(0, dtype=np.uint32)               # handed up by a rank that owns no segment (segments < world)


def join_across_ranks(recursor, node, rank, world, send, recv):
    """Run the cross-rank part of the tree.  `send(array, dst)` / `recv(src) -> array` move one seal (uint32 words).
    Returns the root Node on rank 0 and None elsewhere.

    Every rank takes part in every exchange the schedule gives it, whatever happened before: a rank without a node (empty
    share) hands up an empty message and its partner keeps what it has; a rank whose join failed (or whose child reported a
    failure) hands up a failure marker instead of blocking its partner, so the failure reaches the root, every rank leaves
    the tree, and the first error is re-raised on the rank where it happened (and as "a child rank failed" above it)."""
    failed = None
    for _level, receiver, sender in tree_schedule(world):
        if rank == sender:
            send(FAILED if failed is not None else (EMPTY if node is None else node.seal), receiver)
            if failed is not None:
                raise failed
            return None  # this rank's subtree has been handed up
        if rank == receiver:
            other = recv(sender)
            if other.size == 1 and other[0] == FAILED[0]:
                failed = failed or R0HipError("join tree: rank %d reported a failure in its subtree" % sender)
            elif failed is None and other.size:
                try:
                    node = Node(other, None, None) if node is None else recursor.join(node, Node(other, None, None))
                except Exception as exc:  # noqa: BLE001 -- carried up the tree, re-raised below
                    failed = exc
    if failed is not None:
        raise failed
    return node if rank == 0 else None


def torch_transport(device=None):
    """send/recv over torch.distributed point-to-point (RCCL when the process group is "nccl", gloo otherwise)."""
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    dev = device if on_gpu else "cpu"

    def send(words, dst):
        n = torch.tensor([int(words.size)], dtype=torch.int64, device=dev)
        dist.send(n, dst)
        if words.size:  # an EMPTY message is its length word alone: no zero-byte point-to-point transfer (both sides skip it)
            dist.send(torch.from_numpy(words.view(np.int32).copy()).to(dev), dst)

    def recv(src):
        n = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.recv(n, src)
        if int(n.item()) == 0:
            return np.zeros(0, dtype=np.uint32)
        buf = torch.empty(int(n.item()), dtype=torch.int32, device=dev)
        dist.recv(buf, src)
        return buf.cpu().numpy().view(np.uint32)

    return send, recv
