"""lift + join tree over segment receipts: the exchange schedule of BASELINE.json configs[4].

risc0-zkvm `ProverServer::lift(SegmentReceipt)` turns every segment seal into a recursion-circuit proof, `join(a, b)` folds two of
them into one, and the folds form a binary tree whose levels are the only inter-GPU exchange of the whole path.  The proving side --
lift, join, the composition of claims ({pre: a.pre, post: b.post, ..}), the refusal of two nodes that do not follow one another --
lives in the library behind the C ABI (csrc/recursion.cpp: r0h_recursor_new / r0h_lift / r0h_join / r0h_node_*; what a node's
proof is and is not -- the digest of what it consumed computed in-circuit, the children's seals checked BESIDE the proof, not risc0's
recursion circuit -- is stated there and in include/r0hip.h).
This module is the transport only: which rank hands its subtree to which, and how a node travels.

Transport: `torch.distributed` point-to-point send/recv (backend "nccl" = RCCL over xGMI on a GPU node, "gloo" on CPU), one
process per GPU; partner of rank r at level l is r ^ (1 << l), the lower rank of a pair joins.  No collective is involved.
A node travels as its claim (the 144 bytes of r0h_receipt_claim) followed by its seal.
"""
import ctypes

import numpy as np

from . import Hal, ReceiptClaim, R0HipError, _check, _u32arr, _vp, _sz, lib, seal_digest, verify_seal  # noqa: F401  (package __init__ is the ctypes harness)

RECURSION_PO2 = 18
CLAIM_WORDS = ctypes.sizeof(ReceiptClaim) // 4


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
    """A proven tree node (r0h_node): its seal and the composed claim it is carried with."""

    def __init__(self, handle):
        self.handle = handle
        p, n = _vp(), _sz(0)
        _check(lib().r0h_node_seal(handle, ctypes.byref(p), ctypes.byref(n)))
        self.seal = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value,)).copy()
        self.claim = ReceiptClaim()
        _check(lib().r0h_node_claim(handle, ctypes.byref(self.claim)))

    @classmethod
    def from_parts(cls, seal, claim):
        a, pa = _u32arr(seal)
        h = _vp()
        _check(lib().r0h_node_new(pa, a.size, ctypes.byref(claim), ctypes.byref(h)))
        return cls(h)

    def to_words(self):
        return np.concatenate([np.frombuffer(bytes(self.claim), dtype=np.uint32), self.seal]).astype(np.uint32)

    @classmethod
    def from_words(cls, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        if words.size <= CLAIM_WORDS:
            raise R0HipError("a node on the wire is %d claim words followed by its seal; got %d words" % (CLAIM_WORDS, words.size))
        claim = ReceiptClaim.from_buffer_copy(words[:CLAIM_WORDS].tobytes())
        return cls.from_parts(words[CLAIM_WORDS:], claim)

    @property
    def digest(self):
        return seal_digest(self.seal)

    @property
    def claim_words(self):
        """public inputs 0..7: the words naming the node's composed claim"""
        return self.seal[:8].copy()

    @property
    def consumed_digest(self):
        """public inputs 8..15: the Poseidon2 digest of what the node consumed (a segment seal, or its two children's digests)"""
        return self.seal[8:16].copy()

    def free(self):
        if self.handle:
            lib().r0h_node_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Recursor:
    """lift / join on one device (r0h_recursor).  `segment_blob` is the circuit the leaf seals belong to; `segment_roots`
    {po2: root[8]} binds the leaves to the segment program (without it they are verified against the circuit alone)."""

    def __init__(self, hal, recursion_blob, segment_blob, code_object=None, po2=RECURSION_PO2, segment_roots=None):
        self.hal, self.po2 = hal, po2
        self.recursion_blob = np.ascontiguousarray(recursion_blob, dtype=np.uint32)
        rb, prb = _u32arr(self.recursion_blob)
        sb, psb = _u32arr(segment_blob)
        roots = np.zeros(9 * max(len(segment_roots or {}), 1), dtype=np.uint32)
        for k, (size, root) in enumerate(sorted((segment_roots or {}).items())):
            roots[9 * k] = size
            roots[9 * k + 1:9 * k + 9] = root
        self.handle = _vp()
        _check(lib().r0h_recursor_new(hal.ctx, prb, rb.size, code_object.encode() if code_object else None, po2, psb, sb.size, roots.ctypes.data_as(_vp),
                                      len(segment_roots or {}), ctypes.byref(self.handle)))
        self.control_root = np.zeros(8, dtype=np.uint32)
        _check(lib().r0h_recursor_control_root(self.handle, self.control_root.ctypes.data_as(_vp)))

    def lift(self, segment_seal, claim):
        """risc0 `lift`: one recursion-circuit proof standing for one segment seal and its claim."""
        a, pa = _u32arr(segment_seal)
        h = _vp()
        _check(lib().r0h_lift(self.handle, pa, a.size, ctypes.byref(claim), ctypes.byref(h)))
        return Node(h)

    def join(self, a, b):
        """risc0 `join`: one proof standing for two nodes that follow one another; its claim is their composition."""
        h = _vp()
        _check(lib().r0h_join(self.handle, a.handle, b.handle, ctypes.byref(h)))
        return Node(h)

    def node_from_words(self, words):
        return Node.from_words(words)

    def verify(self, node):
        """the node's seal verifies bound to the recursion circuit's control root and names the claim it is carried with"""
        ok = ctypes.c_int(0)
        rb, prb = _u32arr(self.recursion_blob)
        _check(lib().r0h_node_verify(prb, rb.size, self.control_root.ctypes.data_as(_vp), node.handle, ctypes.byref(ok)))
        return bool(ok.value)

    def fold(self, nodes):
        """Left-to-right binary fold of this rank's own nodes (log depth); neighbours are joined, so consecutive segments stay consecutive."""
        nodes = list(nodes)
        while len(nodes) > 1:
            nxt = [self.join(nodes[i], nodes[i + 1]) for i in range(0, len(nodes) - 1, 2)]
            if len(nodes) & 1:
                nxt.append(nodes[-1])
            nodes = nxt
        return nodes[0]

    def close(self):
        if self.handle:
            _check(lib().r0h_recursor_free(self.handle))
            self.handle = None


FAILED = np.array([0xFFFFFFFF], dtype=np.uint32)  # handed up instead of a seal by a rank whose subtree could not be proved
EMPTY = np.zeros(0, dtype=np.uint32)               # handed up by a rank that owns no segment (segments < world)


def join_across_ranks(recursor, node, rank, world, send, recv):
    """Run the cross-rank part of the tree.  `send(array, dst)` / `recv(src) -> array` move one node (uint32 words: claim, seal).
    A rank's own nodes cover a contiguous run of segments and rank r's run precedes rank r + 1's (driver.shard_contiguous), so the
    receiver's node is always the LEFT operand of the join.
    Returns the root Node on rank 0 and None elsewhere.

    Every rank takes part in every exchange the schedule gives it, whatever happened before: a rank without a node (empty
    share) hands up an empty message and its partner keeps what it has; a rank whose join failed (or whose child reported a
    failure) hands up a failure marker instead of blocking its partner, so the failure reaches the root, every rank leaves
    the tree, and the first error is re-raised on the rank where it happened (and as "a child rank failed" above it)."""
    failed = None
    for _level, receiver, sender in tree_schedule(world):
        if rank == sender:
            send(FAILED if failed is not None else (EMPTY if node is None else node.to_words()), receiver)
            if failed is not None:
                raise failed
            return None  # this rank's subtree has been handed up
        if rank == receiver:
            other = recv(sender)
            if other.size == 1 and other[0] == FAILED[0]:
                failed = failed or R0HipError("join tree: rank %d reported a failure in its subtree" % sender)
            elif failed is None and other.size:
                try:
                    arrived = recursor.node_from_words(other)
                    node = arrived if node is None else recursor.join(node, arrived)
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
