"""A trace-circuit session proved step by step from the tests, so that a test can lie at any stage -- another claim, another boundary
value, another journal: what r0h_prove_elf does in one call (csrc/session.cpp), spelled out over the C ABI's pieces.  The prover is
either the CPU oracle (`backend` = an OrcCircuit: small programs, non-gpu tests) or the device (`backend` = (hal, loaded circuit)).

  phase 1  per segment: witness (host reference r0h_vm_trace_witness, or the device kernel), multiplicities, DATA root
  between  the session challenge from every segment's record (early public inputs, DATA root)
  phase 2  per segment: the segment's sum under the challenge, the seal
"""
import numpy as np

import hyperfridge_r0_amd as r0

EARLY = r0.TRACE_GLOBALS - r0.TRACE_LATE_GLOBALS


def size_of(vm, k):
    s = vm.segments()[k]
    return max(r0.TRACE_MIN_PO2, int(np.ceil(np.log2(max(2, s.user_cycles + s.boundary_rows)))))


def witnesses(vm, claims=None, edit=None, po2=None):
    """-> [(po2, data, glob)] per segment from the host reference; `edit(k, data, glob)` may change a witness in place (multiplicities
    are recounted afterwards when it returns True)"""
    claims = claims or vm.claims()
    out = []
    for k in range(len(vm.segments())):
        size = po2 or size_of(vm, k)
        data, glob = vm.trace_witness(k, size, claim_globals=claims[k].globals())
        if edit is not None and edit(k, data, glob):
            blob = r0.trace_blob()
            r0._check(r0.lib().r0h_logup_multiplicities_host(blob.ctypes.data_as(r0._vp), blob.size, size, data.ctypes.data_as(r0._vp), glob.ctypes.data_as(r0._vp)))
        out.append((size, data, glob))
    return out


class OracleProver:
    def __init__(self, oc):
        self.oc, self.code = oc, {}

    def fixed(self, po2):
        if po2 not in self.code:
            code = self.oc.witgen(po2, 0)[0]
            self.code[po2] = (code, self.oc.code_root(code, po2))
        return self.code[po2]

    def data_root(self, po2, data):
        out = np.zeros(8, dtype=np.uint32)
        self.oc.o.L.orc_code_root(data.ctypes.data_as(r0._vp), self.oc.group_size[2], po2, out.ctypes.data_as(r0._vp))  # the root of any committed group
        return out

    def totals(self, po2, data, glob):
        return self.oc.logup_totals(po2, self.fixed(po2)[0], data, glob)

    def prove(self, po2, data, glob):
        return self.oc.prove(po2, self.fixed(po2)[0], data, glob)

    def control_root(self, po2):
        return self.fixed(po2)[1]


def prove_session(prover, vm, claims=None, edit=None, journal=None, po2=None, challenge=None):
    """-> (Receipt, {po2: control root}).  `challenge(records)` may replace the session challenge (a prover that picks its own)."""
    claims = claims or vm.claims()
    wit = witnesses(vm, claims, edit, po2)
    records = np.zeros((len(wit), r0.SESSION_RECORD_WORDS), dtype=np.uint32)
    for k, (size, data, glob) in enumerate(wit):
        records[k, :EARLY] = glob[:EARLY]
        records[k, EARLY:] = prover.data_root(size, data)
    gamma = r0.session_challenge(records) if challenge is None else challenge(records)
    seals, roots = [], {}
    for size, data, glob in wit:
        glob = glob.copy()
        glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = gamma
        glob = prover.totals(size, data, glob)
        seals.append(prover.prove(size, data, glob))
        roots[size] = prover.control_root(size)
    return r0.Receipt.new(vm.journal if journal is None else journal, seals, claims), roots
