"""Random RV32IM executions (the program generator of tools/soak_trace.py: ALU, M extension, loads / stores, forward branches and jumps,
an outer loop, a READ_WORDS / COMMIT pair) through the executor with the trace kept, on the CPU: the rows and boundary rows keep
their invariants -- every access names the previous access of its address, what is read is what was last written, the boundary rows
are each touched address once, in order, with the value found and the value left -- a run cut into segments hands its values on from
one segment to the next, and the CPU oracle proves every segment with the trace circuit and both verifiers accept it.  (The device
side of the same is tools/soak_trace.py and tests/test_trace_circuit.py -m gpu.)"""
import os
import sys

import numpy as np

import hyperfridge_r0_amd as r0
from conftest import ROOT, circuit_path

sys.path.insert(0, os.path.join(ROOT, "tools"))
from soak_trace import random_program  # noqa: E402

REG = r0.REG_BASE


STAMP = {0: 2, 1: 3, 2: 4, 3: 5, 4: 1}  # access (x[rs1], x[rs2], x[rd], memory, fetch) -> its place in the cycle: the fetch comes first


def _check_segment(vm, k, carried, holder):
    """invariants of segment k; `carried` maps address -> value left by earlier segments, `holder` address -> the segment that left it
    (both updated)"""
    last, value = {}, {}
    rows = vm.preflight(k)
    for c, w in enumerate(rows):
        assert w.cycle == c and (c == 0 or w.pc == rows[c - 1].next_pc)
        i1, i2 = ((w.insn >> 15) & 31, (w.insn >> 20) & 31) if w.insn != 0x73 else (17, 10)  # an ecall reads a7 and a0
        # x0 is a register like any other to the memory argument: every cycle reads two registers
        acc = [(REG + i1, w.rs1_value, w.rs1_value), (REG + i2, w.rs2_value, w.rs2_value),
               (REG + w.rd, w.rd_before, w.rd_after) if w.rd else None, (w.mem_addr >> 2, w.mem_before, w.mem_after) if w.mem_kind else None,
               (w.pc >> 2, w.insn, w.insn)]
        if not i1:
            assert w.rs1_value == 0
        if w.mem_kind == r0.MEM_READ:
            assert w.mem_before == w.mem_after
        for slot in (4, 0, 1, 2, 3):  # in the order of their timestamps
            a = acc[slot]
            if a is None:
                continue
            addr, before, after = a
            assert w.prev[slot] == last.get(addr, 0), (k, c, slot)
            if addr in value:
                assert value[addr][1] == before, (k, c, slot, hex(addr))
            else:
                value[addr] = [before, before]
            value[addr][1] = after
            last[addr] = 5 * c + STAMP[slot]
    bounds = vm.boundary(k)
    seg = vm.segments()[k]
    own = [b for b in bounds if b.addr in value]
    assert [b.addr for b in own] == sorted(value) and [b.addr for b in bounds] == sorted(b.addr for b in bounds)
    if not seg.closing:
        assert len(own) == len(bounds)
    for b in bounds:
        if b.addr in value:
            assert (b.first_value, b.last_value, b.last_ts) == (value[b.addr][0], value[b.addr][1], last[b.addr])
        else:  # a closing row of an address this segment never touched: found and left as it is
            assert seg.closing and b.first_value == b.last_value and b.last_ts == 0
        if b.addr in carried:
            assert carried[b.addr] == b.first_value, (k, hex(b.addr))  # what an earlier segment left there
        assert b.prev_seg == holder.get(b.addr, 0) and b.prev_seg <= k, (k, hex(b.addr))
        carried[b.addr] = b.last_value
        holder[b.addr] = k + 1
    return len(rows), len(bounds)


def test_random_executions_keep_the_trace_invariants_and_prove(orc):
    rng = np.random.default_rng(7)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    size = r0.TRACE_MIN_PO2
    code = c.witgen(size, 0)[0]
    root = c.code_root(code, size)
    proved = cycles = 0
    for trial in range(24):
        prog = random_program(rng, int(rng.integers(30, 400)))
        vm = r0.Vm()
        vm.load(0x1000, prog)
        vm.set_pc(0x1000)
        regs = {}
        for i in range(1, 28):
            regs[REG + i] = int(rng.integers(0, 1 << 32)) if rng.random() < 0.7 else int(rng.choice([0, 1, 0xFFFFFFFF, 0x80000000]))
            vm.set_reg(i, regs[REG + i])
        vm.set_input([int(v) for v in rng.integers(0, 1 << 32, 8)])
        po2 = 9 if trial % 3 else 20  # every run but each third is cut into 2^9-row segments
        assert vm.run(segment_po2=po2, keep_trace=True, boundary_rows=True) == (0, 0)
        segs = vm.segments()
        carried, holder = dict(regs), {}
        image = {(0x1000 >> 2) + i for i in range(len(prog))}
        for k, s in enumerate(segs):
            n_rows, n_bounds = _check_segment(vm, k, carried, holder)
            assert n_rows == s.user_cycles and n_bounds == s.boundary_rows and n_rows + n_bounds <= 1 << po2
            cycles += n_rows
            if trial in (1, 3) and k in (0, len(segs) - 1):  # prove the first and the last segment of two runs (2^16 rows each: the tables' size)
                data, glob = vm.trace_witness(k, size)
                glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(int(v)) for v in rng.integers(0, 2013265921, 16)]  # any challenge will do for one seal
                glob = c.logup_totals(size, code, data, glob)
                seal = c.prove(size, code, data, glob)
                assert c.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
                proved += 1
        # the rows that close the session: every word and register touched, every image word, each once, in address order, with its initial value
        closing = [b for k, s in enumerate(segs) if s.closing for b in vm.boundary(k)]
        assert segs[-1].closing and [b.addr for b in closing] == sorted(set(holder) | image) and all(s.user_cycles == 0 for s in segs if s.closing and s.index != [t.index for t in segs if t.user_cycles][-1])
        for b in closing:
            in_image = b.addr in image
            assert (b.flags & 1) == in_image and b.init_value == (prog[b.addr - (0x1000 >> 2)] if in_image else 0)
        assert sum(s.user_cycles for s in segs) == vm.cycles and (po2 == 20 or len(segs) >= 2 or vm.cycles < 400)
    assert proved >= 3 and cycles > 10_000
