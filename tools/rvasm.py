#!/usr/bin/env python3
"""A small RV32IM assembler (labels, forward references, lui+addi address loads) for the hand-written guests of this repository.
The reference's guest is Rust compiled by the risc0 toolchain (methods/build.rs:2), which is absent here; guests are therefore
written as instruction streams.  Encodings follow the RISC-V unprivileged specification (chapter 2, M extension); the executor
that runs them (csrc/rv32im.cpp) is checked against an independent interpreter in tests/test_rv32im.py."""
import struct

ZERO, RA, SP, GP, TP, T0, T1, T2, S0, S1, A0, A1, A2, A3, A4, A5, A6, A7 = range(18)
S2, S3, S4, S5, S6, S7, S8, S9, S10, S11, T3, T4, T5, T6 = range(18, 32)


def _r(f7, rs2, rs1, f3, rd, op=0x33):
    return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def _i(imm, rs1, f3, rd, op):
    assert -2048 <= imm < 2048, imm
    return ((imm & 0xFFF) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def _s(imm, rs2, rs1, f3):
    assert -2048 <= imm < 2048, imm
    return (((imm >> 5) & 0x7F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((imm & 0x1F) << 7) | 0x23


def _b(imm, rs2, rs1, f3):
    assert -4096 <= imm < 4096 and imm % 2 == 0, imm
    return (((imm >> 12) & 1) << 31) | (((imm >> 5) & 0x3F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (((imm >> 1) & 0xF) << 8) | (((imm >> 11) & 1) << 7) | 0x63


def _j(imm, rd):
    assert -(1 << 20) <= imm < (1 << 20) and imm % 2 == 0, imm
    return (((imm >> 20) & 1) << 31) | (((imm >> 1) & 0x3FF) << 21) | (((imm >> 11) & 1) << 20) | (((imm >> 12) & 0xFF) << 12) | (rd << 7) | 0x6F


class Asm:
    def __init__(self, base):
        self.base, self.words, self.labels, self.fix = base, [], {}, []

    # ---- positions
    @property
    def pc(self):
        return self.base + 4 * len(self.words)

    def label(self, name):
        assert name not in self.labels, name
        self.labels[name] = self.pc

    def emit(self, w):
        self.words.append(w & 0xFFFFFFFF)

    def _later(self, kind, target, *args):
        self.fix.append((len(self.words), kind, target, args))
        self.emit(0)

    # ---- instructions
    def lui(self, rd, imm20): self.emit(((imm20 & 0xFFFFF) << 12) | (rd << 7) | 0x37)
    def addi(self, rd, rs1, imm): self.emit(_i(imm, rs1, 0, rd, 0x13))
    def slti(self, rd, rs1, imm): self.emit(_i(imm, rs1, 2, rd, 0x13))
    def sltiu(self, rd, rs1, imm): self.emit(_i(imm, rs1, 3, rd, 0x13))
    def xori(self, rd, rs1, imm): self.emit(_i(imm, rs1, 4, rd, 0x13))
    def ori(self, rd, rs1, imm): self.emit(_i(imm, rs1, 6, rd, 0x13))
    def andi(self, rd, rs1, imm): self.emit(_i(imm, rs1, 7, rd, 0x13))
    def slli(self, rd, rs1, sh): self.emit(_i(sh, rs1, 1, rd, 0x13))
    def srli(self, rd, rs1, sh): self.emit(_i(sh, rs1, 5, rd, 0x13))
    def srai(self, rd, rs1, sh): self.emit(_i(0x400 | sh, rs1, 5, rd, 0x13))
    def add(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 0, rd))
    def sub(self, rd, rs1, rs2): self.emit(_r(0x20, rs2, rs1, 0, rd))
    def sll(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 1, rd))
    def sltu(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 3, rd))
    def xor(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 4, rd))
    def srl(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 5, rd))
    def or_(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 6, rd))
    def and_(self, rd, rs1, rs2): self.emit(_r(0, rs2, rs1, 7, rd))
    def mul(self, rd, rs1, rs2): self.emit(_r(1, rs2, rs1, 0, rd))
    def mulhu(self, rd, rs1, rs2): self.emit(_r(1, rs2, rs1, 3, rd))
    def lw(self, rd, off, rs1): self.emit(_i(off, rs1, 2, rd, 0x03))
    def lbu(self, rd, off, rs1): self.emit(_i(off, rs1, 4, rd, 0x03))
    def sw(self, rs2, off, rs1): self.emit(_s(off, rs2, rs1, 2))
    def sb(self, rs2, off, rs1): self.emit(_s(off, rs2, rs1, 0))
    def ecall(self): self.emit(0x73)
    def mv(self, rd, rs): self.addi(rd, rs, 0)
    def not_(self, rd, rs): self.xori(rd, rs, -1)
    def ret(self): self.emit(_i(0, RA, 0, ZERO, 0x67))

    def li(self, rd, v):
        v &= 0xFFFFFFFF
        sv = v - (1 << 32) if v >> 31 else v
        if -2048 <= sv < 2048:
            self.addi(rd, ZERO, sv)
            return
        self.lui(rd, ((v + 0x800) >> 12) & 0xFFFFF)
        lo = v & 0xFFF
        lo = lo - 0x1000 if lo >= 0x800 else lo
        if lo:
            self.addi(rd, rd, lo)

    def la(self, rd, target):  # always two words, so layouts do not depend on where a label lands
        self._later("la_hi", target, rd)
        self._later("la_lo", target, rd)

    def beq(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 0)
    def bne(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 1)
    def blt(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 4)
    def bge(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 5)
    def bltu(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 6)
    def bgeu(self, rs1, rs2, t): self._later("b", t, rs1, rs2, 7)
    def j(self, t): self._later("j", t, ZERO)
    def call(self, t): self._later("j", t, RA)

    def rotr(self, rd, rs, n, tmp):  # rd = rs rotated right by n (rd != rs, tmp scratch)
        self.srli(rd, rs, n)
        self.slli(tmp, rs, 32 - n)
        self.or_(rd, rd, tmp)

    # ---- finish
    def assemble(self):
        for at, kind, target, args in self.fix:
            addr = self.labels[target] if isinstance(target, str) else target
            here = self.base + 4 * at
            if kind == "b":
                self.words[at] = _b(addr - here, args[1], args[0], args[2])
            elif kind == "j":
                self.words[at] = _j(addr - here, args[0])
            elif kind == "la_hi":
                self.words[at] = (((addr + 0x800) >> 12) & 0xFFFFF) << 12 | (args[0] << 7) | 0x37
            elif kind == "la_lo":
                lo = addr & 0xFFF
                self.words[at] = _i(lo - 0x1000 if lo >= 0x800 else lo, args[0], 0, args[0], 0x13)
        return list(self.words)


def elf(segments, entry):
    """ELF32 little-endian RISC-V executable from [(vaddr, bytes, flags)]."""
    n = len(segments)
    ehdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, entry, 52, 0, 0, 52, 32, n, 0, 0, 0)
    off, ph, body = 52 + 32 * n, b"", b""
    for vaddr, data, flags in segments:
        data = bytes(data) + bytes(-len(data) % 4)
        ph += struct.pack("<IIIIIIII", 1, off, vaddr, vaddr, len(data), len(data), flags, 4)
        body += data
        off += len(data)
    return ehdr + ph + body
