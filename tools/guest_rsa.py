#!/usr/bin/env python3
"""A guest-shaped RV32IM program over the reference's own inputs (hand-assembled with tools/rvasm.py; the reference's guest is Rust and
needs the risc0 toolchain): what hyperfridge's guest spends its cycles on -- RSA-2048 and SHA-256
(methods/guest/src/main.rs:450-485 verify_bank_signature, :513-517 the two digests, :663-718 the transaction key, :757-833 the
witness signature; docs/hyperfridge-cycles.html: 83 % `BigUint::modpow`).  It

  1. reads `<xml>-SignedInfo`, hashes it (SHA-256, FIPS 180-4), raises the bank's signature to 65537 modulo the bank's modulus by
     Montgomery multiplication (32-bit limbs, CIOS) and compares the result with the PKCS#1 v1.5 encoded message
     00 01 FF..FF 00 || DigestInfo(SHA-256) || digest                                  -- exit 1 if it differs;
  2. raises `<xml>-TransactionKeyDecrypt.bin` to 65537 modulo the client's modulus and compares with the <TransactionKey>
     ciphertext (the guest's check that the supplied key is the one the bank encrypted)   -- exit 2;
  3. hashes the base64-decoded order data and checks the witness signature over that digest the same way as (1) -- exit 3;
  4. commits a serde-framed JSON string with both digests in hex and halts with 0.

AES, inflate and the camt53 XML parse are not reproduced (the statement values the real guest commits come from there).  The
modular arithmetic: R = 2^2048; n' = -n^-1 mod 2^32 by Newton's iteration; R mod n = 2^2048 - n; R^2 mod n from R mod n by 32 modular
doublings and 6 Montgomery squarings; x^65537 = 16 squarings and one product in Montgomery form.  About 3.5 M cycles per RSA
operation, 10.6 M for the run.  tests/test_guest_rsa.py pins the journal on the SURVEY.md section 4 known answers."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rvasm import (A0, A1, A2, A3, A4, A5, A6, A7, RA, S0, S1, S2, S3, S4, S5, S6, S7, S8, S9, S10, S11, T0, T1, T2, T3, T4, T5, T6, ZERO,  # noqa: E402
                   Asm, elf)

TEXT, DATA, BSS = 0x10000, 0x20000, 0x30000
JOURNAL_BASE = 0x20000000  # R0H_JOURNAL_BASE (include/r0hip.h): journal word i is the word at JOURNAL_BASE + 4 i when COMMIT names it
K256 = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
        0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
        0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
        0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
H256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
DER_SHA256 = bytes.fromhex("3031300d060960864801650304020105000420")  # DigestInfo prefix of SHA-256 (RFC 8017 section 9.2 note 1)
TEMPLATE = b'{"signed_info_sha256":"' + b"0" * 64 + b'","order_data_sha256":"' + b"0" * 64 + b'","bank_signature":"ok","transaction_key":"ok","witness_signature":"ok"}'
MAX_MSG = 8192  # bytes of a hashed input (SignedInfo is ~0.8 KB, the order data 2864 B), plus room for the padding


class Layout:
    def __init__(self):
        self.data, self.data_sym, self.bss_sym, self.bss_top = b"", {}, {}, BSS

    def const(self, name, payload):
        self.data += bytes(-len(self.data) % 4)
        self.data_sym[name] = DATA + len(self.data)
        self.data += bytes(payload)

    def var(self, name, n_bytes):
        self.bss_sym[name] = self.bss_top
        self.bss_top += (n_bytes + 3) & ~3

    def fixed(self, name, addr):
        """a buffer at an address the executor's ABI prescribes (the journal window)"""
        self.bss_sym[name] = addr

    def __getitem__(self, name):
        return self.data_sym[name] if name in self.data_sym else self.bss_sym[name]


def build(extend=None):
    """The ELF image, the assembler's labels and the memory layout.  `extend` (tools/guest_camt53.py) is an object with hooks
    layout(L), after_signed_info(a, L, fresh, halt) -- emitted when SignedInfo has been hashed and is still in MSG --,
    after_operands(a, L, k) -- emitted when the operands of RSA operation k = 0, 1, 2 have been read --, main(a, L, fresh,
    halt) -- emitted after the three RSA checks, before the commit of this file (which it replaces when it returns True) -- and
    library(a, L, fresh), emitted after the library of this file."""
    L = Layout()
    L.const("K", struct.pack("<64I", *K256))
    L.const("H0", struct.pack("<8I", *H256))
    L.const("DER", DER_SHA256)
    frame = struct.pack("<I", len(TEMPLATE)) + TEMPLATE
    L.const("JSON", frame)
    for name, size in (("LEN", 4), ("MSG", MAX_MSG + 128), ("W", 256), ("DIGEST", 32), ("DIGEST2", 32), ("BASE", 256), ("N", 256), ("WANT", 256), ("EXPO", 4),
                       ("N0INV", 4), ("T", 66 * 4), ("ONE_M", 256), ("R2", 256), ("X", 256), ("ACC", 256), ("OUT", 256), ("PLAIN1", 256), ("EM", 256)):
        L.var(name, size)
    if extend:
        extend.layout(L)
    a = Asm(TEXT)
    counter = [0]

    def fresh(prefix):
        counter[0] += 1
        return "%s_%d" % (prefix, counter[0])

    # ------------------------------------------------------------------ entry
    def read_words(sym, count_reg=None, count=None):
        a.la(A0, L[sym])
        if count is not None:
            a.li(A1, count)
        else:
            a.mv(A1, count_reg)
        a.li(A7, 1)
        a.ecall()

    def halt(code):
        a.li(A0, code)
        a.li(A7, 0)
        a.ecall()

    def read_message():  # [byte length][bytes, packed little-endian into words]
        read_words("LEN", count=1)
        a.la(T0, L["LEN"])
        a.lw(S0, 0, T0)
        a.li(T1, MAX_MSG)
        ok = fresh("len_ok")
        a.bgeu(T1, S0, ok)
        halt(5)
        a.label(ok)
        a.addi(T0, S0, 3)
        a.srli(T0, T0, 2)
        read_words("MSG", count_reg=T0)

    def read_rsa_operands(with_want):  # base (64 words), modulus (64), [expected value (64)], exponent (1, must be 65537)
        read_words("BASE", count=64)
        read_words("N", count=64)
        if with_want:
            read_words("WANT", count=64)
        read_words("EXPO", count=1)
        a.la(T0, L["EXPO"])
        a.lw(T0, 0, T0)
        a.li(T1, 65537)
        ok = fresh("e_ok")
        a.beq(T0, T1, ok)
        halt(4)
        a.label(ok)

    def hash_message(out_sym):
        a.la(A0, L["MSG"])
        a.mv(A1, S0)
        a.la(A2, L[out_sym])
        a.call("sha256")

    def check_signature(digest_sym, fail_code):  # OUT = BASE^65537 mod N must be the PKCS#1 v1.5 encoding of the digest
        a.call("rsa_pub")
        a.la(A0, L[digest_sym])
        a.call("build_em")
        a.la(A0, L["OUT"])
        a.la(A1, L["EM"])
        a.call("cmp256")
        ok = fresh("sig_ok")
        a.beq(A0, ZERO, ok)
        halt(fail_code)
        a.label(ok)

    a.label("_start")
    # 1. the bank's signature over SHA-256(SignedInfo)
    read_message()
    hash_message("DIGEST")
    if extend:
        extend.after_signed_info(a, L, fresh, halt)  # SignedInfo is still in MSG, its length in LEN
    read_rsa_operands(False)
    if extend:
        extend.after_operands(a, L, 0)
    check_signature("DIGEST", 1)
    # 2. the transaction key: decrypted block ^ e mod n_client == the ciphertext in the response
    read_rsa_operands(True)
    if extend:
        extend.after_operands(a, L, 1)
    if extend:  # the decrypted block is needed again (its last 16 bytes are the AES key): BASE is reused by step 3
        a.la(T0, L["BASE"])
        a.la(T1, L["TXBLOCK"])
        a.addi(T2, T0, 256)
        a.label("keep_block")
        a.lw(T3, 0, T0)
        a.sw(T3, 0, T1)
        a.addi(T0, T0, 4)
        a.addi(T1, T1, 4)
        a.bne(T0, T2, "keep_block")
    a.call("rsa_pub")
    a.la(A0, L["OUT"])
    a.la(A1, L["WANT"])
    a.call("cmp256")
    a.beq(A0, ZERO, "tx_ok")
    halt(2)
    a.label("tx_ok")
    # 3. the witness signature over SHA-256(decoded order data)
    read_message()
    hash_message("DIGEST2")
    read_rsa_operands(False)
    if extend:
        extend.after_operands(a, L, 2)
    check_signature("DIGEST2", 3)
    if extend and extend.main(a, L, fresh, halt):
        pass  # the extension commits and halts itself
    else:
        emit_commit(a, L, frame, halt)
    emit_library(a, L)
    if extend:
        extend.library(a, L, fresh)
    words = a.assemble()
    image = elf([(TEXT, struct.pack("<%dI" % len(words), *words), 5), (DATA, L.data, 6)], a.labels["_start"])
    return image, a.labels, L


def emit_commit(a, L, frame, halt):
    # 4. the commitment: both digests in hex inside the JSON template, serde-framed
    a.la(A0, L["DIGEST"])
    a.la(A1, L["JSON"] + 4 + TEMPLATE.index(b"0" * 64))
    a.call("hex32")
    a.la(A0, L["DIGEST2"])
    a.la(A1, L["JSON"] + 4 + TEMPLATE.rindex(b"0" * 64))
    a.call("hex32")
    # the journal is a window of memory (R0H_JOURNAL_BASE: word i of it at + 4 i): the frame moves there, then COMMIT names its words
    n_words = (len(frame) + 3) >> 2
    a.la(T0, L["JSON"])
    a.li(T1, JOURNAL_BASE)
    a.li(T2, n_words)
    a.label("commit_copy")
    a.lw(T3, 0, T0)
    a.sw(T3, 0, T1)
    a.addi(T0, T0, 4)
    a.addi(T1, T1, 4)
    a.addi(T2, T2, -1)
    a.bne(T2, ZERO, "commit_copy")
    a.li(A0, JOURNAL_BASE)
    a.li(A1, n_words)  # COMMIT takes words
    a.li(A7, 2)
    a.ecall()
    halt(0)


def emit_library(a, L):
    # ------------------------------------------------------------------ hex32(a0 = 8 digest words, a1 = 64 output bytes)
    a.label("hex32")
    a.li(T0, 0)
    a.label("hx_word")
    a.add(T1, A0, T0)
    a.lw(T2, 0, T1)            # a digest word: its most significant byte comes first in the hash
    a.li(T3, 8)
    a.label("hx_nib")
    a.srli(T4, T2, 28)
    a.slli(T2, T2, 4)
    a.sltiu(T5, T4, 10)
    a.addi(T4, T4, 87)         # 'a' - 10
    a.beq(T5, ZERO, "hx_put")
    a.addi(T4, T4, 48 - 87)    # '0'
    a.label("hx_put")
    a.sb(T4, 0, A1)
    a.addi(A1, A1, 1)
    a.addi(T3, T3, -1)
    a.bne(T3, ZERO, "hx_nib")
    a.addi(T0, T0, 4)
    a.li(T6, 32)
    a.bne(T0, T6, "hx_word")
    a.ret()

    # ------------------------------------------------------------------ cmp256(a0, a1): a0 = 0 iff the two 64-word numbers are equal
    a.label("cmp256")
    a.li(T0, 0)
    a.li(T3, 0)
    a.label("cmp_l")
    a.add(T1, A0, T0)
    a.lw(T1, 0, T1)
    a.add(T2, A1, T0)
    a.lw(T2, 0, T2)
    a.xor(T1, T1, T2)
    a.or_(T3, T3, T1)
    a.addi(T0, T0, 4)
    a.li(T6, 256)
    a.bne(T0, T6, "cmp_l")
    a.mv(A0, T3)
    a.ret()

    # ------------------------------------------------------------------ build_em(a0 = digest words): EM = 00 01 FF.. 00 DER digest as 64 LE limbs
    # limb j holds bytes 252-4j .. 255-4j of the 256-byte big-endian message, most significant first
    a.label("build_em")
    a.la(T0, L["EM"])
    a.li(T1, 0)
    a.label("em_digest")       # limbs 0..7: digest words 7..0 (the digest is the last 32 bytes)
    a.li(T2, 28)
    a.sub(T2, T2, T1)
    a.add(T2, A0, T2)
    a.lw(T3, 0, T2)
    a.add(T4, T0, T1)
    a.sw(T3, 0, T4)
    a.addi(T1, T1, 4)
    a.li(T6, 32)
    a.bne(T1, T6, "em_digest")
    # limbs 8..12: 00 || DER (19 bytes) = 20 bytes: big-endian bytes 204..223
    der = b"\x00" + DER_SHA256
    for j in range(5):
        chunk = der[20 - 4 * (j + 1):20 - 4 * j]
        a.li(T3, int.from_bytes(chunk, "big"))
        a.sw(T3, 32 + 4 * j, T0)
    a.li(T3, -1)
    for j in range(13, 63):    # FF padding: bytes 4..203
        a.sw(T3, 4 * j, T0)
    a.li(T3, 0x0001FFFF)       # bytes 0..3: 00 01 FF FF
    a.sw(T3, 252, T0)
    a.ret()

    # ------------------------------------------------------------------ sha256(a0 = message (room for padding), a1 = length, a2 = 8 output words)
    a.label("sha256")
    a.mv(S1, A2)
    a.add(T0, A0, A1)
    a.li(T1, 0x80)
    a.sb(T1, 0, T0)
    a.addi(T0, T0, 1)
    a.li(T3, 56)
    a.label("pad_l")
    a.sub(T2, T0, A0)
    a.andi(T2, T2, 63)
    a.beq(T2, T3, "pad_d")
    a.sb(ZERO, 0, T0)
    a.addi(T0, T0, 1)
    a.j("pad_l")
    a.label("pad_d")
    for k in range(4):
        a.sb(ZERO, k, T0)
    a.slli(T1, A1, 3)          # bit length (< 2^32), big-endian
    for k, sh in enumerate((24, 16, 8, 0)):
        a.srli(T2, T1, sh)
        a.sb(T2, 4 + k, T0)
    a.addi(S2, T0, 8)          # end of the padded message
    a.mv(S0, A0)               # current block
    a.la(T0, L["H0"])
    for k in range(8):
        a.lw(T1, 4 * k, T0)
        a.sw(T1, 4 * k, S1)
    a.la(S10, L["W"])
    a.la(S11, L["K"])
    a.li(T6, 0xFF00)           # byte-swap mask, kept for the whole call
    a.label("blk")
    a.li(T0, 0)
    a.label("w_load")          # W[0..15]: the block's words, big-endian
    a.add(T1, S0, T0)
    a.lw(T2, 0, T1)
    a.slli(T3, T2, 24)
    a.and_(T4, T2, T6)
    a.slli(T4, T4, 8)
    a.or_(T3, T3, T4)
    a.srli(T4, T2, 8)
    a.and_(T4, T4, T6)
    a.or_(T3, T3, T4)
    a.srli(T4, T2, 24)
    a.or_(T3, T3, T4)
    a.add(T1, S10, T0)
    a.sw(T3, 0, T1)
    a.addi(T0, T0, 4)
    a.li(T5, 64)
    a.bne(T0, T5, "w_load")
    a.label("w_ext")           # W[i] = W[i-16] + s0(W[i-15]) + W[i-7] + s1(W[i-2]),  T0 = 4 i
    a.add(T1, S10, T0)
    a.lw(T2, -60, T1)
    a.rotr(T3, T2, 7, T5)
    a.rotr(T4, T2, 18, T5)
    a.xor(T3, T3, T4)
    a.srli(T4, T2, 3)
    a.xor(T3, T3, T4)
    a.lw(T2, -8, T1)
    a.rotr(T4, T2, 17, T5)
    a.xor(A3, T4, ZERO)
    a.rotr(T4, T2, 19, T5)
    a.xor(A3, A3, T4)
    a.srli(T4, T2, 10)
    a.xor(A3, A3, T4)
    a.add(T3, T3, A3)
    a.lw(T2, -64, T1)
    a.add(T3, T3, T2)
    a.lw(T2, -28, T1)
    a.add(T3, T3, T2)
    a.sw(T3, 0, T1)
    a.addi(T0, T0, 4)
    a.li(T5, 256)
    a.bne(T0, T5, "w_ext")
    regs = [S3, S4, S5, S6, S7, S8, S9, A6]  # a b c d e f g h
    for k, r in enumerate(regs):
        a.lw(r, 4 * k, S1)
    ra_, rb, rc, rd_, re, rf, rg, rh = regs
    a.li(T0, 0)
    a.label("round")
    a.rotr(T1, re, 6, T5)
    a.rotr(T2, re, 11, T5)
    a.xor(T1, T1, T2)
    a.rotr(T2, re, 25, T5)
    a.xor(T1, T1, T2)          # S1
    a.and_(T2, re, rf)
    a.not_(T3, re)
    a.and_(T3, T3, rg)
    a.xor(T2, T2, T3)          # ch
    a.add(T1, T1, T2)
    a.add(T1, T1, rh)
    a.add(T2, S11, T0)
    a.lw(T2, 0, T2)
    a.add(T1, T1, T2)
    a.add(T2, S10, T0)
    a.lw(T2, 0, T2)
    a.add(T1, T1, T2)          # temp1
    a.rotr(T2, ra_, 2, T5)
    a.rotr(T3, ra_, 13, T5)
    a.xor(T2, T2, T3)
    a.rotr(T3, ra_, 22, T5)
    a.xor(T2, T2, T3)          # S0
    a.and_(T3, ra_, rb)
    a.and_(T4, ra_, rc)
    a.xor(T3, T3, T4)
    a.and_(T4, rb, rc)
    a.xor(T3, T3, T4)          # maj
    a.add(T2, T2, T3)          # temp2
    a.mv(rh, rg)
    a.mv(rg, rf)
    a.mv(rf, re)
    a.add(re, rd_, T1)
    a.mv(rd_, rc)
    a.mv(rc, rb)
    a.mv(rb, ra_)
    a.add(ra_, T1, T2)
    a.addi(T0, T0, 4)
    a.li(T5, 256)
    a.bne(T0, T5, "round")
    for k, r in enumerate(regs):
        a.lw(T1, 4 * k, S1)
        a.add(T1, T1, r)
        a.sw(T1, 4 * k, S1)
    a.addi(S0, S0, 64)
    a.bne(S0, S2, "blk")
    a.ret()

    # ------------------------------------------------------------------ montmul(a0 = dst, a1 = a, a2 = b): dst = a b / R mod N  (CIOS, 64 limbs)
    a.label("montmul")
    a.la(T0, L["T"])
    a.addi(T1, T0, 66 * 4)
    a.label("mm_zero")
    a.sw(ZERO, 0, T0)
    a.addi(T0, T0, 4)
    a.bne(T0, T1, "mm_zero")
    a.addi(A7, A2, 256)        # end of b
    a.la(T2, L["N0INV"])
    a.lw(S9, 0, T2)            # n' (s9 is dead outside sha256's rounds)
    a.label("mm_outer")
    a.lw(A3, 0, A2)            # b[i]
    a.mv(T0, A1)
    a.la(T1, L["T"])
    a.li(A5, 0)
    a.addi(T6, T0, 256)

    def mac(x_reg, store_off):  # (C, S) = T[j] + mem[t0] * x + C ; T[j + store_off / 4] = S
        a.lw(T2, 0, T0)
        a.lw(T3, 0, T1)
        a.mul(T4, T2, x_reg)
        a.mulhu(T5, T2, x_reg)
        a.add(T4, T4, T3)
        a.sltu(T3, T4, T3)
        a.add(T4, T4, A5)
        a.sltu(T2, T4, A5)
        a.add(A5, T5, T3)
        a.add(A5, A5, T2)
        a.sw(T4, store_off, T1)
        a.addi(T0, T0, 4)
        a.addi(T1, T1, 4)

    a.label("mm_l1")
    mac(A3, 0)
    a.bne(T0, T6, "mm_l1")
    a.lw(T3, 0, T1)            # T[64] += C, T[65] = carry
    a.add(T4, T3, A5)
    a.sltu(T2, T4, A5)
    a.sw(T4, 0, T1)
    a.sw(T2, 4, T1)
    a.la(T1, L["T"])
    a.lw(T3, 0, T1)
    a.mul(A4, T3, S9)          # m = T[0] n' mod 2^32
    a.la(T0, L["N"])
    a.lw(T2, 0, T0)
    a.mul(T4, A4, T2)
    a.mulhu(T5, A4, T2)
    a.add(T4, T4, T3)          # = 0 mod 2^32
    a.sltu(T3, T4, T3)
    a.add(A5, T5, T3)
    a.addi(T0, T0, 4)
    a.addi(T1, T1, 4)
    a.addi(T6, T0, 252)
    a.label("mm_l2")
    mac(A4, -4)
    a.bne(T0, T6, "mm_l2")
    a.lw(T3, 0, T1)            # (C, S) = T[64] + C ; T[63] = S ; T[64] = T[65] + C
    a.add(T4, T3, A5)
    a.sltu(T2, T4, A5)
    a.sw(T4, -4, T1)
    a.lw(T3, 4, T1)
    a.add(T3, T3, T2)
    a.sw(T3, 0, T1)
    a.addi(A2, A2, 4)
    a.bne(A2, A7, "mm_outer")
    a.la(A1, L["T"])           # reduce once if T >= N, then copy out
    a.mv(S8, RA)
    a.call("cond_sub_n")
    a.mv(RA, S8)
    a.la(T1, L["T"])
    a.li(T0, 0)
    a.label("mm_copy")
    a.add(T2, T1, T0)
    a.lw(T3, 0, T2)
    a.add(T2, A0, T0)
    a.sw(T3, 0, T2)
    a.addi(T0, T0, 4)
    a.li(T6, 256)
    a.bne(T0, T6, "mm_copy")
    a.ret()

    # ------------------------------------------------------------------ cond_sub_n(a1 = x, 65 limbs: 64 + an overflow word): x -= N if x >= N
    a.label("cond_sub_n")
    a.lw(T2, 256, A1)
    a.bne(T2, ZERO, "cs_sub")
    a.li(T0, 252)
    a.la(T5, L["N"])
    a.label("cs_cmp")
    a.add(T3, A1, T0)
    a.lw(T3, 0, T3)
    a.add(T4, T5, T0)
    a.lw(T4, 0, T4)
    a.bltu(T4, T3, "cs_sub")
    a.bltu(T3, T4, "cs_done")
    a.addi(T0, T0, -4)
    a.bge(T0, ZERO, "cs_cmp")
    a.label("cs_sub")          # equal falls through here as well: x = N reduces to 0
    a.li(T0, 0)
    a.li(T6, 0)                # borrow
    a.la(T5, L["N"])
    a.label("cs_l")
    a.add(T3, A1, T0)
    a.lw(T1, 0, T3)
    a.add(T4, T5, T0)
    a.lw(T4, 0, T4)
    a.sub(T2, T1, T4)
    a.sltu(T4, T1, T4)
    a.sltu(T1, T2, T6)
    a.sub(T2, T2, T6)
    a.or_(T6, T4, T1)
    a.sw(T2, 0, T3)
    a.addi(T0, T0, 4)
    a.li(T4, 256)
    a.bne(T0, T4, "cs_l")
    a.sw(ZERO, 256, A1)
    a.label("cs_done")
    a.ret()

    # ------------------------------------------------------------------ rsa_pub: OUT = BASE^65537 mod N
    a.label("rsa_pub")
    a.mv(S7, RA)
    # n' = -N[0]^-1 mod 2^32: x <- x (2 - n0 x), five times from x = n0 (correct to 3 bits)
    a.la(T0, L["N"])
    a.lw(T1, 0, T0)
    a.mv(T2, T1)
    for _ in range(5):
        a.mul(T3, T1, T2)
        a.li(T4, 2)
        a.sub(T3, T4, T3)
        a.mul(T2, T2, T3)
    a.sub(T2, ZERO, T2)
    a.la(T0, L["N0INV"])
    a.sw(T2, 0, T0)
    # ONE_M = R mod N = 2^2048 - N (the modulus has its top bit set), X = the same with one overflow limb for the doublings
    a.la(T0, L["N"])
    a.la(T1, L["T"])
    a.li(T2, 0)
    a.li(T5, 1)                # carry of the two's complement
    a.label("rp_neg")
    a.add(T3, T0, T2)
    a.lw(T3, 0, T3)
    a.not_(T3, T3)
    a.add(T3, T3, T5)
    a.sltu(T5, T3, T5)
    a.add(T4, T1, T2)
    a.sw(T3, 0, T4)
    a.addi(T2, T2, 4)
    a.li(T6, 256)
    a.bne(T2, T6, "rp_neg")
    a.sw(ZERO, 256, T1)
    a.la(A0, L["ONE_M"])
    a.call("copy_t")
    # T <- 2^32 T mod N by 32 modular doublings
    a.li(S6, 32)
    a.label("rp_dbl")
    a.la(T1, L["T"])
    a.li(T2, 0)
    a.li(T5, 0)                # bit carried into the next limb
    a.label("rp_shl")
    a.add(T3, T1, T2)
    a.lw(T4, 0, T3)
    a.srli(T6, T4, 31)
    a.slli(T4, T4, 1)
    a.or_(T4, T4, T5)
    a.sw(T4, 0, T3)
    a.mv(T5, T6)
    a.addi(T2, T2, 4)
    a.li(T6, 256)
    a.bne(T2, T6, "rp_shl")
    a.sw(T5, 256, T1)
    a.mv(A1, T1)
    a.call("cond_sub_n")
    a.addi(S6, S6, -1)
    a.bne(S6, ZERO, "rp_dbl")
    a.la(A0, L["R2"])
    a.call("copy_t")
    # six Montgomery squarings: 2^32 R -> 2^64 R -> ... -> 2^2048 R = R^2 mod N
    a.li(S6, 6)
    a.label("rp_sq")
    a.la(A0, L["R2"])
    a.mv(A1, A0)
    a.mv(A2, A0)
    a.call("montmul")
    a.addi(S6, S6, -1)
    a.bne(S6, ZERO, "rp_sq")
    # X = BASE R, ACC = X^(2^16) X, OUT = ACC / R
    a.la(A0, L["X"])
    a.la(A1, L["BASE"])
    a.la(A2, L["R2"])
    a.call("montmul")
    a.la(A0, L["ACC"])
    a.la(A1, L["X"])
    a.la(A2, L["ONE_M"])       # X * (R mod N) / R = X
    a.call("montmul")
    a.li(S6, 16)
    a.label("rp_e")
    a.la(A0, L["ACC"])
    a.mv(A1, A0)
    a.mv(A2, A0)
    a.call("montmul")
    a.addi(S6, S6, -1)
    a.bne(S6, ZERO, "rp_e")
    a.la(A0, L["ACC"])
    a.mv(A1, A0)
    a.la(A2, L["X"])
    a.call("montmul")
    a.la(T0, L["PLAIN1"])      # the number 1 (the buffer is zero from the start; only limb 0 is ever written)
    a.li(T1, 1)
    a.sw(T1, 0, T0)
    a.la(A0, L["OUT"])
    a.la(A1, L["ACC"])
    a.mv(A2, T0)
    a.call("montmul")
    a.mv(RA, S7)
    a.ret()

    # ------------------------------------------------------------------ copy_t(a0 = dst): dst = T[0..63]
    a.label("copy_t")
    a.la(T1, L["T"])
    a.li(T0, 0)
    a.label("ct_l")
    a.add(T2, T1, T0)
    a.lw(T3, 0, T2)
    a.add(T2, A0, T0)
    a.sw(T3, 0, T2)
    a.addi(T0, T0, 4)
    a.li(T6, 256)
    a.bne(T0, T6, "ct_l")
    a.ret()


def limbs(value):
    return [(value >> (32 * i)) & 0xFFFFFFFF for i in range(64)]


def message_frame(data):
    data = bytes(data)
    padded = data + bytes(-len(data) % 4)
    return [len(data)] + list(struct.unpack("<%dI" % (len(padded) // 4), padded))


def input_stream(signed_info, bank_sig, bank_n, tx_plain, client_n, tx_cipher, order_data, witness_sig, witness_n, e=65537):
    """The guest's input words: what host/src/main.rs:389-417 feeds the real guest, reduced to what this guest checks."""
    words = message_frame(signed_info) + limbs(int.from_bytes(bank_sig, "big")) + limbs(bank_n) + [e]
    words += limbs(int.from_bytes(tx_plain, "big")) + limbs(client_n) + limbs(int.from_bytes(tx_cipher, "big")) + [e]
    words += message_frame(order_data) + limbs(int.from_bytes(witness_sig, "big")) + limbs(witness_n) + [e]
    return words


def reference_inputs():
    """The reference's own fixture (tests/golden/camt53: data/test/test.xml-* and the public keys), through the library's pre-processor."""
    sys.path.insert(0, ROOT)
    import hyperfridge_r0_amd as r0
    D = os.path.join(ROOT, "tests", "golden", "camt53")
    rd = lambda name: open(os.path.join(D, name), "rb").read()
    eb = r0.Ebics(rd("response.xml"))
    mod = lambda pem: int(r0.rsa_public_key_decimal(rd(pem))[0])
    return dict(signed_info=eb.part(r0.Ebics.SIGNED_INFO), bank_sig=eb.part(r0.Ebics.SIGNATURE_BIN), bank_n=mod("pub_bank.pem"),
                tx_plain=rd("test.xml-TransactionKeyDecrypt.bin"), client_n=mod("pub_client.pem"), tx_cipher=eb.part(r0.Ebics.TRANSACTION_KEY_BIN),
                order_data=eb.part(r0.Ebics.ORDER_DATA_BIN), witness_sig=bytes.fromhex(rd("test.xml-Witness.hex").decode().replace("\n", "").strip()),
                witness_n=mod("pub_witness.pem"))


def elf_and_input():
    image, _, _ = build()
    return image, input_stream(**reference_inputs()), ("SHA-256 + three RSA-2048 public-key operations on the reference's EBICS fixture (tools/guest_rsa.py), about 10.6 M cycles")


if __name__ == "__main__":
    image, labels, L = build()
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "circuits", "guest_rsa.elf")
    open(out, "wb").write(image)
    print("guest_rsa: %d bytes, entry %#x, text %d words -> %s" % (len(image), labels["_start"], (len(image)) // 4, out))
