#!/usr/bin/env python3
"""The hyperfridge guest's whole pipeline as a hand-assembled RV32IM program (tools/rvasm.py; the reference's guest is Rust and needs
the risc0 toolchain), on the reference's own inputs, committing the reference's own journal.

On top of tools/guest_rsa.py (SHA-256 of SignedInfo + the bank's RSA-2048 signature, the transaction-key check, the witness
signature over SHA-256 of the order data: methods/guest/src/main.rs:450-485, 663-718, 757-833) this guest does what
methods/guest/src/main.rs:719-756 (`decrypt_order_data`) and :836-981 (the camt53 parse) do and commits what :214-262 commits:

  0. hashes the authenticated part of the response (`<xml>-authenticated`), base64-encodes the digest and compares it with the
     <ds:DigestValue> inside the SignedInfo the bank signed (methods/guest/src/main.rs:556-560; test_xmlparse.rs:43-67) -- exit 10;
  5. takes the AES-128 key out of the decrypted transaction-key block (00 02 PS 00 key: the key is the last 16 bytes, the 00 at
     byte 239) and decrypts the order data -- AES-128-CBC, zero ICV (FIPS 197 inverse cipher, byte-oriented, tables for the
     S-boxes and the GF(2^8) products by 9, 11, 13, 14) -- exit 9 on a malformed block or padding;
  6. inflates the zlib stream (RFC 1950 header and Adler-32, RFC 1951 stored / fixed / dynamic blocks, canonical Huffman decoding
     a bit at a time) into the ZIP archive it holds -- exit 6 on a malformed stream;
  7. walks the archive's local file headers, inflates every member (raw deflate; stored members are copied) and checks its
     CRC-32 against the header's -- exit 7 (malformed archive) / 11 (checksum);
  8. in every camt.053 document whose statement account `<Acct><Id><IBAN>` is the IBAN the host named, reads ElctrncSeqNb,
     FrDtTm, ToDtTm and the first balance's Amt / Ccy / Cd, and
  9. commits {"hostinfo":..,"iban":..,"pub_bank_pem":..,"pub_witness_pem":..,"pub_client_pem":..,"stmts":[{"elctrnc_seq_nb":..,
     "fr_dt_tm":..,"to_dt_tm":..,"amt":..,"ccy":..,"cd":..},..]} as a serde-framed string and halts with 0 -- exit 8 when no
     document matches.  The three keys are the moduli the RSA checks used, re-encoded by the guest as SubjectPublicKeyInfo PEM
     (exponent 65537, DER, base64 in lines of 64, line ends escaped as a backslash and an n), as main.rs:237-243 does.

The reference holds two committed receipts for this fixture and the program's journal IS their `journal.bytes`, byte for byte
(tests/test_guest_camt53.py): data/test/test.xml-Receipt-6bb958..-latest.json in the current form above (commitment form 1), and
data/test/test.xml-Receipt-test.json, written before the guest began to commit the keys (form 0: hostinfo, iban, stmts; the form
is the last word of the input).  Not reproduced: the XML tokenizer (tags are located by substring search), the
parse of the pre-processed EbicsResponse snippets, and the RSA PRIVATE-key decryption of the transaction key,
which the reference's guest can also be told to skip by handing it the decrypted block (`decrypted_tx_key_bin`, as here)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import guest_rsa  # noqa: E402
from rvasm import A0, A1, A2, A3, A4, A5, A6, A7, RA, S0, S1, S2, S3, S4, S5, S6, S7, S8, S9, S10, S11, SP, T0, T1, T2, T3, T4, T5, T6, ZERO  # noqa: E402

STACK_TOP = 0x7FFF0
ZIP_MAX, DOC_MAX, OUT_MAX = 16384, 16384, 4096


def _aes_tables():
    sbox, p, q = [0] * 256, 1, 1
    while True:  # the multiplicative inverse through 3 as generator (FIPS 197 section 5.1.1), then the affine map
        p = p ^ ((p << 1) & 0xFF) ^ (0x1B if p & 0x80 else 0)
        q ^= q << 1
        q ^= q << 2
        q ^= q << 4
        q &= 0xFF
        if q & 0x80:
            q ^= 0x09
        x = q ^ ((q << 1) | (q >> 7)) & 0xFF ^ ((q << 2) | (q >> 6)) & 0xFF ^ ((q << 3) | (q >> 5)) & 0xFF ^ ((q << 4) | (q >> 4)) & 0xFF
        sbox[p] = (x ^ 0x63) & 0xFF
        if p == 1:
            break
    sbox[0] = 0x63
    inv = [0] * 256
    for i, v in enumerate(sbox):
        inv[v] = i

    def xt(v):
        return ((v << 1) ^ (0x1B if v & 0x80 else 0)) & 0xFF

    def mul(v, k):
        r = 0
        while k:
            if k & 1:
                r ^= v
            v, k = xt(v), k >> 1
        return r
    # InvShiftRows on the column-major state: new[r + 4 c] = old[r + 4 ((c - r) mod 4)]
    perm = [r + 4 * ((c - r) % 4) for c in range(4) for r in range(4)]
    rcon, v = [], 1
    for _ in range(10):
        rcon.append(v)
        v = xt(v)
    return bytes(sbox), bytes(inv), {k: bytes(mul(v, k) for v in range(256)) for k in (9, 11, 13, 14)}, bytes(perm), bytes(rcon)


LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
STRINGS = {  # needles in the camt.053 documents and the pieces of the commitment
    "N_IBAN": b"<Acct><Id><IBAN>", "N_IBAN_END": b"</IBAN>", "N_SEQ": b"<ElctrncSeqNb>", "N_FR": b"<FrDtTm>", "N_TO": b"<ToDtTm>", "N_BAL": b"<Bal>",
    "N_CD": b"<Cd>", "N_AMT": b'<Amt Ccy="', "N_DIGEST": b"<ds:DigestValue>",
    "J_HOST": b'{"hostinfo":"', "J_IBAN": b'","iban":"', "J_STMTS": b'","stmts":[', "J_BANK": b'","pub_bank_pem":"', "J_WITNESS": b'","pub_witness_pem":"',
    "J_CLIENT": b'","pub_client_pem":"', "PEM_BEGIN": b"-----BEGIN PUBLIC KEY-----\\n", "PEM_END": b"-----END PUBLIC KEY-----\\n", "ESC_NL": b"\\n",
    "B64": b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/",
    # SubjectPublicKeyInfo of a 2048-bit RSA key with exponent 65537: SEQUENCE { SEQUENCE { rsaEncryption, NULL }, BIT STRING { SEQUENCE { INTEGER n, INTEGER e } } }
    "DER_HEAD": bytes.fromhex("30820122300d06092a864886f70d01010105000382010f003082010a0282010100"), "DER_TAIL": bytes.fromhex("0203010001"), "J_SEQ": b'{"elctrnc_seq_nb":"', "J_FR": b'","fr_dt_tm":"',
    "J_TO": b'","to_dt_tm":"', "J_AMT": b'","amt":"', "J_CCY": b'","ccy":"', "J_CD": b'","cd":"', "J_STMT_END": b'"}', "J_END": b"]}", "J_COMMA": b",",
}


class Camt53:
    """The extension hooks of guest_rsa.build()."""

    def layout(self, L):
        sbox, inv, muls, perm, rcon = _aes_tables()
        L.const("SBOX", sbox)
        L.const("INV_SBOX", inv)
        for k in (9, 11, 13, 14):
            L.const("MUL%d" % k, muls[k])
        L.const("ISR_PERM", perm)
        L.const("RCON", rcon)
        for name, table in (("LBASE", LBASE), ("LEXT", LEXT), ("DBASE", DBASE), ("DEXT", DEXT), ("ORDER", ORDER)):
            L.const(name, struct.pack("<%dI" % len(table), *table))
        for name, text in STRINGS.items():
            L.const(name, text)
        crc = []
        for n in range(256):  # CRC-32 (IEEE 802.3, reflected, polynomial 0xEDB88320): the table of the byte-at-a-time form
            c = n
            for _ in range(8):
                c = (c >> 1) ^ (0xEDB88320 if c & 1 else 0)
            crc.append(c)
        L.const("CRC_TABLE", struct.pack("<256I", *crc))
        for name, size in (("AUTH_LEN", 4), ("AUTH", guest_rsa.MAX_MSG + 128), ("DIGEST3", 32), ("B64OUT", 48), ("N_BANK", 256), ("N_CLIENT", 256), ("N_WITNESS", 256), ("DERBUF", 296), ("FORM", 4), ("IBAN_LEN", 4), ("IBAN", 64), ("HOST_LEN", 4), ("HOST", 256), ("RK", 176), ("ST", 16), ("TMP16", 16), ("KEY", 16),
                           ("PLAIN", guest_rsa.MAX_MSG + 32), ("LENCODE", 64 + 4 * 288), ("DISTCODE", 64 + 4 * 32), ("CLCODE", 64 + 4 * 19), ("OFFS", 64),
                           ("CL_LENGTHS", 4 * 19), ("LENGTHS", 4 * 320), ("ZIPBUF", ZIP_MAX), ("DOC", DOC_MAX), ("JP", 4), ("N_STMTS", 4),
                           ("TXBLOCK", 256)):
            L.var(name, size)
        L.fixed("JOUT", guest_rsa.JOURNAL_BASE)  # the commitment is put together in the journal window: COMMIT names its words there
        assert L.bss_top < STACK_TOP - 4096

    def after_signed_info(self, a, L, fresh, halt):
        """step 0: SHA-256 of the authenticated part, base64, must be the <ds:DigestValue> of the SignedInfo in MSG"""
        a.li(SP, STACK_TOP)
        a.la(A0, L["AUTH_LEN"])
        a.li(A1, 1)
        a.li(A7, 1)
        a.ecall()
        a.la(T0, L["AUTH_LEN"])
        a.lw(S0, 0, T0)
        a.li(T1, guest_rsa.MAX_MSG)
        ok = fresh("auth_len_ok")
        a.bgeu(T1, S0, ok)
        halt(5)
        a.label(ok)
        a.addi(A1, S0, 3)
        a.srli(A1, A1, 2)
        a.la(A0, L["AUTH"])
        a.li(A7, 1)
        a.ecall()
        a.la(A0, L["AUTH"])
        a.la(T0, L["AUTH_LEN"])
        a.lw(A1, 0, T0)
        a.la(A2, L["DIGEST3"])
        a.call("sha256")
        a.la(A0, L["DIGEST3"])
        a.la(A1, L["B64OUT"])
        a.call("b64_digest")
        a.la(A0, L["MSG"])           # SignedInfo
        a.la(T0, L["LEN"])
        a.lw(A1, 0, T0)
        a.add(A1, A0, A1)
        a.mv(S0, A1)
        a.la(A2, L["N_DIGEST"])
        a.li(A3, len(STRINGS["N_DIGEST"]))
        a.call("find")
        bad = fresh("digest_bad")
        a.beq(A0, ZERO, bad)
        a.addi(T0, A0, 45)           # 44 characters and the '<' of the closing tag
        a.bltu(S0, T0, bad)
        a.la(T1, L["B64OUT"])
        a.li(T2, 0)
        a.label("digest_cmp")
        a.add(T3, A0, T2)
        a.lbu(T3, 0, T3)
        a.add(T4, T1, T2)
        a.lbu(T4, 0, T4)
        a.bne(T3, T4, bad)
        a.addi(T2, T2, 1)
        a.li(T5, 44)
        a.bne(T2, T5, "digest_cmp")
        a.lbu(T3, 44, A0)
        a.li(T4, ord("<"))
        a.bne(T3, T4, bad)
        ok = fresh("digest_ok")
        a.j(ok)
        a.label(bad)
        halt(10)
        a.label(ok)

    def after_operands(self, a, L, k):  # the modulus of RSA operation k is kept: the commitment names the three keys
        a.la(T0, L["N"])
        a.la(T1, L[("N_BANK", "N_CLIENT", "N_WITNESS")[k]])
        a.addi(T2, T0, 256)
        a.label("keep_n_%d" % k)
        a.lw(T3, 0, T0)
        a.sw(T3, 0, T1)
        a.addi(T0, T0, 4)
        a.addi(T1, T1, 4)
        a.bne(T0, T2, "keep_n_%d" % k)

    # ------------------------------------------------------------------ the main program after the three RSA checks
    def main(self, a, L, fresh, halt):
        def read_words(sym, count_reg=None, count=None):
            a.la(A0, L[sym])
            if count is not None:
                a.li(A1, count)
            else:
                a.mv(A1, count_reg)
            a.li(A7, 1)
            a.ecall()

        def read_string(len_sym, buf_sym, max_len):  # [byte length][bytes]
            read_words(len_sym, count=1)
            a.la(T0, L[len_sym])
            a.lw(S0, 0, T0)
            a.li(T1, max_len)
            ok = fresh("str_ok")
            a.bgeu(T1, S0, ok)
            halt(5)
            a.label(ok)
            a.addi(T0, S0, 3)
            a.srli(T0, T0, 2)
            read_words(buf_sym, count_reg=T0)

        a.li(SP, STACK_TOP)
        read_string("IBAN_LEN", "IBAN", 60)
        read_string("HOST_LEN", "HOST", 250)
        read_words("FORM", count=1)
        # 5. the AES key: bytes 240..255 of the decrypted transaction-key block 00 02 PS 00 key (kept by step 2 as 64 limbs in TXBLOCK)
        a.call("aes_key_from_block")
        a.la(A0, L["KEY"])
        a.call("aes_key_expand")
        a.la(T0, L["LEN"])           # the order data is still in MSG (step 3 hashed it there); its length in LEN
        a.lw(S0, 0, T0)
        a.andi(T1, S0, 15)
        bad = fresh("bad_cipher")
        a.bne(T1, ZERO, bad)
        a.beq(S0, ZERO, bad)
        a.li(S1, 0)                  # offset of the block
        a.label("cbc_loop")
        a.la(A0, L["MSG"])
        a.add(A0, A0, S1)
        a.la(A1, L["PLAIN"])
        a.add(A1, A1, S1)
        a.call("aes_decrypt_block")
        a.beq(S1, ZERO, "cbc_next")  # zero ICV: the first block is the cipher's output itself
        a.la(T0, L["MSG"])
        a.add(T0, T0, S1)
        a.la(T1, L["PLAIN"])
        a.add(T1, T1, S1)
        a.li(T2, 0)
        a.label("cbc_xor")
        a.add(T3, T0, T2)
        a.lbu(T4, -16, T3)           # the previous ciphertext block
        a.add(T3, T1, T2)
        a.lbu(T5, 0, T3)
        a.xor(T5, T5, T4)
        a.sb(T5, 0, T3)
        a.addi(T2, T2, 1)
        a.li(T6, 16)
        a.bne(T2, T6, "cbc_xor")
        a.label("cbc_next")
        a.addi(S1, S1, 16)
        a.bne(S1, S0, "cbc_loop")
        a.la(T0, L["PLAIN"])         # padding (ANSI X9.23 / ISO 10126-2): the last byte counts the padding bytes, 1..16
        a.add(T0, T0, S0)
        a.lbu(T1, -1, T0)
        a.beq(T1, ZERO, bad)
        a.li(T2, 16)
        a.bltu(T2, T1, bad)
        a.sub(S0, S0, T1)            # length of the zlib stream
        ok = fresh("cipher_ok")
        a.j(ok)
        a.label(bad)
        halt(9)
        a.label(ok)
        # 6. zlib -> the ZIP archive
        a.la(A0, L["PLAIN"])
        a.add(A1, A0, S0)
        a.la(A2, L["ZIPBUF"])
        a.li(A3, ZIP_MAX)
        a.add(A3, A2, A3)
        a.call("zlib_inflate")
        a.mv(S1, A0)                 # end of the archive
        # the commitment opens with hostinfo and iban
        a.la(T0, L["JOUT"] + 4)
        a.la(T1, L["JP"])
        a.sw(T0, 0, T1)
        self.append_const(a, L, "J_HOST")
        self.append_var(a, L, "HOST", "HOST_LEN")
        self.append_const(a, L, "J_IBAN")
        self.append_var(a, L, "IBAN", "IBAN_LEN")
        a.la(T0, L["FORM"])
        a.lw(T0, 0, T0)
        a.beq(T0, ZERO, "no_keys")   # commitment form 0: the earlier one, without the keys
        for piece, modulus in (("J_BANK", "N_BANK"), ("J_WITNESS", "N_WITNESS"), ("J_CLIENT", "N_CLIENT")):
            self.append_const(a, L, piece)
            a.la(A0, L[modulus])
            a.call("append_pem")
        a.label("no_keys")
        self.append_const(a, L, "J_STMTS")
        # 7. every member of the archive
        a.la(S0, L["ZIPBUF"])
        a.label("zip_loop")
        a.addi(T0, S0, 30)
        a.bltu(S1, T0, "zip_done")
        a.mv(A0, S0)
        a.call("ldu32")
        a.li(T0, 0x04034B50)
        a.bne(A0, T0, "zip_done")    # the central directory follows the last member
        a.addi(A0, S0, 8)
        a.call("ldu16")
        a.mv(S2, A0)                 # method
        a.addi(A0, S0, 18)
        a.call("ldu32")
        a.mv(S3, A0)                 # compressed size
        a.addi(A0, S0, 22)
        a.call("ldu32")
        a.mv(S4, A0)                 # size
        a.addi(A0, S0, 26)
        a.call("ldu16")
        a.mv(S5, A0)
        a.addi(A0, S0, 28)
        a.call("ldu16")
        a.add(S5, S5, A0)
        a.addi(S5, S5, 30)
        a.add(S5, S0, S5)            # the member's data
        a.add(S6, S5, S3)            # ... and its end = the next header
        zbad = fresh("zip_bad")
        a.bltu(S1, S6, zbad)
        a.li(T0, DOC_MAX)
        a.bltu(T0, S4, zbad)
        a.li(T0, 8)
        a.beq(S2, T0, "zip_deflated")
        a.bne(S2, ZERO, zbad)
        a.bne(S3, S4, zbad)          # stored: copied as it is
        a.la(T1, L["DOC"])
        a.mv(T2, S5)
        a.label("zip_copy")
        a.beq(T2, S6, "zip_copied")
        a.lbu(T3, 0, T2)
        a.sb(T3, 0, T1)
        a.addi(T1, T1, 1)
        a.addi(T2, T2, 1)
        a.j("zip_copy")
        a.label("zip_copied")
        a.mv(A0, T1)
        a.j("zip_have")
        a.label("zip_deflated")
        # the inflate routine keeps its state in s2..s9: this loop's live registers go on the stack around it
        a.addi(SP, SP, -16)
        a.sw(S0, 0, SP)
        a.sw(S1, 4, SP)
        a.sw(S4, 8, SP)
        a.sw(S6, 12, SP)
        a.mv(A0, S5)
        a.mv(A1, S6)
        a.la(A2, L["DOC"])
        a.li(A3, DOC_MAX)
        a.add(A3, A2, A3)
        a.call("inflate")
        a.lw(S0, 0, SP)
        a.lw(S1, 4, SP)
        a.lw(S4, 8, SP)
        a.lw(S6, 12, SP)
        a.addi(SP, SP, 16)
        a.label("zip_have")
        a.la(T0, L["DOC"])
        a.sub(T1, A0, T0)
        a.bne(T1, S4, zbad)          # the size the header announced
        a.addi(SP, SP, -16)
        a.sw(S0, 0, SP)
        a.sw(S1, 4, SP)
        a.sw(S6, 8, SP)
        a.sw(A0, 12, SP)             # end of the document
        a.mv(A1, A0)
        a.mv(A0, T0)
        a.call("crc32")              # ... and the checksum it announced
        a.mv(S2, A0)
        a.lw(S0, 0, SP)
        a.addi(A0, S0, 14)
        a.call("ldu32")
        crc_ok = fresh("crc_ok")
        a.beq(A0, S2, crc_ok)
        halt(11)
        a.label(crc_ok)
        a.la(A0, L["DOC"])
        a.lw(A1, 12, SP)
        a.call("statement")
        a.lw(S0, 0, SP)
        a.lw(S1, 4, SP)
        a.lw(S6, 8, SP)
        a.addi(SP, SP, 16)
        a.mv(S0, S6)
        a.j("zip_loop")
        a.label(zbad)
        halt(7)
        a.label("zip_done")
        a.la(T0, L["N_STMTS"])
        a.lw(T0, 0, T0)
        some = fresh("some_stmt")
        a.bne(T0, ZERO, some)
        halt(8)
        a.label(some)
        # 9. close the JSON, frame it, commit
        self.append_const(a, L, "J_END")
        a.la(T0, L["JP"])
        a.lw(T0, 0, T0)
        a.la(T1, L["JOUT"])
        a.sub(T2, T0, T1)
        a.addi(T2, T2, -4)           # length of the string
        a.sw(T2, 0, T1)
        a.addi(T2, T2, 7)            # 4 bytes of length + the string, zero-padded to a word boundary
        a.srli(A1, T2, 2)            # COMMIT takes words
        a.mv(A0, T1)
        a.li(A7, 2)
        a.ecall()
        halt(0)
        return True

    @staticmethod
    def append_const(a, L, name):
        a.la(A0, L[name])
        a.li(A1, len(STRINGS[name]))
        a.call("append")

    @staticmethod
    def append_var(a, L, buf, len_sym):
        a.la(A0, L[buf])
        a.la(A1, L[len_sym])
        a.lw(A1, 0, A1)
        a.call("append")

    # ------------------------------------------------------------------ the routines
    def library(self, a, L, fresh):
        def prologue(*saved):
            a.addi(SP, SP, -4 * (1 + len(saved)))
            a.sw(RA, 0, SP)
            for k, r in enumerate(saved):
                a.sw(r, 4 * (k + 1), SP)

        def epilogue(*saved):
            a.lw(RA, 0, SP)
            for k, r in enumerate(saved):
                a.lw(r, 4 * (k + 1), SP)
            a.addi(SP, SP, 4 * (1 + len(saved)))
            a.ret()

        def die(code):
            a.li(A0, code)
            a.li(A7, 0)
            a.ecall()

        # ---- append(a0 = bytes, a1 = count): to the commitment under construction
        a.label("append")
        a.la(T0, L["JP"])
        a.lw(T1, 0, T0)
        a.la(T2, L["JOUT"] + OUT_MAX - 8)
        a.add(T3, T1, A1)
        ok = fresh("app_ok")
        a.bgeu(T2, T3, ok)
        die(7)
        a.label(ok)
        a.add(T3, A0, A1)
        a.label("app_l")
        a.beq(A0, T3, "app_d")
        a.lbu(T4, 0, A0)
        a.sb(T4, 0, T1)
        a.addi(A0, A0, 1)
        a.addi(T1, T1, 1)
        a.j("app_l")
        a.label("app_d")
        a.sw(T1, 0, T0)
        a.ret()

        # ---- append_pem(a0 = modulus, 64 little-endian limbs): the key as PEM text with escaped line ends, as the reference's guest commits it
        a.label("append_pem")
        prologue(S0, S1, S2)
        a.mv(S0, A0)
        a.la(T0, L["DERBUF"])        # DER: the fixed head, the modulus most significant byte first, the exponent
        a.la(T1, L["DER_HEAD"])
        a.li(T2, len(STRINGS["DER_HEAD"]))
        a.add(T2, T1, T2)
        a.label("pem_head")
        a.lbu(T3, 0, T1)
        a.sb(T3, 0, T0)
        a.addi(T0, T0, 1)
        a.addi(T1, T1, 1)
        a.bne(T1, T2, "pem_head")
        a.li(T1, 255)                # byte b of the modulus = limb (255 - b) >> 2 shifted by 8 ((255 - b) & 3)
        a.label("pem_mod")
        a.srli(T2, T1, 2)
        a.slli(T2, T2, 2)
        a.add(T2, S0, T2)
        a.lw(T2, 0, T2)
        a.andi(T3, T1, 3)
        a.slli(T3, T3, 3)
        a.srl(T2, T2, T3)
        a.sb(T2, 0, T0)
        a.addi(T0, T0, 1)
        a.addi(T1, T1, -1)
        a.bge(T1, ZERO, "pem_mod")
        a.la(T1, L["DER_TAIL"])
        a.li(T2, len(STRINGS["DER_TAIL"]))
        a.add(T2, T1, T2)
        a.label("pem_tail")
        a.lbu(T3, 0, T1)
        a.sb(T3, 0, T0)
        a.addi(T0, T0, 1)
        a.addi(T1, T1, 1)
        a.bne(T1, T2, "pem_tail")
        self.append_const(a, L, "PEM_BEGIN")
        a.la(S0, L["DERBUF"])        # base64, three bytes at a time (294 = 98 x 3: no padding), a line end every 64 characters
        a.li(S1, 0)                  # characters on the current line
        a.li(S2, 294)
        a.add(S2, S0, S2)
        a.label("pem_b64")
        a.lbu(T0, 0, S0)
        a.lbu(T1, 1, S0)
        a.lbu(T2, 2, S0)
        a.slli(T0, T0, 16)
        a.slli(T1, T1, 8)
        a.or_(T0, T0, T1)
        a.or_(T0, T0, T2)
        a.la(T4, L["B64"])
        a.la(T5, L["TMP16"])
        for k, shift in enumerate((18, 12, 6, 0)):
            a.srli(T1, T0, shift)
            a.andi(T1, T1, 63)
            a.add(T1, T4, T1)
            a.lbu(T1, 0, T1)
            a.sb(T1, k, T5)
        a.mv(A0, T5)
        a.li(A1, 4)
        a.call("append")
        a.addi(S0, S0, 3)
        a.addi(S1, S1, 4)
        a.li(T0, 64)
        a.bne(S1, T0, "pem_same_line")
        self.append_const(a, L, "ESC_NL")
        a.li(S1, 0)
        a.label("pem_same_line")
        a.bne(S0, S2, "pem_b64")
        a.beq(S1, ZERO, "pem_closed")
        self.append_const(a, L, "ESC_NL")
        a.label("pem_closed")
        self.append_const(a, L, "PEM_END")
        epilogue(S0, S1, S2)

        # ---- crc32(a0 = bytes, a1 = end) -> a0 (IEEE 802.3, as ZIP stores it)
        a.label("crc32")
        a.li(T0, -1)
        a.la(T1, L["CRC_TABLE"])
        a.label("crc_l")
        a.beq(A0, A1, "crc_d")
        a.lbu(T2, 0, A0)
        a.xor(T2, T2, T0)
        a.andi(T2, T2, 0xFF)
        a.slli(T2, T2, 2)
        a.add(T2, T1, T2)
        a.lw(T2, 0, T2)
        a.srli(T0, T0, 8)
        a.xor(T0, T0, T2)
        a.addi(A0, A0, 1)
        a.j("crc_l")
        a.label("crc_d")
        a.not_(A0, T0)
        a.ret()

        # ---- b64_digest(a0 = 8 digest words, a1 = 44 characters out): base64 of the 32 digest bytes (ten triples, then two bytes and '=')
        a.label("b64_digest")
        a.la(T0, L["TMP16"])         # the bytes, most significant byte of each word first; TMP16 and the 16 bytes of KEY behind it
        a.li(T1, 0)
        a.label("b64d_bytes")
        a.add(T2, A0, T1)
        a.lw(T2, 0, T2)
        a.add(T3, T0, T1)
        for k, sh in enumerate((24, 16, 8, 0)):
            a.srli(T4, T2, sh)
            a.sb(T4, k, T3)
        a.addi(T1, T1, 4)
        a.li(T6, 32)
        a.bne(T1, T6, "b64d_bytes")
        a.la(T5, L["B64"])
        a.li(T1, 0)
        a.label("b64d_triple")
        a.add(T2, T0, T1)
        a.lbu(T3, 0, T2)
        a.lbu(T4, 1, T2)
        a.slli(T3, T3, 16)
        a.slli(T4, T4, 8)
        a.or_(T3, T3, T4)
        a.li(T6, 30)
        a.beq(T1, T6, "b64d_tail")   # the last group has two bytes: three characters and the pad
        a.lbu(T4, 2, T2)
        a.or_(T3, T3, T4)
        for sh in (18, 12, 6, 0):
            a.srli(T4, T3, sh)
            a.andi(T4, T4, 63)
            a.add(T4, T5, T4)
            a.lbu(T4, 0, T4)
            a.sb(T4, 0, A1)
            a.addi(A1, A1, 1)
        a.addi(T1, T1, 3)
        a.j("b64d_triple")
        a.label("b64d_tail")
        for sh in (18, 12, 6):
            a.srli(T4, T3, sh)
            a.andi(T4, T4, 63)
            a.add(T4, T5, T4)
            a.lbu(T4, 0, T4)
            a.sb(T4, 0, A1)
            a.addi(A1, A1, 1)
        a.li(T4, ord("="))
        a.sb(T4, 0, A1)
        a.ret()

        # ---- ldu16 / ldu32(a0 = any address) -> a0: little-endian loads byte by byte
        a.label("ldu16")
        a.lbu(T0, 0, A0)
        a.lbu(T1, 1, A0)
        a.slli(T1, T1, 8)
        a.or_(A0, T0, T1)
        a.ret()
        a.label("ldu32")
        a.lbu(T0, 0, A0)
        a.lbu(T1, 1, A0)
        a.slli(T1, T1, 8)
        a.or_(T0, T0, T1)
        a.lbu(T1, 2, A0)
        a.slli(T1, T1, 16)
        a.or_(T0, T0, T1)
        a.lbu(T1, 3, A0)
        a.slli(T1, T1, 24)
        a.or_(A0, T0, T1)
        a.ret()

        # ---- find(a0 = text, a1 = end, a2 = needle, a3 = its length) -> a0 = just past the first occurrence, or 0
        a.label("find")
        a.sub(T0, A1, A3)            # last start that still fits
        a.label("find_at")
        a.bltu(T0, A0, "find_no")
        a.li(T1, 0)
        a.label("find_cmp")
        a.beq(T1, A3, "find_yes")
        a.add(T2, A0, T1)
        a.lbu(T2, 0, T2)
        a.add(T3, A2, T1)
        a.lbu(T3, 0, T3)
        a.bne(T2, T3, "find_next")
        a.addi(T1, T1, 1)
        a.j("find_cmp")
        a.label("find_next")
        a.addi(A0, A0, 1)
        a.j("find_at")
        a.label("find_yes")
        a.add(A0, A0, A3)
        a.ret()
        a.label("find_no")
        a.li(A0, 0)
        a.ret()

        # ---- field(a0 = from, a1 = end, a2 = needle, a3 = its length, a4 = delimiter): append the text between the needle and the
        # delimiter; a0 = the position of the delimiter.  A document without the field ends the run (exit 7).
        a.label("field")
        prologue(S0, S1)
        a.mv(S0, A1)
        a.mv(S1, A4)
        a.call("find")
        fbad = fresh("field_bad")
        a.beq(A0, ZERO, fbad)
        a.mv(T5, A0)                 # start of the text
        a.label("field_scan")
        a.bgeu(A0, S0, fbad)
        a.lbu(T0, 0, A0)
        a.beq(T0, S1, "field_end")
        a.addi(A0, A0, 1)
        a.j("field_scan")
        a.label("field_end")
        a.addi(SP, SP, -4)
        a.sw(A0, 0, SP)
        a.sub(A1, A0, T5)
        a.mv(A0, T5)
        a.call("append")
        a.lw(A0, 0, SP)
        a.addi(SP, SP, 4)
        epilogue(S0, S1)
        a.label(fbad)
        die(7)

        # ---- statement(a0 = document, a1 = its end): if the statement's account is the IBAN asked for, append its commitment
        a.label("statement")
        prologue(S0, S1, S2)
        a.mv(S0, A0)
        a.mv(S1, A1)
        a.la(A2, L["N_IBAN"])
        a.li(A3, len(STRINGS["N_IBAN"]))
        a.call("find")
        a.beq(A0, ZERO, "stmt_skip")
        a.la(T0, L["IBAN_LEN"])
        a.lw(T0, 0, T0)
        a.add(T1, A0, T0)
        a.addi(T2, T1, len(STRINGS["N_IBAN_END"]))
        a.bltu(S1, T2, "stmt_skip")
        a.la(T2, L["IBAN"])
        a.label("stmt_cmp")          # the account, character by character ...
        a.beq(A0, T1, "stmt_tail")
        a.lbu(T3, 0, A0)
        a.lbu(T4, 0, T2)
        a.bne(T3, T4, "stmt_skip")
        a.addi(A0, A0, 1)
        a.addi(T2, T2, 1)
        a.j("stmt_cmp")
        a.label("stmt_tail")         # ... and the closing tag right behind it
        a.la(T2, L["N_IBAN_END"])
        a.li(T5, len(STRINGS["N_IBAN_END"]))
        a.add(T5, A0, T5)
        a.label("stmt_cmp2")
        a.beq(A0, T5, "stmt_mine")
        a.lbu(T3, 0, A0)
        a.lbu(T4, 0, T2)
        a.bne(T3, T4, "stmt_skip")
        a.addi(A0, A0, 1)
        a.addi(T2, T2, 1)
        a.j("stmt_cmp2")
        a.label("stmt_mine")
        a.la(T0, L["N_STMTS"])
        a.lw(T1, 0, T0)
        a.addi(T2, T1, 1)
        a.sw(T2, 0, T0)
        a.beq(T1, ZERO, "stmt_first")
        self.append_const(a, L, "J_COMMA")
        a.label("stmt_first")

        def field(json_piece, start_reg, needle, delim):
            self.append_const(a, L, json_piece)
            a.mv(A0, start_reg)
            a.mv(A1, S1)
            a.la(A2, L[needle])
            a.li(A3, len(STRINGS[needle]))
            a.li(A4, delim)
            a.call("field")

        field("J_SEQ", S0, "N_SEQ", ord("<"))
        field("J_FR", S0, "N_FR", ord("<"))
        field("J_TO", S0, "N_TO", ord("<"))
        a.mv(A0, S0)                 # the first balance
        a.mv(A1, S1)
        a.la(A2, L["N_BAL"])
        a.li(A3, len(STRINGS["N_BAL"]))
        a.call("find")
        sbad = fresh("stmt_bad")
        a.beq(A0, ZERO, sbad)
        a.mv(S2, A0)
        # <Amt Ccy="CHF">31709.14</Amt>: the currency up to the quote, then the amount behind the '>' that follows it
        self.append_const(a, L, "J_AMT")
        a.mv(A0, S2)
        a.mv(A1, S1)
        a.la(A2, L["N_AMT"])
        a.li(A3, len(STRINGS["N_AMT"]))
        a.call("find")
        a.beq(A0, ZERO, sbad)
        a.label("stmt_gt")
        a.bgeu(A0, S1, sbad)
        a.lbu(T0, 0, A0)
        a.addi(A0, A0, 1)
        a.li(T1, ord(">"))
        a.bne(T0, T1, "stmt_gt")
        a.addi(SP, SP, -4)
        a.sw(A0, 0, SP)              # where the amount starts
        a.li(T1, ord("<"))
        a.label("stmt_amt_end")
        a.bgeu(A0, S1, sbad)
        a.lbu(T0, 0, A0)
        a.beq(T0, T1, "stmt_amt_have")
        a.addi(A0, A0, 1)
        a.j("stmt_amt_end")
        a.label("stmt_amt_have")
        a.lw(T2, 0, SP)
        a.addi(SP, SP, 4)
        a.sub(A1, A0, T2)
        a.mv(A0, T2)
        a.call("append")
        field("J_CCY", S2, "N_AMT", ord('"'))
        field("J_CD", S2, "N_CD", ord("<"))
        self.append_const(a, L, "J_STMT_END")
        a.label("stmt_skip")
        epilogue(S0, S1, S2)
        a.label(sbad)
        die(7)

        # ---- aes_key_from_block: TXBLOCK holds the decrypted transaction-key block as 64 little-endian limbs (limb j = big-endian
        # bytes 252 - 4 j .. 255 - 4 j): 00 02 PS 00 key with the key in the last 16 bytes
        a.label("aes_key_from_block")
        a.la(T0, L["TXBLOCK"])
        a.lw(T1, 252, T0)            # bytes 0..3: 00 02 ..
        a.srli(T1, T1, 16)
        a.li(T2, 2)
        kbad = fresh("key_bad")
        a.bne(T1, T2, kbad)
        a.lw(T1, 16, T0)             # bytes 236..239: the 00 that ends the padding is byte 239
        a.andi(T1, T1, 0xFF)
        a.bne(T1, ZERO, kbad)
        a.la(T3, L["KEY"])
        a.li(T4, 0)                  # key byte k = byte 240 + k = limb (15 - k) >> 2, shifted by 8 ((15 - k) & 3)
        a.label("key_l")
        a.li(T5, 15)
        a.sub(T5, T5, T4)
        a.srli(T6, T5, 2)
        a.slli(T6, T6, 2)
        a.add(T6, T0, T6)
        a.lw(T6, 0, T6)
        a.andi(T5, T5, 3)
        a.slli(T5, T5, 3)
        a.srl(T6, T6, T5)
        a.add(T1, T3, T4)
        a.sb(T6, 0, T1)
        a.addi(T4, T4, 1)
        a.li(T5, 16)
        a.bne(T4, T5, "key_l")
        a.ret()
        a.label(kbad)
        die(9)

        # ---- aes_key_expand(a0 = 16 key bytes): RK = the eleven round keys (FIPS 197 section 5.2), as bytes
        a.label("aes_key_expand")
        a.la(T0, L["RK"])
        a.li(T1, 0)
        a.label("kx_copy")
        a.add(T2, A0, T1)
        a.lbu(T3, 0, T2)
        a.add(T2, T0, T1)
        a.sb(T3, 0, T2)
        a.addi(T1, T1, 1)
        a.li(T6, 16)
        a.bne(T1, T6, "kx_copy")
        a.la(A1, L["SBOX"])
        a.la(A2, L["RCON"])
        a.li(T1, 16)                 # position of the word being made
        a.label("kx_word")
        a.add(T2, T0, T1)
        a.lbu(A3, -4, T2)
        a.lbu(A4, -3, T2)
        a.lbu(A5, -2, T2)
        a.lbu(A6, -1, T2)
        a.andi(T3, T1, 15)
        a.bne(T3, ZERO, "kx_plain")
        a.add(T3, A1, A4)            # SubWord(RotWord(t)) xor Rcon
        a.lbu(T3, 0, T3)
        a.add(T4, A1, A5)
        a.lbu(T4, 0, T4)
        a.add(T5, A1, A6)
        a.lbu(T5, 0, T5)
        a.add(T6, A1, A3)
        a.lbu(T6, 0, T6)
        a.lbu(A3, 0, A2)
        a.addi(A2, A2, 1)
        a.xor(A3, A3, T3)
        a.mv(A4, T4)
        a.mv(A5, T5)
        a.mv(A6, T6)
        a.label("kx_plain")
        for k, r in enumerate((A3, A4, A5, A6)):
            a.lbu(T3, -16 + k, T2)
            a.xor(T3, T3, r)
            a.sb(T3, k, T2)
        a.addi(T1, T1, 4)
        a.li(T6, 176)
        a.bne(T1, T6, "kx_word")
        a.ret()

        # ---- aes_decrypt_block(a0 = 16 bytes in, a1 = 16 bytes out): the inverse cipher of FIPS 197 section 5.3
        a.label("aes_decrypt_block")
        a.la(T0, L["RK"])
        a.la(T1, L["ST"])
        a.li(T2, 0)
        a.label("ad_first")          # AddRoundKey with the last round key
        a.add(T3, A0, T2)
        a.lbu(T4, 0, T3)
        a.add(T3, T0, T2)
        a.lbu(T5, 160, T3)
        a.xor(T4, T4, T5)
        a.add(T3, T1, T2)
        a.sb(T4, 0, T3)
        a.addi(T2, T2, 1)
        a.li(T6, 16)
        a.bne(T2, T6, "ad_first")
        a.li(A2, 144)                # offset of the round key: rounds 9 .. 1, then 0
        a.la(A3, L["ISR_PERM"])
        a.la(A4, L["INV_SBOX"])
        a.label("ad_round")
        a.la(A5, L["TMP16"])
        a.li(T2, 0)
        a.label("ad_sub")            # InvShiftRows, InvSubBytes, AddRoundKey
        a.add(T3, A3, T2)
        a.lbu(T3, 0, T3)
        a.add(T3, T1, T3)
        a.lbu(T3, 0, T3)
        a.add(T3, A4, T3)
        a.lbu(T3, 0, T3)
        a.add(T4, T0, A2)
        a.add(T4, T4, T2)
        a.lbu(T4, 0, T4)
        a.xor(T3, T3, T4)
        a.add(T4, A5, T2)
        a.sb(T3, 0, T4)
        a.addi(T2, T2, 1)
        a.li(T6, 16)
        a.bne(T2, T6, "ad_sub")
        a.beq(A2, ZERO, "ad_out")
        a.li(T2, 0)                  # InvMixColumns, one column at a time
        a.label("ad_mix")
        a.add(T3, A5, T2)
        a.lbu(S8, 0, T3)
        a.lbu(S9, 1, T3)
        a.lbu(S10, 2, T3)
        a.lbu(S11, 3, T3)
        a.add(T3, T1, T2)
        rows = ((14, 11, 13, 9), (9, 14, 11, 13), (13, 9, 14, 11), (11, 13, 9, 14))
        for r, coeff in enumerate(rows):
            for k, (c, reg) in enumerate(zip(coeff, (S8, S9, S10, S11))):
                a.la(T4, L["MUL%d" % c])
                a.add(T4, T4, reg)
                a.lbu(T4, 0, T4)
                if k == 0:
                    a.mv(T5, T4)
                else:
                    a.xor(T5, T5, T4)
            a.sb(T5, r, T3)
        a.addi(T2, T2, 4)
        a.li(T6, 16)
        a.bne(T2, T6, "ad_mix")
        a.addi(A2, A2, -16)
        a.j("ad_round")
        a.label("ad_out")
        a.li(T2, 0)
        a.label("ad_copy")
        a.add(T3, A5, T2)
        a.lbu(T4, 0, T3)
        a.add(T3, A1, T2)
        a.sb(T4, 0, T3)
        a.addi(T2, T2, 1)
        a.li(T6, 16)
        a.bne(T2, T6, "ad_copy")
        a.ret()

        # ---- the bit reader of inflate: s2 = next input byte, s3 = end of input, s4 = bit buffer, s5 = bits in it
        # bits(a0 = n <= 16) -> a0
        a.label("bits")
        a.label("bits_l")
        a.bge(S5, A0, "bits_have")
        a.bgeu(S2, S3, "inflate_bad")
        a.lbu(T0, 0, S2)
        a.addi(S2, S2, 1)
        a.sll(T0, T0, S5)
        a.or_(S4, S4, T0)
        a.addi(S5, S5, 8)
        a.j("bits_l")
        a.label("bits_have")
        a.li(T0, 1)
        a.sll(T0, T0, A0)
        a.addi(T0, T0, -1)
        a.and_(T1, S4, T0)
        a.srl(S4, S4, A0)
        a.sub(S5, S5, A0)
        a.mv(A0, T1)
        a.ret()
        a.label("inflate_bad")
        die(6)

        # ---- decode(a0 = code: 16 counts, then the symbols in canonical order) -> a0 = symbol; one bit at a time (RFC 1951 3.2.2)
        a.label("decode")
        a.li(T0, 0)                  # code
        a.li(T1, 0)                  # first code of this length
        a.li(T2, 0)                  # index of its first symbol
        a.li(T3, 1)                  # length
        a.label("dec_l")
        a.bne(S5, ZERO, "dec_bit")
        a.bgeu(S2, S3, "inflate_bad")
        a.lbu(S4, 0, S2)
        a.addi(S2, S2, 1)
        a.li(S5, 8)
        a.label("dec_bit")
        a.andi(T4, S4, 1)
        a.srli(S4, S4, 1)
        a.addi(S5, S5, -1)
        a.or_(T0, T0, T4)
        a.slli(T5, T3, 2)
        a.add(T5, A0, T5)
        a.lw(T5, 0, T5)              # codes of this length
        a.sub(T6, T0, T5)
        a.blt(T6, T1, "dec_found")
        a.add(T2, T2, T5)
        a.add(T1, T1, T5)
        a.slli(T1, T1, 1)
        a.slli(T0, T0, 1)
        a.addi(T3, T3, 1)
        a.li(T6, 16)
        a.bne(T3, T6, "dec_l")
        a.j("inflate_bad")
        a.label("dec_found")
        a.sub(T6, T0, T1)
        a.add(T6, T6, T2)
        a.slli(T6, T6, 2)
        a.add(T6, A0, T6)
        a.lw(A0, 64, T6)
        a.ret()

        # ---- construct(a0 = code, a1 = lengths (words), a2 = symbols): the canonical code of those lengths
        a.label("construct")
        a.li(T0, 0)
        a.label("con_zero")
        a.add(T1, A0, T0)
        a.sw(ZERO, 0, T1)
        a.addi(T0, T0, 4)
        a.li(T6, 64)
        a.bne(T0, T6, "con_zero")
        a.li(T0, 0)
        a.label("con_count")
        a.beq(T0, A2, "con_counted")
        a.slli(T1, T0, 2)
        a.add(T1, A1, T1)
        a.lw(T1, 0, T1)
        a.slli(T1, T1, 2)
        a.add(T1, A0, T1)
        a.lw(T2, 0, T1)
        a.addi(T2, T2, 1)
        a.sw(T2, 0, T1)
        a.addi(T0, T0, 1)
        a.j("con_count")
        a.label("con_counted")
        a.la(T3, L["OFFS"])
        a.sw(ZERO, 4, T3)
        a.li(T0, 1)
        a.label("con_offs")
        a.slli(T1, T0, 2)
        a.add(T2, T3, T1)
        a.lw(T4, 0, T2)
        a.add(T5, A0, T1)
        a.lw(T5, 0, T5)
        a.add(T4, T4, T5)
        a.sw(T4, 4, T2)
        a.addi(T0, T0, 1)
        a.li(T6, 15)
        a.bne(T0, T6, "con_offs")
        a.li(T0, 0)
        a.label("con_place")
        a.beq(T0, A2, "con_done")
        a.slli(T1, T0, 2)
        a.add(T1, A1, T1)
        a.lw(T1, 0, T1)
        a.beq(T1, ZERO, "con_next")
        a.slli(T1, T1, 2)
        a.add(T1, T3, T1)
        a.lw(T2, 0, T1)
        a.addi(T4, T2, 1)
        a.sw(T4, 0, T1)
        a.slli(T2, T2, 2)
        a.add(T2, A0, T2)
        a.sw(T0, 64, T2)
        a.label("con_next")
        a.addi(T0, T0, 1)
        a.j("con_place")
        a.label("con_done")
        a.ret()

        # ---- inflate(a0 = input, a1 = its end, a2 = output, a3 = its limit) -> a0 = end of the output (RFC 1951)
        # s2..s5 the bit reader, s6 = next output byte, s7 = limit, s8 = start of the output, s9 = last-block flag
        a.label("inflate")
        prologue(S0, S1, S10, S11)
        a.mv(S2, A0)
        a.mv(S3, A1)
        a.li(S4, 0)
        a.li(S5, 0)
        a.mv(S6, A2)
        a.mv(S7, A3)
        a.mv(S8, A2)
        a.label("inf_block")
        a.li(A0, 1)
        a.call("bits")
        a.mv(S9, A0)
        a.li(A0, 2)
        a.call("bits")
        a.beq(A0, ZERO, "inf_stored")
        a.li(T0, 1)
        a.beq(A0, T0, "inf_fixed")
        a.li(T0, 2)
        a.beq(A0, T0, "inf_dynamic")
        a.j("inflate_bad")

        a.label("inf_stored")        # the rest of the byte is skipped; LEN, NLEN, then LEN bytes as they are
        a.li(S4, 0)
        a.li(S5, 0)
        a.addi(T0, S2, 4)
        a.bltu(S3, T0, "inflate_bad")
        a.lbu(T1, 0, S2)
        a.lbu(T2, 1, S2)
        a.slli(T2, T2, 8)
        a.or_(T1, T1, T2)
        a.lbu(T2, 2, S2)
        a.lbu(T3, 3, S2)
        a.slli(T3, T3, 8)
        a.or_(T2, T2, T3)
        a.xor(T2, T2, T1)
        a.li(T3, 0xFFFF)
        a.bne(T2, T3, "inflate_bad")
        a.mv(S2, T0)
        a.add(T0, S2, T1)
        a.bltu(S3, T0, "inflate_bad")
        a.add(T2, S6, T1)
        a.bltu(S7, T2, "inflate_bad")
        a.label("inf_st_copy")
        a.beq(S2, T0, "inf_block_done")
        a.lbu(T3, 0, S2)
        a.sb(T3, 0, S6)
        a.addi(S2, S2, 1)
        a.addi(S6, S6, 1)
        a.j("inf_st_copy")

        def fill(first, last, value):  # LENGTHS[first..last) = value
            a.la(T0, L["LENGTHS"] + 4 * first)
            a.la(T1, L["LENGTHS"] + 4 * last)
            a.li(T2, value)
            lab = fresh("fill")
            a.label(lab)
            a.sw(T2, 0, T0)
            a.addi(T0, T0, 4)
            a.bne(T0, T1, lab)

        a.label("inf_fixed")         # RFC 1951 3.2.6
        fill(0, 144, 8)
        fill(144, 256, 9)
        fill(256, 280, 7)
        fill(280, 288, 8)
        a.la(A0, L["LENCODE"])
        a.la(A1, L["LENGTHS"])
        a.li(A2, 288)
        a.call("construct")
        fill(0, 30, 5)
        a.la(A0, L["DISTCODE"])
        a.la(A1, L["LENGTHS"])
        a.li(A2, 30)
        a.call("construct")
        a.j("inf_codes")

        a.label("inf_dynamic")       # RFC 1951 3.2.7
        a.li(A0, 5)
        a.call("bits")
        a.addi(S10, A0, 257)         # literal / length codes
        a.li(A0, 5)
        a.call("bits")
        a.addi(S11, A0, 1)           # distance codes
        a.li(A0, 4)
        a.call("bits")
        a.addi(S0, A0, 4)            # code length codes
        a.li(T0, 286)
        a.bltu(T0, S10, "inflate_bad")
        a.li(T0, 30)
        a.bltu(T0, S11, "inflate_bad")
        a.la(T0, L["CL_LENGTHS"])
        a.li(T1, 0)
        a.label("dyn_clz")
        a.add(T2, T0, T1)
        a.sw(ZERO, 0, T2)
        a.addi(T1, T1, 4)
        a.li(T6, 76)
        a.bne(T1, T6, "dyn_clz")
        a.li(S1, 0)
        a.label("dyn_cl")
        a.beq(S1, S0, "dyn_cl_done")
        a.li(A0, 3)
        a.call("bits")
        a.la(T0, L["ORDER"])
        a.slli(T1, S1, 2)
        a.add(T0, T0, T1)
        a.lw(T0, 0, T0)
        a.slli(T0, T0, 2)
        a.la(T1, L["CL_LENGTHS"])
        a.add(T1, T1, T0)
        a.sw(A0, 0, T1)
        a.addi(S1, S1, 1)
        a.j("dyn_cl")
        a.label("dyn_cl_done")
        a.la(A0, L["CLCODE"])
        a.la(A1, L["CL_LENGTHS"])
        a.li(A2, 19)
        a.call("construct")
        a.add(S0, S10, S11)          # lengths to read
        a.li(S1, 0)
        a.label("dyn_len")
        a.bgeu(S1, S0, "dyn_len_done")
        a.la(A0, L["CLCODE"])
        a.call("decode")
        a.li(T0, 16)
        a.bgeu(A0, T0, "dyn_rep")
        a.la(T0, L["LENGTHS"])
        a.slli(T1, S1, 2)
        a.add(T0, T0, T1)
        a.sw(A0, 0, T0)
        a.addi(S1, S1, 1)
        a.j("dyn_len")
        a.label("dyn_rep")           # 16: the previous length 3..6 times; 17: zero 3..10 times; 18: zero 11..138 times
        a.li(T0, 16)
        a.bne(A0, T0, "dyn_zero")
        a.beq(S1, ZERO, "inflate_bad")
        a.la(T0, L["LENGTHS"])
        a.slli(T1, S1, 2)
        a.add(T0, T0, T1)
        a.lw(T0, -4, T0)
        a.addi(SP, SP, -4)
        a.sw(T0, 0, SP)
        a.li(A0, 2)
        a.call("bits")
        a.addi(A0, A0, 3)
        a.lw(T2, 0, SP)
        a.addi(SP, SP, 4)
        a.j("dyn_run")
        a.label("dyn_zero")
        a.li(T0, 17)
        a.bne(A0, T0, "dyn_zero_long")
        a.li(A0, 3)
        a.call("bits")
        a.addi(A0, A0, 3)
        a.li(T2, 0)
        a.j("dyn_run")
        a.label("dyn_zero_long")
        a.li(A0, 7)
        a.call("bits")
        a.addi(A0, A0, 11)
        a.li(T2, 0)
        a.label("dyn_run")           # a0 times the length t2
        a.add(T3, S1, A0)
        a.bltu(S0, T3, "inflate_bad")
        a.la(T0, L["LENGTHS"])
        a.label("dyn_run_l")
        a.beq(S1, T3, "dyn_len")
        a.slli(T1, S1, 2)
        a.add(T1, T0, T1)
        a.sw(T2, 0, T1)
        a.addi(S1, S1, 1)
        a.j("dyn_run_l")
        a.label("dyn_len_done")
        a.la(T0, L["LENGTHS"] + 4 * 256)
        a.lw(T0, 0, T0)
        a.beq(T0, ZERO, "inflate_bad")  # no end-of-block code
        a.la(A0, L["LENCODE"])
        a.la(A1, L["LENGTHS"])
        a.mv(A2, S10)
        a.call("construct")
        a.la(A0, L["DISTCODE"])
        a.la(A1, L["LENGTHS"])
        a.slli(T0, S10, 2)
        a.add(A1, A1, T0)
        a.mv(A2, S11)
        a.call("construct")

        a.label("inf_codes")         # literals, lengths and distances until the end-of-block symbol
        a.la(A0, L["LENCODE"])
        a.call("decode")
        a.li(T0, 256)
        a.bgeu(A0, T0, "inf_not_literal")
        a.bgeu(S6, S7, "inflate_bad")
        a.sb(A0, 0, S6)
        a.addi(S6, S6, 1)
        a.j("inf_codes")
        a.label("inf_not_literal")
        a.beq(A0, T0, "inf_block_done")
        a.addi(S0, A0, -257)
        a.li(T0, 29)
        a.bgeu(S0, T0, "inflate_bad")
        a.slli(S0, S0, 2)
        a.la(T0, L["LEXT"])
        a.add(T0, T0, S0)
        a.lw(A0, 0, T0)
        a.call("bits")
        a.la(T0, L["LBASE"])
        a.add(T0, T0, S0)
        a.lw(T0, 0, T0)
        a.add(S1, T0, A0)            # length of the match
        a.la(A0, L["DISTCODE"])
        a.call("decode")
        a.li(T0, 30)
        a.bgeu(A0, T0, "inflate_bad")
        a.slli(S0, A0, 2)
        a.la(T0, L["DEXT"])
        a.add(T0, T0, S0)
        a.lw(A0, 0, T0)
        a.call("bits")
        a.la(T0, L["DBASE"])
        a.add(T0, T0, S0)
        a.lw(T0, 0, T0)
        a.add(T0, T0, A0)            # distance
        a.sub(T1, S6, S8)
        a.bltu(T1, T0, "inflate_bad")   # reaches back before the start of the output
        a.add(T1, S6, S1)
        a.bltu(S7, T1, "inflate_bad")
        a.sub(T2, S6, T0)
        a.label("inf_match")
        a.lbu(T3, 0, T2)
        a.sb(T3, 0, S6)
        a.addi(T2, T2, 1)
        a.addi(S6, S6, 1)
        a.bne(S6, T1, "inf_match")
        a.j("inf_codes")

        a.label("inf_block_done")
        a.beq(S9, ZERO, "inf_block")
        a.mv(A0, S6)
        epilogue(S0, S1, S10, S11)

        # ---- zlib_inflate(a0 = stream, a1 = its end, a2 = output, a3 = its limit) -> a0 = end of the output (RFC 1950)
        a.label("zlib_inflate")
        prologue(S0, S1)
        a.addi(T0, A0, 6)
        a.bltu(A1, T0, "inflate_bad")
        a.lbu(T1, 0, A0)             # CMF: method 8, window <= 32 K;  FLG: no preset dictionary, (CMF * 256 + FLG) % 31 == 0
        a.lbu(T2, 1, A0)
        a.andi(T3, T1, 15)
        a.li(T4, 8)
        a.bne(T3, T4, "inflate_bad")
        a.andi(T3, T2, 32)
        a.bne(T3, ZERO, "inflate_bad")
        a.slli(T3, T1, 8)
        a.or_(T3, T3, T2)
        a.li(T4, 31)
        a.emit((1 << 25) | (T4 << 20) | (T3 << 15) | (7 << 12) | (T3 << 7) | 0x33)  # remu t3, t3, t4
        a.bne(T3, ZERO, "inflate_bad")
        a.mv(S0, A2)                 # start of the output, for the checksum
        a.addi(A0, A0, 2)
        a.call("inflate")
        a.mv(S1, A0)
        # Adler-32 of the output against the four big-endian bytes behind the deflate data (s2 = where the bit reader stopped)
        a.addi(T0, S2, 4)
        a.bltu(S3, T0, "inflate_bad")
        a.li(T1, 1)                  # a
        a.li(T2, 0)                  # b
        a.li(T6, 65521)
        a.mv(T3, S0)
        a.label("adler_l")
        a.beq(T3, S1, "adler_d")
        a.lbu(T4, 0, T3)
        a.add(T1, T1, T4)
        a.bltu(T1, T6, "adler_a")
        a.sub(T1, T1, T6)
        a.label("adler_a")
        a.add(T2, T2, T1)
        a.bltu(T2, T6, "adler_b")
        a.sub(T2, T2, T6)
        a.label("adler_b")
        a.addi(T3, T3, 1)
        a.j("adler_l")
        a.label("adler_d")
        a.slli(T2, T2, 16)
        a.or_(T1, T1, T2)
        a.lbu(T3, 0, S2)
        a.slli(T3, T3, 24)
        a.lbu(T4, 1, S2)
        a.slli(T4, T4, 16)
        a.or_(T3, T3, T4)
        a.lbu(T4, 2, S2)
        a.slli(T4, T4, 8)
        a.or_(T3, T3, T4)
        a.lbu(T4, 3, S2)
        a.or_(T3, T3, T4)
        a.bne(T1, T3, "inflate_bad")
        a.mv(A0, S1)
        epilogue(S0, S1)


def build():
    return guest_rsa.build(extend=Camt53())


def input_stream(iban, host_info, authenticated, form=1, e=65537, **x):
    """the guest's input words, in the order it reads them: SignedInfo, the authenticated part, the operands of the three RSA operations
    (as tools/guest_rsa.py takes them), the two strings the commitment opens with (host/src/main.rs:405-409: iban, host_info) and the
    commitment form (1: the current one with the three keys; 0: the earlier one of test.xml-Receipt-test.json)"""
    limbs, frame = guest_rsa.limbs, guest_rsa.message_frame
    words = frame(x["signed_info"]) + frame(authenticated) + limbs(int.from_bytes(x["bank_sig"], "big")) + limbs(x["bank_n"]) + [e]
    words += limbs(int.from_bytes(x["tx_plain"], "big")) + limbs(x["client_n"]) + limbs(int.from_bytes(x["tx_cipher"], "big")) + [e]
    words += frame(x["order_data"]) + limbs(int.from_bytes(x["witness_sig"], "big")) + limbs(x["witness_n"]) + [e]
    return words + frame(iban.encode()) + frame(host_info.encode()) + [form]


def reference_authenticated():
    return open(os.path.join(ROOT, "tests", "golden", "camt53", "test.xml-authenticated"), "rb").read()


REFERENCE_IBAN, REFERENCE_HOST_INFO = "CH4308307000289537312", "host:main"  # host/src/main.rs:452 TEST_IBAN; the host info of the reference's test run


def elf_and_input(form=1):
    image, _, _ = build()
    return image, input_stream(REFERENCE_IBAN, REFERENCE_HOST_INFO, reference_authenticated(), form=form, **guest_rsa.reference_inputs()), (
        "the hyperfridge pipeline on the reference's EBICS fixture (tools/guest_camt53.py): SHA-256, three RSA-2048 public-key operations, AES-128-CBC, "
        "inflate, unzip, camt.053 field extraction; commits the journal of the reference's receipt fixture")


if __name__ == "__main__":
    image, labels, L = build()
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "circuits", "guest_camt53.elf")
    open(out, "wb").write(image)
    print("guest_camt53: %d bytes, entry %#x -> %s" % (len(image), labels["_start"], out))
