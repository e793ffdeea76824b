#!/usr/bin/env python3
"""Seed corpus for tools/fuzz/fuzz_host.cpp: one genuine input per target (first byte = target index), from tests/golden/ and
circuits/.  usage: python tools/fuzz/make_seeds.py <corpus dir>"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "tests", "golden")


def rd(*p):
    return open(os.path.join(*p), "rb").read()


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)

    def put(name, target, payload):
        open(os.path.join(out, name), "wb").write(bytes([target]) + payload)

    put("ebics_response", 0, rd(G, "camt53", "response.xml"))
    for lvl in (0, 1, 9):
        put("zlib_%d" % lvl, 1, zlib.compress(rd(G, "camt53", "test.xml-SignedInfo") * 3, lvl))
    for k, name in enumerate(["pub_bank.pem", "pub_client.pem", "client.pem", "witness.pem", "test.xml-Witness.hex", "test.xml-TransactionKeyDecrypt.bin"]):
        put("key_%d" % k, 2, rd(G, "camt53", name))
    # ELF32 RISC-V executable: li a0, 42; li a7, 0; ecall
    code = struct.pack("<3I", 0x02A00513, 0x00000893, 0x00000073)
    entry = vaddr = 0x10000
    ehdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, entry, 52, 0, 0, 52, 32, 1, 0, 0, 0)
    phdr = struct.pack("<IIIIIIII", 1, 84, vaddr, vaddr, len(code), len(code) + 64, 5, 4)
    put("elf", 3, ehdr + phdr + code)
    # raw words: a loop with loads, stores, a multiply, a read ecall, a commit and a halt
    prog = [0x00001137, 0x00400593, 0x00100893, 0x00010513, 0x00000073, 0x00012283, 0x02528333, 0x00612223, 0x00200893, 0x00010513,
            0x00800593, 0x00000073, 0xFE0298E3, 0x00000893, 0x00000513, 0x00000073]
    put("words", 4, struct.pack("<%dI" % len(prog), *prog))
    for name in ("reference_receipt_test.json", "reference_receipt_6bb95807_latest.json"):
        put("receipt_" + name[:20], 5, rd(G, name))
    seal = np.load(os.path.join(G, "seal_tiny_po2_9_seed_1.npy")).astype(np.uint32)
    try:  # a composite receipt in our writer's form, if the library is built
        sys.path.insert(0, ROOT)
        import hyperfridge_r0_amd as r0
        rc = r0.Receipt.new(b'{"iban":"CH43"}', [seal], None)
        put("receipt_composite", 5, rc.to_json().encode() if isinstance(rc.to_json(), str) else rc.to_json())
        rc.image_proof = seal[:64]  # (the optional image proof of a trace-circuit session)
        put("receipt_composite_image_proof", 5, rc.to_json().encode())
    except Exception as e:  # noqa: BLE001
        print("no composite receipt seed:", e)
    put("seal", 6, seal.tobytes())
    put("circuit", 7, rd(ROOT, "circuits", "tiny.r0c"))
    for name in ("recursion", "trace"):  # the sections only these carry: PERIODIC / SPONGE, LATE / LOGUP
        path = os.path.join(ROOT, "circuits", name + ".r0c")
        if os.path.exists(path):
            put("circuit_" + name, 7, open(path, "rb").read())
    put("seal_edit", 8, struct.pack("<II", 1000, 12345))
    put("seal_cut", 8, struct.pack("<IIB", 7, 7, 3))
    print("seeds written to", out)


if __name__ == "__main__":
    main()
