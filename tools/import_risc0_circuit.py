#!/usr/bin/env python3
"""Convert a risc0 circuit's generated Rust tables into a circuit blob (include/r0hip_circuit.h).

risc0 compiles its circuits from two machine-generated Rust files (recalled layout; the crates are not vendored in the
reference, so this tool has only been exercised on text produced by its own --emit-rust mode):

  taps.rs      pub const TAPSET: &TapSet = &TapSet::<'static> { taps: &[TapData { offset: 0, back: 0, group: 0, combo: 0,
               skip: 1 }, ...], combo_taps: &[...], combo_begin: &[...], group_begin: &[...], ... };
  poly_ext.rs  pub const DEF: PolyExtStepDef = PolyExtStepDef { block: &[PolyExtStep::Const(1), PolyExtStep::Get(5),
               PolyExtStep::GetGlobal(0, 3), PolyExtStep::Add(0, 1), PolyExtStep::Sub(..), PolyExtStep::Mul(..),
               PolyExtStep::True, PolyExtStep::AndEqz(0, 5), PolyExtStep::AndCond(1, 7, 0), ...], ret: 1234 };

The resulting blob has GROUPS / TAPS / GLOBALS / POLY / INFO and no WITGEN/ACCUM sections: witness generation and
accumulation stay with risc0's own step functions; the per-op C ABI (NTT, Merkle, eval_check, FRI...) consumes the blob.

usage:
  import_risc0_circuit.py taps.rs poly_ext.rs out.r0c [--info RV32IM:v2_______] [--n-global N] [--n-mix N]
  import_risc0_circuit.py --emit-rust in.r0c out_dir      (writes taps.rs / poly_ext.rs text from a blob: used by the tests)
"""
import argparse
import os
import re
import struct
import sys

MAGIC = 0x31433052
SEC_GROUPS, SEC_TAPS, SEC_GLOBALS, SEC_POLY, SEC_INFO = 1, 2, 3, 4, 7
OPS = {"Const": 0, "Get": 2, "GetGlobal": 3, "Add": 4, "Sub": 5, "Mul": 6, "True": 7, "AndEqz": 8, "AndCond": 9}
OP_NAMES = {v: k for k, v in OPS.items()}


def parse_taps(text):
    taps = []
    for m in re.finditer(r"TapData\s*\{([^}]*)\}", text):
        fields = dict((k, int(v)) for k, v in re.findall(r"(\w+)\s*:\s*(\d+)", m.group(1)))
        taps.append((fields["group"], fields["offset"], fields["back"]))
    if not taps:
        raise SystemExit("no TapData entries found")
    if taps != sorted(taps):
        raise SystemExit("taps are not sorted by (group, offset, back)")
    return taps


def parse_poly(text):
    m = re.search(r"block\s*:\s*&\[(.*?)\]\s*,\s*ret\s*:\s*(\d+)", text, re.S)
    if not m:
        raise SystemExit("no `block: &[...], ret: N` found")
    steps = []
    for sm in re.finditer(r"PolyExtStep::(\w+)(?:\(([^)]*)\))?", m.group(1)):
        name, args = sm.group(1), [int(a) for a in re.findall(r"\d+", sm.group(2) or "")]
        if name not in OPS:
            raise SystemExit("unknown PolyExtStep::%s" % name)
        args += [0] * (3 - len(args))
        steps.append((OPS[name], args[0], args[1], args[2]))
    return steps, int(m.group(2))


def build_blob(taps, steps, ret, info, n_global=None, n_mix=None):
    groups = [0, 0, 0]
    for g, off, _ in taps:
        if g > 2:
            raise SystemExit("tap group %d: this prover expects ACCUM=0, CODE=1, DATA=2" % g)
        groups[g] = max(groups[g], off + 1)
    need = [0, 0]
    for op, a, b, _ in steps:
        if op == OPS["GetGlobal"]:
            if a > 1:
                raise SystemExit("GetGlobal base %d: expected 0 (global/out) or 1 (mix)" % a)
            need[a] = max(need[a], b + 1)
    n_global = need[0] if n_global is None else n_global
    n_mix = need[1] if n_mix is None else n_mix
    tag = info.encode()
    if len(tag) != 16:
        raise SystemExit("--info must be exactly 16 bytes")

    def section(tagid, words):
        return [tagid, len(words)] + list(words)

    words = [MAGIC, 1, 5]
    words += section(SEC_INFO, list(struct.unpack("<4I", tag)))
    words += section(SEC_GROUPS, groups)
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, n_mix])
    words += section(SEC_POLY, [len(steps), ret] + [w for s in steps for w in s])
    return words


def read_sections(path):
    w = struct.unpack("<%dI" % (os.path.getsize(path) // 4), open(path, "rb").read())
    assert w[0] == MAGIC and w[1] == 1
    pos, out = 3, {}
    for _ in range(w[2]):
        out[w[pos]] = w[pos + 2:pos + 2 + w[pos + 1]]
        pos += 2 + w[pos + 1]
    return out


def emit_rust(blob_path, out_dir):
    sec = read_sections(blob_path)
    taps = sec[SEC_TAPS]
    n = taps[0]
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "taps.rs"), "w") as f:
        f.write("pub const TAPSET: &TapSet = &TapSet::<'static> {\n    taps: &[\n")
        for i in range(n):
            g, off, back = taps[1 + 3 * i:4 + 3 * i]
            f.write("        TapData { offset: %d, back: %d, group: %d, combo: 0, skip: 1 },\n" % (off, back, g))
        f.write("    ],\n};\n")
    poly = sec[SEC_POLY]
    with open(os.path.join(out_dir, "poly_ext.rs"), "w") as f:
        f.write("pub const DEF: PolyExtStepDef = PolyExtStepDef {\n    block: &[")
        items = []
        for i in range(poly[0]):
            op, a, b, c = poly[2 + 4 * i:6 + 4 * i]
            name = OP_NAMES[op]
            nargs = {"Const": 1, "Get": 1, "GetGlobal": 2, "Add": 2, "Sub": 2, "Mul": 2, "True": 0, "AndEqz": 2, "AndCond": 3}[name]
            items.append("PolyExtStep::%s%s" % (name, "(%s)" % ", ".join(str(v) for v in (a, b, c)[:nargs]) if nargs else ""))
        f.write(", ".join(items))
        f.write("],\n    ret: %d,\n};\n" % poly[1])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--emit-rust":
        emit_rust(sys.argv[2], sys.argv[3])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("taps_rs")
    ap.add_argument("poly_ext_rs")
    ap.add_argument("out")
    ap.add_argument("--info", default="RV32IM:v2_______")
    ap.add_argument("--n-global", type=int, default=None)
    ap.add_argument("--n-mix", type=int, default=None)
    a = ap.parse_args()
    taps = parse_taps(open(a.taps_rs).read())
    steps, ret = parse_poly(open(a.poly_ext_rs).read())
    words = build_blob(taps, steps, ret, a.info, a.n_global, a.n_mix)
    with open(a.out, "wb") as f:
        f.write(struct.pack("<%dI" % len(words), *words))
    print("wrote %s: %d taps, %d steps, groups from taps" % (a.out, len(taps), len(steps)))


if __name__ == "__main__":
    main()
