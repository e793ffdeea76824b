#!/usr/bin/env python3
"""Development check of the trace circuit on the device (run under gpurun): device witness == host witness, device totals and
seal == the oracle's, a camt53 session proved in two phases and verified with the ELF, wrong ELF refused, timing.
usage: python tools/dev_trace_gpu.py [--skip-session] [--po2 20]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-session", action="store_true")
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.ensure_built()
    import hyperfridge_r0_amd as r0
    import orc_binding
    from test_rv32im import _guest
    blob = np.fromfile(entry.circuit_blob_path("trace"), dtype=np.uint32)
    hal = r0.Hal(0)
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    orc = orc_binding.load()
    oc = orc.circuit(blob)
    if not args.skip_parity:
        prog, base = _guest(3000), 0x400
        vm = r0.Vm()
        vm.load(base, prog)
        vm.set_pc(base)
        vm.set_input([7, 0x01020304])
        assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 0)
        po2 = 16
        data, glob = vm.trace_witness(0, po2)
        rows, bounds = vm.preflight_arrays(0)
        seg = vm.segments()[0]
        dev, dglob = hal.trace_witgen(rows, bounds, po2, number=1, closing=bool(seg.closing), idle_pc=seg.pre.pc, circuit=gc)
        got = dev.to_host()
        cols = r0.trace_column_names()
        diff = [cols[c] for c in range(r0.TRACE_COLUMNS) if not np.array_equal(got.reshape(r0.TRACE_COLUMNS, -1)[c], data.reshape(r0.TRACE_COLUMNS, -1)[c])]
        print("witness columns that differ:", diff, "globals equal:", np.array_equal(dglob, glob))
        assert not diff and np.array_equal(dglob, glob)
        rng = np.random.default_rng(1)
        glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(int(x)) for x in rng.integers(0, 2013265921, 16)]
        code, synthetic, _ = hal.witgen(gc, po2, 0)
        synthetic.free()
        ocode, _, _ = oc.witgen(po2, 0)
        assert np.array_equal(ocode, code.to_host()), "CODE columns differ"
        want = oc.logup_totals(po2, ocode, data, glob)
        have = hal.logup_totals(gc, po2, code, dev, glob)
        print("totals", want[36:40], have[36:40])
        assert np.array_equal(want, have)
        glob = have
        # accumulation parity under a made-up mix
        mix = np.array([orc.enc(int(x)) for x in rng.integers(0, 2013265921, oc.n_mix)], dtype=np.uint32)
        t0 = time.perf_counter()
        acc = hal.accum_public(gc, po2, code, dev, glob, mix)
        hal.sync()
        print("device accum %.2f ms" % (1e3 * (time.perf_counter() - t0)))
        acc_h = acc.to_host()
        acc_o = oc.accum_public(po2, ocode, data, glob, mix)
        bad = [k for k in range(oc.group_size[0]) if not np.array_equal(acc_h.reshape(oc.group_size[0], -1)[k], acc_o.reshape(oc.group_size[0], -1)[k])]
        print("accum columns that differ:", bad)
        assert not bad
        n = 1 << po2
        last = acc_o.reshape(oc.group_size[0], n)[4 * (oc.group_size[0] // 4 - 2):4 * (oc.group_size[0] // 4 - 1), n - 1]
        print("chain total (must be 0):", last)
        assert not last.any()
        cc = hal.code_commit(gc, po2, code)
        seal = hal.prove_segment(gc, po2, cc, dev, glob)
        root = cc.root()
        print("device seal", seal.size, "words; oracle verify:", oc.verify(seal, code_root=root), "product verify:", r0.verify_seal(blob, seal, code_root=root)[:2])
        want_seal = oc.prove(po2, ocode, data, glob)
        print("seal == oracle seal:", np.array_equal(seal, want_seal))
        assert np.array_equal(seal, want_seal)
        print("phases (ms):", ", ".join("%s=%.2f" % p for p in hal.last_profile()))
        cc.free(); code.free(); dev.free(); acc.free()
    if not args.skip_session:
        import guest_camt53
        elf, stream, what = guest_camt53.elf_and_input()
        for it in range(args.repeat):
            hal.kernel_timing(True)
            t0 = time.perf_counter()
            receipt, image_id, cycles = hal.prove_elf(gc, elf, stream, segment_po2=args.po2)
            wall = time.perf_counter() - t0
            st = hal.last_session_stats()
            n = len(receipt.seals())
            print(json.dumps({"run": it, "segments": n, "cycles": cycles, "wall_s": round(wall, 4), "segments_per_s": round(n / wall, 3), "stats": st}))
        ks = hal.kernel_stats()
        tot = sum(v["total_ms"] for v in ks.values())
        print("kernel families (lane 0 context only):", json.dumps({k: round(v["total_ms"], 2) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])}), "total", round(tot, 1))
        seals = receipt.seals()
        roots = {}
        for _, seal in seals:
            size = r0.verify_seal(blob, seal)[2]
            if size not in roots:
                t0 = time.perf_counter()
                roots[size] = r0.control_root_host(blob, size)
                print("control root of 2^%d on the host: %.1f s" % (size, time.perf_counter() - t0))
        t0 = time.perf_counter()
        v = receipt.verify(blob, roots, None, elf=elf)
        print("verify with the ELF:", v, "%.2f s" % (time.perf_counter() - t0))
        assert v[:2] == (0, "ok"), v
        print("verify with the image id alone:", receipt.verify(blob, roots, image_id))
        other = bytearray(elf)
        other[-8] ^= 1  # one bit of the image
        print("verify with another ELF:", receipt.verify(blob, roots, None, elf=bytes(other)))
        assert receipt.verify(blob, roots, None, elf=bytes(other))[0] != 0
        for _, seal in seals[:1]:
            print("oracle verifier on seal 0:", oc.verify(seal, code_root=roots[r0.verify_seal(blob, seal)[2]]))
        print("closing flags:", [int(orc.dec(int(s[16]))) for _, s in seals], "cycles:", [int(orc.dec(int(s[10]))) for _, s in seals])
    gc.free()
    hal.close()
    print("dev_trace_gpu ok")


if __name__ == "__main__":
    main()
