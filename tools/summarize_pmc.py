#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into profiles/<round>/pmc_traffic.json.
usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<how it was collected>" [segments profiled] [setup launches JSON] [circuit]
The last argument, e.g. '{"hash_rows_kernel": 1}', names launches that belong to the one-off setup of the run (the CODE group committed once
per (circuit, po2) before any segment is proved): that many of a kernel's FIRST dispatches are left out, so that launches / segments
profiled is the per-segment count bench.py compares with its own."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # device_code_fingerprint(): the traffic is only quoted for exactly the code it was measured on


def load(path, counter, skip):
    agg = collections.defaultdict(lambda: [0, 0.0])
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    left = dict(skip)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("r0h::", "")
        if left.get(name, 0) > 0:  # a setup launch (see the usage text)
            left[name] -= 1
            continue
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    skip = json.loads(sys.argv[6]) if len(sys.argv) > 6 else {}
    circuit = sys.argv[7] if len(sys.argv) > 7 else "bench"
    f, w = load(sys.argv[1], "FETCH_SIZE", skip), load(sys.argv[2], "WRITE_SIZE", skip)
    out = {"_about": sys.argv[4] + "  Counters are in KB.  fetch_corrected doubles FETCH_SIZE (gfx950 tallies the 128-B requests of coalesced "
                     "streaming reads at 64 B: MI355X_MICROARCH.md, HBM section).  Calibrated on this code's own access patterns: every in-place pass "
                     "(ntt_strided16 with its 128-byte tile rows, bit_reverse_tiled, the in-place ntt_local16) reads exactly what it writes and shows "
                     "raw FETCH_SIZE = WRITE_SIZE / 2, so the factor applies to all kernels here.  All byte figures are per launch.", "device_code_sha256": bench.device_code_fingerprint(circuit), "circuit": circuit, "kernels": {}}
    for k, (n, fs) in f.items():
        wn, ws = w.get(k, [0, 0.0])
        if not n:
            continue
        raw, wr = fs * 1024 / n, ws * 1024 / max(wn, 1)
        # the scan kernels (divide_*, prefix_*) give every lane its own run of 16-byte elements, 256 bytes apart from its neighbour's:
        # their requests are 64-byte ones, which FETCH_SIZE counts in full
        corr = raw if k.startswith(("divide_", "prefix_")) else 2 * raw
        out["kernels"][k] = {"launches": n, "fetch_raw_bytes": round(raw), "fetch_corrected_bytes": round(corr), "write_bytes": round(wr),
                             "hbm_bytes_per_launch": round(corr + wr)}
    if len(sys.argv) > 5:
        out["segments_profiled"] = int(sys.argv[5])  # launches / this = launches per segment (bench.py compares it with its own count)
    if skip:
        out["setup_launches_left_out"] = skip
    out["kernel_names"] = sorted(out["kernels"])
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
