"""bench.py's roofline.valu_view prices the Poseidon2 permutation with profiles/r02/p2_issue_floor.json; that file must describe
the code as it is now (tools/p2_issue_floor.py re-derives it from the compiled kernel -- hipcc only, no GPU)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_committed_issue_floor_matches_the_current_kernel(tmp_path):
    out = str(tmp_path / "floor.json")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "p2_issue_floor.py"), out], stdout=subprocess.DEVNULL)
    fresh = json.load(open(out))
    committed = json.load(open(os.path.join(ROOT, "profiles", "r02", "p2_issue_floor.json")))
    assert fresh["issue_floor_simd_cycles_per_wave_permutation"] < fresh["rate_table_estimate_simd_cycles_per_wave_permutation"]
    for key in ("valu_instructions_per_permutation", "issue_floor_simd_cycles_per_wave_permutation", "rate_table_estimate_simd_cycles_per_wave_permutation", "by_class"):
        assert fresh[key] == committed[key], "%s: re-run tools/p2_issue_floor.py and commit profiles/r02/p2_issue_floor.json" % key
