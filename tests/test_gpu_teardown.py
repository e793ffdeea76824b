"""Teardown order.  The first bench run of round 1 ended with
    terminate called after throwing an instance of 'std::bad_variant_access'   what(): std::get: wrong index for variant
at interpreter exit, after its JSON line had printed (gpurun_out/bench1.err).  Nothing in this repository uses std::variant: the
exception came out of the HIP runtime, which was handed a dead stream -- `r0h_ctx_destroy` deleted the context at once, and the
device buffers / circuit that Python's garbage collector released afterwards still pointed at it (`r0h_buf_free` synchronises the
buffer's context stream before hipFree; a circuit unloads its hipRTC / code-object module).  Commit 7ff8ba7 made the context
reference-counted: the handle, every buffer, every circuit and every proof in flight hold a reference, and the device state goes
with the last of them.  These tests pin that in a child process each, so that an abort is seen as an exit status."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PRELUDE = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as entry
import hyperfridge_r0_amd as r0
hal = r0.Hal(0)
tiny = np.fromfile(entry.circuit_blob_path("tiny"), dtype=np.uint32)     # eval_check compiled in-process (hipRTC module)
small = np.fromfile(entry.circuit_blob_path("small"), dtype=np.uint32)   # eval_check from a code object file
c_rtc = hal.load_circuit(tiny)
c_obj = hal.load_circuit(small, entry.code_object_path("small"))
code, data, glob = hal.witgen(c_obj, 10, seed=3)
buf = hal.copy_from(np.arange(1 << 16, dtype=np.uint32))
view = buf.slice(16, 64)
proof, mix = hal.proof_begin(c_obj, 10, code, data, glob)               # a proof in flight: pooled device buffers + transcript
seal = hal.prove_segment(c_obj, 10, code, data, glob)
""" % ROOT


def _run(body):
    out = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    return out


def test_context_destroyed_first_then_everything_else_is_released():
    out = _run("""
        hal.close()                      # the handle goes first; buffers, circuits and the proof are still alive
        hal.ctx = None
        lib = r0.lib()
        assert not lib.r0h_proof_abort(proof)
        for obj in (view, buf, code, data, c_rtc, c_obj):
            obj.free()
        print("released", seal.size)
    """)
    assert out.returncode == 0 and "released" in out.stdout and "terminate" not in out.stderr, out.stderr[-2000:]


def test_interpreter_exit_with_every_handle_still_alive():
    out = _run("""
        print("exiting with live handles", seal.size)
    """)
    assert out.returncode == 0 and "exiting with live handles" in out.stdout and "terminate" not in out.stderr, out.stderr[-2000:]


def test_garbage_collection_in_the_bad_old_order():
    """What round 1's first bench did: close the context, then let the objects die in whatever order the interpreter picks."""
    out = _run("""
        hal.close()
        hal.ctx = None
        del view, buf
        import gc; gc.collect()
        del c_rtc, code
        gc.collect()
        print("collected", seal.size)
    """)
    assert out.returncode == 0 and "collected" in out.stdout and "terminate" not in out.stderr, out.stderr[-2000:]


def test_the_library_and_torch_share_a_process_in_either_import_order():
    """Round 3's verdict, item 5: libr0hip.so links /opt/rocm's HIP runtime, torch ships its own; whichever initialised second found no
    device when that second one was torch's (gpurun_out/z_sharded2.err under torch.distributed.run).  hyperfridge_r0_amd.lib() now
    puts torch's runtime first whatever the caller's import order: a child process each way round, a context and a device tensor in
    both, exit status 0."""
    orders = {
        "library first": "import hyperfridge_r0_amd as r0\nhal = r0.Hal(0)\nimport torch\nt = torch.zeros(4, device='cuda') + 1\n",
        "torch first": "import torch\nt = torch.zeros(4, device='cuda') + 1\nimport hyperfridge_r0_amd as r0\nhal = r0.Hal(0)\n",
    }
    tail = "import numpy as np\nb = hal.copy_from(np.arange(64, dtype=np.uint32))\nassert int(t.sum().item()) == 4 and b.to_host()[63] == 63\nhal.close()\nprint('both work')\n"
    for name, head in orders.items():
        out = subprocess.run([sys.executable, "-c", "import sys\nsys.path.insert(0, %r)\n" % ROOT + head + tail], capture_output=True, text=True, cwd=ROOT, timeout=600)
        assert out.returncode == 0 and "both work" in out.stdout, (name, out.stderr[-2000:])
