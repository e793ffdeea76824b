#!/bin/bash
# R real camt53 sessions through the compiled host, C in flight: how many sessions should share the card (bench.py's default)
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tools")
import guest_camt53
_, stream, _ = guest_camt53.elf_and_input(form=1)
np.array(stream, dtype=np.uint32).tofile("gpurun_out/env_camt53.bin")
PY
mkdir -p gpurun_out/rcpt
for C in 1 2 3 4; do
  for L in 2 3; do
    R0H_SESSION_LANES=$L ./hyperfridge-r0_amd/r0h_prove circuits/trace.r0c --code-object circuits/trace.evalcheck.hsaco --elf circuits/guest_camt53.elf \
      --input gpurun_out/env_camt53.bin --po2 20 --receipts $((4*C)) --contexts $C --receipt-dir gpurun_out/rcpt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('contexts', $C, 'lanes', $L, 'receipts', d['receipts'], 'seg/s', d['segments_per_s'], 'receipts/s', d['receipts_per_s'], 'verify_s', d['verify_seconds'])"
  done
done
rm -rf gpurun_out/rcpt gpurun_out/env_camt53.bin
