#!/bin/bash
# Differential fuzzing of the product's seal verifier against the oracle's (tools/fuzz/fuzz_verify_diff.cpp).  CPU only.
#   tools/fuzz/run_diff.sh [seconds, default 120] [work dir, default /tmp/r0h_fuzz_diff] [parallel jobs, default 1]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SECS=${1:-120}; WORK=${2:-/tmp/r0h_fuzz_diff}; JOBS=${3:-1}
CLANG=/opt/rocm/lib/llvm/bin/clang++; CC=/opt/rocm/lib/llvm/bin/clang
mkdir -p "$WORK/obj" "$WORK/corpus"
# coverage counters only: the compare-tracing hooks of -fsanitize=fuzzer-no-link slow field arithmetic down a hundredfold
FLAGS="-O2 -g -fno-omit-frame-pointer -fsanitize=address -fsanitize-coverage=inline-8bit-counters,pc-table"
for f in verify ctx circuit; do
  src=$ROOT/hyperfridge-r0_amd/csrc/$f.cpp; [ -f "$src" ] || src=$ROOT/hyperfridge-r0_amd/csrc/$f.hip  # host-only units are .cpp, circuit is .hip
  if [ ! -f "$WORK/obj/$f.o" ] || [ "$src" -nt "$WORK/obj/$f.o" ]; then
    if [ $f = circuit ]; then MODE="--offload-arch=gfx950 -fno-gpu-sanitize"; else MODE="-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"; fi
    /opt/rocm/bin/hipcc $MODE $FLAGS -w -c "$src" -o "$WORK/obj/$f.o"
  fi
done
for src in "$ROOT"/oracle/*.c; do   # the oracle, single-threaded (its OpenMP pragmas are ignored), under the same sanitizers
  o=$WORK/obj/orc_$(basename "$src" .c).o
  # (ASan only for plain clang objects: with UBSan as well, ASan's start-up check trips over merged string literals)
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then $CC $FLAGS -w -I"$ROOT/oracle" -I"$ROOT/include" -c "$src" -o "$o"; fi
done
cat > "$WORK/stubs.cpp" <<'STUB'
#include <stdint.h>
struct r0h_ctx; struct r0h_buf;
struct r0h_circuit;
namespace r0h { const char* ntt_init_device() { return nullptr; } void session_rows_free(r0h_ctx*) {}
const char* logup_accum(r0h_ctx*, const r0h_circuit*, uint32_t, const r0h_buf*, const r0h_buf*, const uint32_t*, const uint32_t*, r0h_buf*) { __builtin_trap(); } }
extern "C" const char* r0h_prefix_products(r0h_ctx*, r0h_buf*, uint32_t) { __builtin_trap(); }
STUB
$CLANG $FLAGS -c "$WORK/stubs.cpp" -o "$WORK/obj/stubs.o"
$CLANG -O1 -g -fsanitize=address,fuzzer-no-link -I"$ROOT/include" -c "$ROOT/tools/fuzz/fuzz_verify_diff.cpp" -o "$WORK/harness.o"
$CLANG -fsanitize=address,fuzzer "$WORK/harness.o" "$WORK"/obj/*.o -L/opt/rocm/lib -lamdhip64 -lhiprtc -lm -Wl,-rpath,/opt/rocm/lib -o "$WORK/fuzz_verify_diff"
printf '\0\0\0\0\0\1\0\0\0' > "$WORK/corpus/one_word"
printf '\2\100\0\0\0\0\0\0\0' > "$WORK/corpus/cut"
cd "$WORK"
R0H_FUZZ_ROOT=$ROOT ASAN_OPTIONS=detect_leaks=1:detect_odr_violation=0 ./fuzz_verify_diff corpus $([ "$JOBS" -gt 1 ] && echo -fork=$JOBS) -max_total_time=$SECS -timeout=30 -rss_limit_mb=4000 -max_len=900 -print_final_stats=1
