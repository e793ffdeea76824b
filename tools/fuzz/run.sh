#!/bin/bash
# libFuzzer + AddressSanitizer + UndefinedBehaviorSanitizer over the host-only parsers (tools/fuzz/fuzz_host.cpp).  CPU only: the host-only translation units (.cpp) and the host
# side of circuit.hip are compiled with the sanitizers; no device code is run.
#   tools/fuzz/run.sh [seconds, default 120] [work dir, default /tmp/r0h_fuzz] [parallel jobs, default 1]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SECS=${1:-120}; WORK=${2:-/tmp/r0h_fuzz}; JOBS=${3:-1}
CLANG=/opt/rocm/lib/llvm/bin/clang++
mkdir -p "$WORK/obj" "$WORK/corpus"
FLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined,fuzzer-no-link -fno-sanitize-recover=undefined"
for f in ebics rv32im receipt claim verify ctx circuit image; do
  src=$ROOT/hyperfridge-r0_amd/csrc/$f.cpp; [ -f "$src" ] || src=$ROOT/hyperfridge-r0_amd/csrc/$f.hip  # host-only units are .cpp, circuit is .hip
  if [ ! -f "$WORK/obj/$f.o" ] || [ "$src" -nt "$WORK/obj/$f.o" ]; then
    # circuit.hip (the blob parser and the code generator live there) carries kernels: its host stubs need the code object, so it
    # is compiled whole with the sanitizer on the host side only; the rest is host code
    if [ $f = circuit ]; then MODE="--offload-arch=gfx950 -fno-gpu-sanitize"; else MODE="-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"; fi
    # field arithmetic (verifier, transcript, claims) gets edge counters only: libFuzzer's compare-tracing hooks slow it down a
    # hundredfold and guide nothing there; the parsers keep them (magic numbers, lengths)
    case $f in verify|ctx|claim) F=${FLAGS/fuzzer-no-link/} ; F="${F/-fsanitize=address,undefined,/-fsanitize=address,undefined} -fsanitize-coverage=inline-8bit-counters,pc-table" ;; *) F=$FLAGS ;; esac
    /opt/rocm/bin/hipcc $MODE $F -w -c "$src" -o "$WORK/obj/$f.o"
  fi
done
# what these objects reference from the kernel translation units that are not part of this build (never reached without a GPU)
cat > "$WORK/stubs.cpp" <<'STUB'
#include <stddef.h>
#include <stdint.h>
struct r0h_ctx; struct r0h_buf;
namespace r0h { const char* ntt_init_device() { return nullptr; } void session_rows_free(r0h_ctx*) {} }  // (session.cpp needs a device)
struct r0h_circuit;
namespace r0h { const char* logup_accum(r0h_ctx*, const r0h_circuit*, uint32_t, const r0h_buf*, const r0h_buf*, const uint32_t*, const uint32_t*, r0h_buf*) { __builtin_trap(); } }
extern "C" const char* r0h_prefix_products(r0h_ctx*, r0h_buf*, uint32_t) { __builtin_trap(); }
extern "C" const char* r0h_logup_totals(r0h_ctx*, const r0h_circuit*, uint32_t, const r0h_buf*, const r0h_buf*, uint32_t*) { __builtin_trap(); }
extern "C" const char* r0h_prove_segment(r0h_ctx*, const r0h_circuit*, uint32_t, const r0h_buf*, const r0h_buf*, const uint32_t*, uint32_t*, size_t, size_t*) { __builtin_trap(); }
STUB
$CLANG $FLAGS -c "$WORK/stubs.cpp" -o "$WORK/obj/stubs.o"
# the harness itself carries ASan only: with UBSan on this file too, ASan's start-up check trips over two merged string literals
$CLANG -O1 -g -fsanitize=address,fuzzer-no-link -I"$ROOT/include" -c "$ROOT/tools/fuzz/fuzz_host.cpp" -o "$WORK/harness.o"
$CLANG -fsanitize=address,undefined,fuzzer "$WORK/harness.o" "$WORK"/obj/*.o -L/opt/rocm/lib -lamdhip64 -lhiprtc -Wl,-rpath,/opt/rocm/lib -o "$WORK/fuzz_host"
python3 "$ROOT/tools/fuzz/make_seeds.py" "$WORK/corpus"
cd "$WORK"
R0H_FUZZ_ROOT=$ROOT ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1:detect_odr_violation=0 ./fuzz_host corpus $([ "$JOBS" -gt 1 ] && echo -fork=$JOBS) -max_total_time=$SECS -timeout=30 -rss_limit_mb=6000 -max_len=400000 -print_final_stats=1
