#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02 [tag]
# 1. kernel trace + stats of the bench command on one context  -> profiles/<round>/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
# 2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes   -> profiles/<round>/pmc_traffic.json (tools/summarize_pmc.py)
# 3. SQ counters in two passes                                   -> profiles/<round>/pmc_valu.json (tools/summarize_pmc_valu.py)
# The program goes directly after `--` (no env/bash hop under the profiler).  Writes scratch under gpurun_out/prof_<round>/.
set -o pipefail
ROUND=${1:-r02}; TAG=${2:-a}
OUT=gpurun_out/prof_$ROUND; DST=gpurun_out/profiles_$ROUND
mkdir -p $OUT $DST
export TMPDIR=/tmp
find_csv() { find "$1" -name "*$2" | head -1; }

# round 4: the command is the headline itself -- sessions of the camt53 guest over the trace circuit (one session in flight, so that the
# per-kernel table is one session's; --profile-mode leaves the synthetic secondary runs out)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --cpu-po2 0 --contexts 1 --profile-mode > $DST/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err || exit 11
cp "$(find_csv $OUT/stats kernel_stats.csv)" $DST/${TAG}_kernel_stats.csv || exit 12

CMD="python3 bench.py --steps 1 --warmup 1 --cpu-po2 0 --contexts 1 --profile-mode"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.out 2> $OUT/fetch.err || exit 21
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.out 2> $OUT/write.err || exit 22
python3 tools/summarize_pmc.py "$(find_csv $OUT/fetch counter_collection.csv)" "$(find_csv $OUT/write counter_collection.csv)" $DST/pmc_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over '$CMD' (3 sessions of the camt53 guest = 36 segments: warm-up, timed, accounting pass; every session commits the CODE group of its two trace sizes itself), po2 = 20, trace circuit." 36 '{}' trace || exit 23

rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq1 -- $CMD > $OUT/sq1.out 2> $OUT/sq1.err || exit 31
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.out 2> $OUT/sq2.err || exit 32
python3 tools/summarize_pmc_valu.py "$(find_csv $OUT/sq1 counter_collection.csv)" "$(find_csv $OUT/sq2 counter_collection.csv)" $DST/pmc_valu.json "rocprofv3 SQ counters in two passes over the same command." || exit 33
echo "profiles collected into $DST"
