#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of `bench.py --config <cfg>` (BASELINE configs[3] / configs[4] at full size).
# Usage: tools/rocprof_cfg.sh cfg5 r02   -> gpurun_out/prof_<tag>/kernel_stats_<cfg>_<tag>.txt + bench_line_<cfg>_<tag>.json
CFG=${1:-cfg5}
TAG=${2:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $CFG --steps 10 --warmup 5 --no-cpu-baseline --no-unfolded --no-b16"
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats_$CFG --output-format csv -- $CMD > $OUT/bench_under_rocprof_$CFG.log 2>&1
python $R/tools/rocprof_stats_summary.py /tmp/prof_stats_$CFG > $OUT/kernel_stats_${CFG}_$TAG.txt
grep '^{"metric"' $OUT/bench_under_rocprof_$CFG.log | tail -1 > $OUT/bench_line_${CFG}_$TAG.json
