#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of bench.py + separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# Usage: tools/rocprof_round.sh r01   -> writes summaries under gpurun_out/prof_<tag>/
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unfolded"
rocprofv3 --kernel-trace --stats -d /tmp/prof_stats --output-format csv -- $CMD > $OUT/bench_under_rocprof.log 2>&1
python $R/tools/rocprof_stats_summary.py /tmp/prof_stats > $OUT/kernel_stats_$TAG.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_fetch --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_write --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/prof_fetch /tmp/prof_write > $OUT/pmc_traffic_$TAG.json
grep '^{"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line_$TAG.json
ls -la $OUT
