#!/bin/bash
# GPU box: the round's rocprofv3 evidence for all three bench configurations.
#   tools/rocprof_all.sh r03  ->  gpurun_out/prof_r03/{kernel_stats[_cfgN]_r03.txt, bench_line[_cfgN]_r03.json, pmc_traffic_<cfg>_r03.json}
# Per config: one --kernel-trace --stats run, then two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE; no tracing flags).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in cfg2 cfg4 cfg5; do
  SFX=_$CFG; [ $CFG == cfg2 ] && SFX=""
  CMD="python $R/bench.py --config $CFG --steps 10 --warmup 5 --no-cpu-baseline --no-unfolded --no-b16"
  rm -rf /tmp/ps_$CFG /tmp/pf_$CFG /tmp/pw_$CFG
  rocprofv3 --kernel-trace --stats -d /tmp/ps_$CFG --output-format csv -- $CMD > $OUT/bench_under_rocprof$SFX.log 2>&1
  python $R/tools/rocprof_stats_summary.py /tmp/ps_$CFG > $OUT/kernel_stats${SFX}_$TAG.txt
  grep '^{"metric"' $OUT/bench_under_rocprof$SFX.log | tail -1 > $OUT/bench_line${SFX}_$TAG.json
  PCMD="python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-unfolded --no-b16 --no-profile"
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$CFG --output-format csv -- $PCMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$CFG --output-format csv -- $PCMD > /dev/null 2>&1
  python $R/tools/pmc_traffic.py /tmp/pf_$CFG /tmp/pw_$CFG > $OUT/pmc_traffic_${CFG}_$TAG.json
done
python $R/bench.py > $OUT/bench_default_$TAG.json 2> $OUT/bench_default_$TAG.err
ls -la $OUT
