#!/bin/bash
# GPU box: the round-4 evidence in one call: tools/rocprof_all.sh r04 (kernel stats + traffic for three configs + default bench line),
# SQ counters for cfg2 / cfg5 (three separate --pmc passes each), derived MFMA-busy tables.
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/rocprof_all.sh r04 > /dev/null 2>&1
for CFG in cfg2 cfg5 cfg4; do
  bash $R/tools/pmc_sq.sh --config $CFG
  cp $R/gpurun_out/pmc_sq.txt $R/gpurun_out/prof_r04/pmc_sq_${CFG}_r04.txt
  python $R/tools/pmc_derived.py $R/gpurun_out/pmc_sq.txt > $R/gpurun_out/prof_r04/pmc_mfma_${CFG}_r04.txt
done
ls $R/gpurun_out/prof_r04
