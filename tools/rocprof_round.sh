#!/bin/bash
# GPU box: a round's evidence in one call:  tools/rocprof_round.sh r05
#   tools/rocprof_all.sh <tag> (kernel stats + FETCH_SIZE / WRITE_SIZE passes for three configs + the default bench line), the
#   fingerprint of the kernel sources the counters ran on (bench.py: csrc_sha -> profiles/traffic.json), SQ counters per config
#   (three separate --pmc passes each, no tracing flags), the derived MFMA-busy tables, a DL4DS_FORCE_DIST=1 line (1-rank RCCL path).
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_$TAG
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_sha())" > $R/gpurun_out/prof_$TAG/csrc_sha.txt
bash $R/tools/rocprof_all.sh $TAG > /dev/null 2>&1
for CFG in cfg2 cfg5 cfg4; do
  bash $R/tools/pmc_sq.sh --config $CFG
  cp $R/gpurun_out/pmc_sq.txt $R/gpurun_out/prof_$TAG/pmc_sq_${CFG}_$TAG.txt
  python $R/tools/pmc_derived.py $R/gpurun_out/pmc_sq.txt > $R/gpurun_out/prof_$TAG/pmc_mfma_${CFG}_$TAG.txt
done
( cd $R && DL4DS_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-unfolded --no-b16 > gpurun_out/prof_$TAG/bench_force_dist_$TAG.json 2>/dev/null )
ls $R/gpurun_out/prof_$TAG
