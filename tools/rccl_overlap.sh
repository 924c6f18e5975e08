#!/bin/bash
# GPU box: kernel trace of a 1-rank-communicator run (DL4DS_FORCE_DIST=1: the RCCL path with one rank) of bench.py
# (with the stand-in kernel of csrc/dist.cpp after every bucket launch: a 1-rank in-place all-reduce launches no kernel)
# -> gpurun_out/rccl_overlap_<cfg>.txt (copy to profiles/rccl_overlap_rNN.txt).  Usage: tools/rccl_overlap.sh cfg2|cfg5
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-cfg2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ov_$CFG
DL4DS_TEST_HOOKS=1 DL4DS_FORCE_DIST=1 DL4DS_DIST_STANDIN=${STANDIN:-1} rocprofv3 --kernel-trace -d /tmp/ov_$CFG --output-format csv -- python $R/bench.py --config $CFG --steps 3 --warmup 2 \
    --no-cpu-baseline --no-profile --no-unfolded > $R/gpurun_out/rccl_overlap_${CFG}_bench.json 2> $R/gpurun_out/rccl_overlap_${CFG}.err
python $R/tools/rccl_overlap.py /tmp/ov_$CFG $CFG > $R/gpurun_out/rccl_overlap_$CFG.txt
