#!/bin/bash
# GPU box: launch-ordered kernel timeline of one bench.py train step -> gpurun_out/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl --output-format csv -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-profile --no-unfolded "$@" > /dev/null 2>&1
python $R/tools/step_timeline.py /tmp/tl ${TL_ADAMS:-1} > $R/gpurun_out/timeline.txt
