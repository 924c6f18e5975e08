#!/bin/bash
# GPU box: round 6's evidence in two gpurun calls (the second after `python tools/build_traffic_json.py r06` has put the first one's PMC
# traffic into profiles/traffic.json, so that the default bench line carries roofline.traffic):
#   tools/evidence_r06.sh 1   GPU tests + smoke + tools/rocprof_round.sh r06 + the predict line + a DL4DS_SPLIT=1 line -> gpurun_out/prof_r06/
#   tools/evidence_r06.sh 2   GPU tests (parity_r06.json) + the default line + the DL4DS_SPLIT=1 line + step gaps      -> gpurun_out/prof_r06b/, gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
if [ "${1:-1}" == "1" ]; then
  O=gpurun_out/prof_r06; mkdir -p $O
  ( python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 ; tail -3 $O/gpu_tests.log )
  python __graft_entry__.py smoke > $O/smoke.log 2>&1
  bash tools/rocprof_round.sh r06 > $O/round.log 2>&1
  python bench.py --predict > $O/bench_predict_r06.json 2> $O/bench_predict_r06.err
  DL4DS_SPLIT=1 python bench.py --no-cpu-baseline --no-unfolded --no-b16 > $O/bench_split_r06.json 2> $O/bench_split_r06.err
  DL4DS_NO_SPLIT=1 python bench.py --no-cpu-baseline --no-unfolded --no-b16 > $O/bench_nosplit_r06.json 2> $O/bench_nosplit_r06.err
else
  O=gpurun_out/prof_r06b; mkdir -p $O
  # (SKIP_TESTS=1: call 1 has already run the GPU suite on these kernel sources)
  [ -n "$SKIP_TESTS" ] || ( python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 ; tail -3 $O/gpu_tests.log )
  python bench.py > $O/bench_default_r06.json 2> $O/bench_default_r06.err
  DL4DS_SPLIT=1 python bench.py --no-cpu-baseline --no-unfolded --no-b16 > $O/bench_split_r06.json 2> $O/bench_split_r06.err
  DL4DS_NO_SPLIT=1 python bench.py --no-cpu-baseline --no-unfolded --no-b16 > $O/bench_nosplit_r06.json 2> $O/bench_nosplit_r06.err
  for c in cfg2 cfg4 cfg5; do bash tools/step_gaps.sh $c > /dev/null 2>&1; done
fi
ls $O
