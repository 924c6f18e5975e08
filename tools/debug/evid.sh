mkdir -p gpurun_out/prof_r06
( python -m pytest tests -x -q -m gpu > gpurun_out/prof_r06/gpu_tests.log 2>&1 ; tail -3 gpurun_out/prof_r06/gpu_tests.log )
python __graft_entry__.py smoke > gpurun_out/prof_r06/smoke.log 2>&1
bash tools/rocprof_round.sh r06 > gpurun_out/prof_r06/round.log 2>&1
python bench.py --predict > gpurun_out/prof_r06/bench_predict_r06.json 2> gpurun_out/prof_r06/bench_predict_r06.err
ls gpurun_out/prof_r06
