cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
bash tools/step_gaps.sh cfg5
bash tools/step_gaps.sh cfg2
timeout 400 python bench.py > gpurun_out/bench_default_g.json 2> gpurun_out/bench_default_g.err
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=40 2>&1 | tail -90 > gpurun_out/gputest_r05_g.log
