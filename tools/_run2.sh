cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/gputest_r05_b.log
for c in cfg2 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${c}_b.json 2> gpurun_out/bench_${c}_b.err; done
