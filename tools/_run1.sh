cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
DL4DS_PARITY_REPORT_ONLY=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/gputest_r05_a.log
for c in cfg2 cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${c}_base.json 2> gpurun_out/bench_${c}_base.err; done
