cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/gputest_r05_f.log
timeout 400 python bench.py > gpurun_out/bench_default_f.json 2> gpurun_out/bench_default_f.err
export DL4DS_BENCH_BREAKDOWN=1
timeout 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5_f.json 2> gpurun_out/bench_cfg5_f.err
DL4DS_SEQ_TRACE=1 timeout 300 python bench.py --config cfg4 --no-cpu-baseline --steps 2 --warmup 1 --no-profile > /dev/null 2> gpurun_out/seq_trace_f.txt
