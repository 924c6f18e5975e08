cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "conv_lstm2d or narrow16" 2>&1 | tail -15 > gpurun_out/gputest_r05_e.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_batchprep.py -m gpu -q -p no:cacheprovider -k "cfg4 or external_lr" 2>&1 | tail -15 >> gpurun_out/gputest_r05_e.log
export DL4DS_BENCH_BREAKDOWN=1
timeout 300 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4_e.json 2> gpurun_out/bench_cfg4_e.err
timeout 300 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4_e2.json 2>/dev/null
