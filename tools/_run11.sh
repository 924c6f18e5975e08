cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "attention or chatt or cfg4 or recnet" 2>&1 | tail -8 > gpurun_out/gputest_r05_j.log
export DL4DS_BENCH_BREAKDOWN=1
timeout 300 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4_j.json 2> gpurun_out/bench_cfg4_j.err
