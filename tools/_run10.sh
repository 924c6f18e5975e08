cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_trainers.py -m gpu -q -p no:cacheprovider -k "cgan or cfg5 or discriminator" 2>&1 | tail -15 > gpurun_out/gputest_r05_h.log
export DL4DS_BENCH_BREAKDOWN=1
timeout 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5_h.json 2> gpurun_out/bench_cfg5_h.err
DL4DS_TEST_HOOKS=1 DL4DS_NO_TWO_ADD_INPLACE=1 timeout 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5_h_off.json 2>/dev/null
