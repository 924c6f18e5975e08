cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "sixteen_input or test_conv2d_forward or narrow16" 2>&1 | tail -12 > gpurun_out/gputest_r05_k.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "cfg4 or cfg5_generator" 2>&1 | tail -8 >> gpurun_out/gputest_r05_k.log
export DL4DS_BENCH_BREAKDOWN=1
for c in cfg4 cfg5; do timeout 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${c}_k.json 2> gpurun_out/bench_${c}_k.err; done
