cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batchprep.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "external_lr or eight_input or conv_lstm2d or test_conv2d_forward or narrow16" 2>&1 | tail -30 > gpurun_out/gputest_r05_d.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "cfg4" 2>&1 | tail -15 >> gpurun_out/gputest_r05_d.log
export DL4DS_BENCH_BREAKDOWN=1
for c in cfg2 cfg4; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-unfolded > gpurun_out/bench_${c}_d.json 2> gpurun_out/bench_${c}_d.err; done
DL4DS_NARROW16_NO_C8=1 timeout 300 python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4_d_noc8.json 2>/dev/null
