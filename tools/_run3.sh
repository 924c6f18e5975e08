cd /root/repo; mkdir -p gpurun_out
export DL4DS_BENCH_BREAKDOWN=1
A="--config cfg2 --no-cpu-baseline --no-unfolded"
timeout 200 python bench.py $A > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
DL4DS_WINO_NO_FILTER_CACHE=1 timeout 200 python bench.py $A > gpurun_out/ab_nocache.json 2>/dev/null
DL4DS_WINO_WGRAD_TWO_SUMS=1 timeout 200 python bench.py $A > gpurun_out/ab_twosums.json 2>/dev/null
DL4DS_WINO_NO_FILTER_CACHE=1 DL4DS_WINO_WGRAD_TWO_SUMS=1 timeout 200 python bench.py $A > gpurun_out/ab_old.json 2>/dev/null
timeout 200 python bench.py $A > gpurun_out/ab_new2.json 2>/dev/null
unset DL4DS_BENCH_BREAKDOWN
timeout 900 python -m pytest tests/test_gpu_batchprep.py tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "other_input_forms or external_lr or eight_input or conv_lstm2d" 2>&1 | tail -30 > gpurun_out/gputest_r05_c.log
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "supervised_forward_grads or normalization_and_dropout or random_builder or cgan" 2>&1 | tail -40 >> gpurun_out/gputest_r05_c.log
