mkdir -p gpurun_out/base
python __graft_entry__.py smoke > gpurun_out/base/smoke.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/base/cfg2.json 2> gpurun_out/base/cfg2.err
python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/base/cfg4.json 2> gpurun_out/base/cfg4.err
python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/base/cfg5.json 2> gpurun_out/base/cfg5.err
python bench.py --predict --no-cpu-baseline > gpurun_out/base/predict.json 2> gpurun_out/base/predict.err
tail -2 gpurun_out/base/smoke.log
