cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/gputest_r05_final.log
python __graft_entry__.py smoke > gpurun_out/smoke_final.log 2>&1
timeout 400 python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
