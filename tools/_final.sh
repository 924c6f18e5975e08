cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
bash tools/rocprof_round.sh r05 > gpurun_out/rocprof_round_r05.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > gpurun_out/gputest_r05_final.log
python __graft_entry__.py smoke > gpurun_out/smoke_final.log 2>&1
