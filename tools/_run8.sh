cd /root/repo; mkdir -p gpurun_out
bash tools/rocprof_round.sh r05 > gpurun_out/rocprof_round_r05.log 2>&1
