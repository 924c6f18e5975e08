cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/parity_r05.json
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/gputest_r05_final.log
