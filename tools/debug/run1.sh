mkdir -p gpurun_out/r1
export DL4DS_SPLIT=1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ps_split
rocprofv3 --kernel-trace --stats -d /tmp/ps_split --output-format csv -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-unfolded --no-b16 > $R/gpurun_out/r1/bench_under_rocprof_split.log 2>&1
python $R/tools/rocprof_stats_summary.py /tmp/ps_split > $R/gpurun_out/r1/kernel_stats_split_r06.txt
cd $R
bash tools/pmc_sq.sh
cp gpurun_out/pmc_sq.txt gpurun_out/r1/pmc_sq_split_r06.txt
python tools/pmc_derived.py gpurun_out/pmc_sq.txt > gpurun_out/r1/pmc_mfma_split_r06.txt
head -12 gpurun_out/r1/kernel_stats_split_r06.txt | cut -c1-130
