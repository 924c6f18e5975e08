cd /root/repo; mkdir -p gpurun_out
DL4DS_HIP_LIB=/root/repo/dl4ds_amd/libdl4ds_hip_exp.so DL4DS_ADD_DEBUG=1 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 --no-profile > /dev/null 2> gpurun_out/add_debug_cfg5.txt
