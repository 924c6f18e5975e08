#!/bin/bash
# GPU box: SQ counters (MFMA busy, LDS conflicts, wait cycles) per kernel for bench.py -> gpurun_out/pmc_sq.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-unfolded --no-b16 $@"
rm -rf /tmp/sq1 /tmp/sq2 /tmp/sq3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/sq1 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d /tmp/sq2 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d /tmp/sq3 --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_sq_summary.py /tmp/sq1 /tmp/sq2 /tmp/sq3 > $R/gpurun_out/pmc_sq.txt
