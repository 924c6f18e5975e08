#!/bin/bash
# GPU box: memory-unit / issue counters for tools/conv_microbench.py (args passed through) -> gpurun_out/pmc_micro.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/conv_microbench.py 3 $@"
rm -rf /tmp/m1 /tmp/m2 /tmp/m3 /tmp/m4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d /tmp/m1 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum -d /tmp/m2 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d /tmp/m3 --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/m4 --output-format csv -- $CMD > /dev/null 2>&1
python $R/tools/pmc_sq_summary.py /tmp/m1 /tmp/m2 /tmp/m3 /tmp/m4 > $R/gpurun_out/pmc_micro.txt
