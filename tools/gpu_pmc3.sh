#!/bin/bash
# third PMC pass: issue / wait / cache counters of the SPD pair kernels (stall diagnosis)
TAG=${1:-pmc3}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-prof --graph off --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_a -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_b -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_c -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_d -o p -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c gpurun_out/${TAG}_d
