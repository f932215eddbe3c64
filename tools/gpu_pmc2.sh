#!/bin/bash
# second PMC pass: stall breakdown of the pair kernels
TAG=${1:-pmc2}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-prof --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/${TAG}_a -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_b -o p -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/${TAG}_a gpurun_out/${TAG}_b
