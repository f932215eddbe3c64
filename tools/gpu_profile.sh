#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of bench.py.
# Usage: tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>_*.csv
TAG=${1:-prof}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $B --steps 20 --warmup 5 > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- $B --steps 3 --warmup 1 --no-prof > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- $B --steps 3 --warmup 1 --no-prof > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/${TAG}_pmc_write -o p -- $B --steps 3 --warmup 1 --no-prof > $OUT/${TAG}_pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
find $OUT -name "*kernel_stats.csv" -exec head -12 {} \;
