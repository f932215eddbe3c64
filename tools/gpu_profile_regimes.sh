#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the SPD(3) pair kernels outside the
# headline regime — mid-training spread (||log X|| = 0.35), fp64, and the fused training step.
# Usage: tools/gpu_profile_regimes.sh <tag>   -> gpurun_out/<tag>_{mid32,mid64,init64,stepf,stepu}/
TAG=${1:-reg}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
$P -d $OUT/${TAG}_mid32 -o s -- python3 /root/repo/tools/bench_configs.py --only n5000_f32_mid > /dev/null 2>&1
$P -d $OUT/${TAG}_mid64 -o s -- python3 /root/repo/tools/bench_configs.py --only n5000_f64_mid > /dev/null 2>&1
$P -d $OUT/${TAG}_init64 -o s -- python3 /root/repo/tools/bench_configs.py --only spd3_n5000_f64 > /dev/null 2>&1
$P -d $OUT/${TAG}_stepf -o s -- python3 /root/repo/tools/step_profile.py > /dev/null 2>&1
$P -d $OUT/${TAG}_stepu -o s -- python3 /root/repo/tools/step_profile.py --unfused > /dev/null 2>&1
for d in mid32 mid64 init64 stepf stepu; do echo "== $d"; grep "mm::" $OUT/${TAG}_$d/s_kernel_stats.csv | cut -c1-60,200-400 | head -6; done
