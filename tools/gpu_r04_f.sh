#!/bin/bash
# round 4, session F: the whole GPU suite after the tolerance note of the forced-ordered product case, then a same-box A/B of
# the regular library against the max-ILP-scheduled build (libmm_ilp.so: spd.hip + spd_loss.hip with
# -mllvm -amdgpu-sched-strategy=max-ilp) on the SPD cases of every dtype / dimension.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -a "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -8
export MM_AB_ROUNDS=2
export MM_AB_CASES="pdist 3 5000 f32 0.1;pdist 3 5000 f32 0.35;pdist 3 5000 f64 0.1;pdist 3 5000 f64 0.35;pdist 4 16384 f32 0.1;pdist 4 2274 f32 0.1"
bash tools/gpu_ab_libs.sh main ilp 2>&1 | tee $OUT/ab_ilp.txt
