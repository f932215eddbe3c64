#!/bin/bash
# round 4, session O: the SPD Jacobi fallback as a one-sided Jacobi on B = L_i^-1 L_j (d <= 4): accuracy on ill-conditioned points
# (tools/illcond_probe.py) and speed, against the two-sided route (libmm_twosided.so); SPD test suites.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04o
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants
{ echo "== one-sided (tree)"; python tools/illcond_probe.py 2>&1 | grep -v amdgpu; echo "== two-sided on A = B B^T (round 1-3)"; MM_MANIFOLDS_LIB=$V/libmm_twosided.so python tools/illcond_probe.py 2>&1 | grep -v amdgpu; } | tee $OUT/illcond.txt
timeout 1500 python -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_reference_suite_gpu.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
export MM_AB_ROUNDS=2
export MM_AB_CASES="pdist 3 5000 f32 0.1;pdist 3 5000 f32 0.6;pdist 3 5000 f32 1.5;pdist 3 5000 f32 3.0;pdist 3 5000 f64 0.35;pdist 4 2274 f32 0.6;pdist 4 2274 f32 1.5"
bash tools/gpu_ab_libs.sh main twosided 2>&1 | tee $OUT/ab_onesided.txt
