#!/bin/bash
# round 4, session G: graphembed.linalg.fast on the GPU, the two-tier fp64 Cayley tables against the one-tier build, and how
# much of a step is the boundary between two graph launches (tools/graph_unroll_probe.py).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_linalg_fast.py tests/test_spd_gpu.py tests/test_symeig.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
python tools/graph_unroll_probe.py all 2>&1 | grep -v amdgpu.ids | tee $OUT/graph_unroll.txt
export MM_AB_ROUNDS=3
export MM_AB_CASES="pdist 3 5000 f64 0.35;pdist 3 5000 f64 0.6;pdist 3 5000 f64 0.1"
bash tools/gpu_ab_libs.sh main onetier 2>&1 | tee $OUT/ab_two_tier.txt
