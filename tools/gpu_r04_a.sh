#!/bin/bash
# round 4, session A: correctness of the ring-form Cayley logarithm / invariants-only fp64 forward / scalar-operand fp64
# constants (whole -m gpu suite on the main library), then the same-box A/B of the variants.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
export MM_AB_CASES="pdist 3 5000 f64 0.1;pdist 3 5000 f64 0.35;pdist 3 5000 f64 0.6;pdist 3 5000 f32 0.6;pdist 3 5000 f32 0.1"
export MM_AB_ROUNDS=2
bash tools/gpu_ab_libs.sh mat ring sconst sconst3 2>&1 | tee $OUT/ab.txt
