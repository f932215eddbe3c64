#!/bin/bash
# round 4, session P: the randomised campaigns once more on the FINAL library (second solve for ill-conditioned fp32 pairs, one-sided
# Jacobi for the Grassmann angles, kinds as template arguments, minibatch fixes), seeds 56001...
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 2400 python "$@" 2>&1 | grep -v "amdgpu.ids\|will be ignored\|^ok case\|^skipped" | tail -6; }
{
run tests/fuzz_pdist.py 1500 56001
run tests/fuzz_pdist.py 60 56002 --big
run tests/fuzz_misc.py 800 56003
run tests/fuzz_maps.py 400 56004
run tests/fuzz_optim.py 300 56005
run tests/fuzz_metrics.py 200 56006
run tools/fuzz_product.py 600 56007
run tools/fuzz_product.py 600 56008 --single
run tools/fuzz_product.py 30 56009 --big --single
run tools/fuzz_step.py 700 56010
run tools/fuzz_step.py 25 56011 --big
run tools/fuzz_graph.py 300 56012
} | tee $OUT/fuzz.txt
