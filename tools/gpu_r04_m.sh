#!/bin/bash
# round 4, session M: the randomised campaigns with fresh seeds on the round's new paths (node minibatches inside the single
# factors' pair kernels, two-tier fp64 Cayley tables, product kernels with the kinds as template arguments, linalg.fast).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 1500 python "$@" 2>&1 | grep -v amdgpu.ids | tail -12; }
{
run tools/fuzz_product.py 400 41001
run tools/fuzz_product.py 400 41002 --single
run tools/fuzz_product.py 30 41003 --big
run tools/fuzz_product.py 30 41004 --big --single
run tools/fuzz_step.py 250 41005
run tools/fuzz_step.py 20 41006 --big
run tools/fuzz_graph.py 120 41007
run tests/fuzz_pdist.py 500 41008
run tests/fuzz_pdist.py 30 41009 --big
run tests/fuzz_optim.py 150 41010
run tests/fuzz_maps.py 150 41011
run tests/fuzz_misc.py 100 41012
run tests/fuzz_metrics.py 100 41013
} | tee $OUT/fuzz.txt
