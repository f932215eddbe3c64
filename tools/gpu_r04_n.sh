#!/bin/bash
# round 4, session N: second pass of the randomised campaigns (other seeds, more cases per family)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 2400 python "$@" 2>&1 | grep -v "amdgpu.ids\|will be ignored\|^ok case" | tail -8; }
{
run tests/fuzz_misc.py 1500 52001
run tests/fuzz_maps.py 600 52002
run tests/fuzz_pdist.py 1500 52003
run tests/fuzz_metrics.py 400 52004
run tests/fuzz_optim.py 500 52005
run tools/fuzz_product.py 800 52006 --single
run tools/fuzz_product.py 800 52007
run tools/fuzz_step.py 600 52008
run tools/fuzz_graph.py 300 52009
run tests/fuzz_pdist.py 80 52010 --big
} | tee $OUT/fuzz.txt
