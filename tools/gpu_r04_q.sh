#!/bin/bash
# round 4, session Q: a long pass of the randomised campaigns on the final library (seeds 57001...)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04q
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 3000 python "$@" 2>&1 | grep -v "amdgpu.ids\|will be ignored\|^ok case\|^skipped" | tail -6; }
{
run tests/fuzz_pdist.py 4000 57001
run tests/fuzz_misc.py 2500 57002
run tests/fuzz_maps.py 1200 57003
run tests/fuzz_optim.py 1000 57004
run tests/fuzz_metrics.py 600 57005
run tools/fuzz_product.py 2000 57006
run tools/fuzz_product.py 2000 57007 --single
run tools/fuzz_step.py 2000 57008
run tools/fuzz_graph.py 800 57009
run tools/fuzz_product.py 80 57010 --big
run tools/fuzz_step.py 60 57011 --big
run tests/fuzz_pdist.py 150 57012 --big
} | tee $OUT/fuzz.txt
