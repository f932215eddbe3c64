#!/bin/bash
# round 4, session R: a long pass of the randomised campaigns on the final library (seeds 58001...)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 3000 python "$@" 2>&1 | grep -v "amdgpu.ids\|will be ignored\|^ok case\|^skipped" | tail -6; }
{
run tests/fuzz_pdist.py 4000 58001
run tests/fuzz_misc.py 2500 58002
run tests/fuzz_maps.py 1200 58003
run tests/fuzz_optim.py 1000 58004
run tests/fuzz_metrics.py 600 58005
run tools/fuzz_product.py 2000 58006
run tools/fuzz_product.py 2000 58007 --single
run tools/fuzz_step.py 2000 58008
run tools/fuzz_graph.py 800 58009
run tools/fuzz_product.py 80 58010 --big
run tools/fuzz_step.py 60 58011 --big
run tests/fuzz_pdist.py 150 58012 --big
} | tee $OUT/fuzz.txt
