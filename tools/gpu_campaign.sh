O=gpurun_out/r05fz5; mkdir -p $O
run() { echo "== $*" >> $O/campaign.txt; "$@" 2>&1 | grep -v "^ok \|amdgpu.ids" | tail -4 >> $O/campaign.txt; echo "rc=${PIPESTATUS[0]}" >> $O/campaign.txt; }
run python3 tools/fuzz_walk.py 2500 3301
run python3 tests/fuzz_pdist.py 2500 3302
run python3 tests/fuzz_pdist.py 40 3303 --big
run python3 tools/fuzz_product.py 600 3304
run python3 tools/fuzz_product.py 300 3305 --single
run python3 tools/fuzz_product.py 40 3306 --big
run python3 tools/fuzz_step.py 600 3307
run python3 tools/fuzz_step.py 20 3308 --big
run python3 tools/fuzz_graph.py 200 3309
run python3 tests/fuzz_maps.py 300 3310
run python3 tests/fuzz_misc.py 300 3311
run python3 tests/fuzz_optim.py 300 3312
run python3 tests/fuzz_metrics.py 100 3313
cat $O/campaign.txt
