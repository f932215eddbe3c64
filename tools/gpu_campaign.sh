O=gpurun_out/r06fz1; mkdir -p $O
run() { echo "== $*" >> $O/campaign.txt; "$@" 2>&1 | grep -v "^ok \|amdgpu.ids" | tail -4 >> $O/campaign.txt; echo "rc=${PIPESTATUS[0]}" >> $O/campaign.txt; }
run python3 tools/fuzz_walk.py 2500 3601
run python3 tests/fuzz_pdist.py 2500 3602
run python3 tests/fuzz_pdist.py 40 3603 --big
run python3 tools/fuzz_product.py 600 3604
run python3 tools/fuzz_product.py 300 3605 --single
run python3 tools/fuzz_product.py 40 3606 --big
run python3 tools/fuzz_step.py 600 3607
run python3 tools/fuzz_step.py 20 3608 --big
run python3 tools/fuzz_graph.py 200 3609
run python3 tests/fuzz_maps.py 300 3610
run python3 tests/fuzz_misc.py 300 3611
run python3 tests/fuzz_optim.py 300 3612
run python3 tests/fuzz_metrics.py 100 3613
cat $O/campaign.txt
