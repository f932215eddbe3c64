O=gpurun_out/r06fz2; mkdir -p $O
run() { echo "== $*" >> $O/campaign.txt; "$@" 2>&1 | grep -v "^ok \|amdgpu.ids" | tail -4 >> $O/campaign.txt; echo "rc=${PIPESTATUS[0]}" >> $O/campaign.txt; }
run python3 tools/fuzz_walk.py 2500 3901
run python3 tests/fuzz_pdist.py 2500 3902
run python3 tests/fuzz_pdist.py 40 3903 --big
run python3 tools/fuzz_product.py 600 3904
run python3 tools/fuzz_product.py 300 3905 --single
run python3 tools/fuzz_product.py 40 3906 --big
run python3 tools/fuzz_step.py 600 3907
run python3 tools/fuzz_step.py 20 3908 --big
run python3 tools/fuzz_graph.py 200 3909
run python3 tests/fuzz_maps.py 300 3910
run python3 tests/fuzz_misc.py 300 3911
run python3 tests/fuzz_optim.py 300 3912
run python3 tests/fuzz_metrics.py 100 3913
cat $O/campaign.txt
