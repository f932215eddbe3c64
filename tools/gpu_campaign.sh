O=gpurun_out/r05fz4; mkdir -p $O
run() { echo "== $*" >> $O/campaign.txt; "$@" 2>&1 | grep -v "^ok \|amdgpu.ids" | tail -4 >> $O/campaign.txt; echo "rc=${PIPESTATUS[0]}" >> $O/campaign.txt; }
run python3 tools/fuzz_walk.py 2500 1201
run python3 tests/fuzz_pdist.py 2500 1202
run python3 tests/fuzz_pdist.py 40 1203 --big
run python3 tools/fuzz_product.py 600 1204
run python3 tools/fuzz_product.py 300 1205 --single
run python3 tools/fuzz_product.py 40 1206 --big
run python3 tools/fuzz_step.py 600 1207
run python3 tools/fuzz_step.py 20 1208 --big
run python3 tools/fuzz_graph.py 200 1209
run python3 tests/fuzz_maps.py 300 1210
run python3 tests/fuzz_misc.py 300 1211
run python3 tests/fuzz_optim.py 300 1212
run python3 tests/fuzz_metrics.py 100 1213
cat $O/campaign.txt
