#!/bin/bash
# Round-end evidence: the whole GPU suite, then tools/gpu_profile_r02.sh (kernel stats, PMC passes, stamped pmc_head.json,
# a default `python bench.py` run), the backward's workgroup timeline and the layout A/B microbenchmark.
TAG=${1:-r02v2}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest_gpu.log
bash tools/gpu_profile_r02.sh $TAG > gpurun_out/$TAG/profile.log 2>&1
tail -c 1500 gpurun_out/$TAG/profile.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python tools/bench_configs.py > gpurun_out/$TAG/configs.json 2> gpurun_out/$TAG/configs.err; echo "configs rc=$?"
