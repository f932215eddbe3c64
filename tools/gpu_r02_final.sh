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
# same-box A/B of the matrix-core backward: ordered tiles vs symmetric tiles (Lorentz(11), n = 4039)
cd /tmp && export TMPDIR=/tmp
for ORD in 1 0; do
  MM_GRAM_BWD_ORDERED=$ORD rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/lorentz_ord${ORD} -o s -- python3 /root/repo/tools/profile_case.py vec 11 4039 f32 lorentz 40 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - gpurun_out/$TAG <<'PY' > gpurun_out/$TAG/lorentz_ab.txt
import csv, sys
print('Lorentz(11) n = 4039 fp32, rocprofv3 --kernel-trace --stats, 40 launches each, same box:')
for o, name in ((1, 'MM_GRAM_BWD_ORDERED=1 (ordered tiles)'), (0, 'MM_GRAM_BWD_ORDERED=0 (symmetric tiles, default)')):
    for r in csv.DictReader(open(f'{sys.argv[1]}/lorentz_ord{o}/s_kernel_stats.csv')):
        if 'gram' in r['Name']:
            print(f"  {name}: {r['Name'][9:60]:52s} avg {float(r['AverageNs']) / 1e3:6.1f} us  min {float(r['MinNs']) / 1e3:6.1f}")
PY
cat gpurun_out/$TAG/lorentz_ab.txt
timeout 900 python tools/bench_configs.py > gpurun_out/$TAG/configs.json 2> gpurun_out/$TAG/configs.err; echo "configs rc=$?"
