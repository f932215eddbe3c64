#!/bin/bash
# Gram forward with 256-B row segments: parity (vector tests, configs, fuzz slices) + rocprof stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_vec_gpu.py tests/test_configs_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for CASE in "vec 11 4039 f32 lorentz" "vec 6 5000 f32 sphere" "vec 32 3000 f32 lorentz"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME} -o s -- python3 /root/repo/tools/profile_case.py $CASE 40 > /dev/null 2>&1
  echo "== $CASE"; python3 - $OUT/${NAME}/s_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gram' in r['Name']: print('  ', r['Name'][9:75], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3), 'min %.1f' % (float(r['MinNs']) / 1e3))
PY
done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 /root/repo/tools/profile_case.py vec 11 4039 f32 lorentz 3 > /dev/null 2>&1
python3 - $OUT/pmc_write <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gram_fwd' in r['Kernel_Name']: print('fwd WRITE_SIZE KiB', r['Counter_Value']); break
PY
