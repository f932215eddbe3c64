#!/bin/bash
# Gram forward with dwordx4 stores through LDS: parity + timeline + stats
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02y
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_vec_gpu.py tests/test_configs_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_gstamp.so python3 tools/gram_timeline.py 4039 11 fwd 2>&1 | grep -v amdgpu.ids
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
