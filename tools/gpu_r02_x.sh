#!/bin/bash
# symmetric matrix-core backward: 1 / 2 / 4 workgroups per super-tile (same box), parity for the split forms
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for P in 2 4; do
  MM_GRAM_BWD_PARTS=$P timeout 600 python -m pytest tests/test_vec_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $OUT/pytest_p$P.log 2>&1; echo "parts=$P pytest rc=$?"; tail -1 $OUT/pytest_p$P.log
done
cd /tmp && export TMPDIR=/tmp
for P in 1 2 4 1 2; do
  for CASE in "vec 11 4039 f32 lorentz" "vec 6 5000 f32 sphere"; do
    NAME=$(echo $CASE | tr ' .' '__')_p$P
    MM_GRAM_BWD_PARTS=$P rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME} -o s -- python3 /root/repo/tools/profile_case.py $CASE 40 > /dev/null 2>&1
    python3 - $OUT/${NAME}/s_kernel_stats.csv "parts=$P $CASE" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gram_bwd' in r['Name']: print(sys.argv[2], r['Name'][9:50], 'avg %.1f us' % (float(r['AverageNs']) / 1e3), 'min %.1f' % (float(r['MinNs']) / 1e3))
PY
  done
done
