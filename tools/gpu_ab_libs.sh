#!/bin/bash
# same-box A/B of library variants (matrix-manifolds_amd/lib/variants/libmm_NAME.so): tools/gpu_ab_libs.sh NAME1 NAME2 ... ;
# alternates the variants three times on the headline case and prints the rocprofv3 averages
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_libs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ROUND in 1 2 3; do
  for V in "$@"; do
    for CASE in "pdist 3 5000 f32 0.1" "pdist 3 5000 f32 0.35"; do
      NAME=$(echo $CASE | tr ' .' '__')_${V}_$ROUND
      MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_$V.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME -o s -- python3 /root/repo/tools/profile_case.py $CASE 60 > /dev/null 2>&1
      python3 - $OUT/$NAME/s_kernel_stats.csv "$V round $ROUND: $CASE" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'pdist_bwd' in r['Name'] or 'pdist_fwd' in r['Name']:
        out.append('%s avg %.1f min %.1f' % ('bwd' if 'bwd' in r['Name'] else 'fwd', float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
print(sys.argv[2], '|', '; '.join(out))
PY
    done
  done
done
