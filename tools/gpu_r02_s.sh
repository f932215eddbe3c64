#!/bin/bash
# A/B: register caps (4 wavefronts per SIMD for fp32 SPD(3)/SPD(4) backward; variant w64: fp64 SPD(3) fwd at 4, bwd at 3)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
show() { python3 - $1 <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pdist' in r['Name']: print('  ', r['Name'][9:75], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3), 'min %.1f' % (float(r['MinNs']) / 1e3))
PY
}
for CASE in "pdist 3 5000 f32 0.1" "pdist 4 16384 f32 0.1" "loss 4 16384 f32" "pdist 4 2274 f32 0.1" "pdist 3 5000 f64 0.1" "pdist 3 5000 f64 0.35"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME} -o s -- $C $CASE 20 > /dev/null 2>&1
  echo "== $CASE (main)"; show $OUT/${NAME}/s_kernel_stats.csv
done
export MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_w64.so
for CASE in "pdist 3 5000 f64 0.1" "pdist 3 5000 f64 0.35"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_w64 -o s -- $C $CASE 20 > /dev/null 2>&1
  echo "== $CASE (w64)"; show $OUT/${NAME}_w64/s_kernel_stats.csv
done
