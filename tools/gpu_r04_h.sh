#!/bin/bash
# round 4, session H: linalg.fast again, fp64 Jacobi sweeps as a loop (libmm_roll64.so) against the unrolled build, and the
# kernel trace of the config-3 training step captured one and two steps per graph (which kernel is slower in the second form?).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_linalg_fast.py -m gpu -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
export MM_AB_ROUNDS=2
export MM_AB_CASES="pdist 3 5000 f64 0.6;pdist 3 5000 f64 0.35;pdist 3 5000 f64 0.1;pdist 4 2274 f64 0.1;pdist 4 2274 f64 0.6"
bash tools/gpu_ab_libs.sh main roll64 2>&1 | tee $OUT/ab_roll64.txt
cd /tmp && export TMPDIR=/tmp
for U in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_u$U -o s -- python3 /root/repo/tools/graph_unroll_probe.py config3 $U > $OUT/c3_u$U.txt 2>&1
  python3 - $OUT/c3_u$U/s_kernel_stats.csv "config 3 step, $U step(s) per graph" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'mm::' in r['Name'] and int(r['Calls']) > 100:
        out.append('%s avg %.1f min %.1f (x%s)' % (r['Name'].split('(')[0].replace('void mm::', '')[:48], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
  grep "us/step" $OUT/c3_u$U.txt
done | tee $OUT/c3_unroll.txt
