#!/bin/bash
# round 4, session L: XCD-aware tile order of the ordered mixed-manifold pair kernel (time, FETCH / WRITE counters), product
# parity suites, the SPD(4) fp64 case of the mid-distance regime test.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_spd_gpu.py tests/test_minibatch_golden.py tests/test_configs_gpu.py tests/test_fused_step_gpu.py -m gpu -x -q -k "recentred or product or config4 or minibatch or tree40" > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
for R in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod_$R -o s -- $C product 1025 f32 60 > /dev/null 2>&1
  python3 - $OUT/prod_$R/s_kernel_stats.csv "XCD-aware order, round $R: product n=1025 f32" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'product_' in r['Name'] and int(r['Calls']) > 10:
        out.append('%s avg %.2f min %.2f (x%s)' % (r['Name'].split('(')[0].replace('void mm::', '')[:56], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
done | tee $OUT/product_xcd.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prod_fetch -o p -- $C product 1025 f32 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/prod_write -o p -- $C product 1025 f32 3 > /dev/null 2>&1
python3 - $OUT <<'PY' | tee -a $OUT/product_xcd.txt
import csv, glob, sys, collections
for kind in ('fetch', 'write'):
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + f'/prod_{kind}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'product_pair_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{k}: {sum(v) / len(v):.1f} per launch' + (' KiB (x2 on gfx950 for FETCH_SIZE)' if 'SIZE' in k else ''))
PY
