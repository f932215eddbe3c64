#!/bin/bash
# round 4, session C: mixed-manifold pair kernel at csphd size (pad width 8, closed-form SPD(2), rows per wavefront), the
# subset vector kernel's launch shape, shard balance of config 5 under cut policies, bench.py's new fields.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
prod() {  # label, env...
  local L=$1; shift
  env "$@" true
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod_$L -o s -- $C product 1025 f32 60 > /dev/null 2>&1 )
  python3 - $OUT/prod_$L/s_kernel_stats.csv "product n=1025 f32 $L" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    nm = r['Name']
    if 'product_' in nm and int(r['Calls']) > 10:
        out.append('%s avg %.1f min %.1f (x%s)' % (nm.split('(')[0].replace('void mm::', '')[:48], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
}
for R in 1 2; do
  prod default_$R MM_DUMMY=1
  prod pw16_$R MM_PRODUCT_PW16=1
  for TI in 5 6 7 9 10 12; do prod ti${TI}_$R MM_PRODUCT_TI=$TI; done
  prod sym_$R MM_PRODUCT_SYM=1
done 2>&1 | tee $OUT/product.txt
cd $GRAFT_REPO_ROOT
python tools/bench_configs.py --only lorentz24_minibatch512 > $OUT/lorentz24.json 2>/dev/null; cat $OUT/lorentz24.json
for K in 0 100000; do for TC in 0 1; do
  MM_SHARD_K=$K MM_SPD4_BWD_TWO_COLS=$TC python3 tools/shard_balance.py 8
done; done 2>/dev/null | tee $OUT/shard_balance.txt
for K in 0 100000; do MM_SHARD_K=$K python3 tools/shard_balance.py 4; done 2>/dev/null | tee -a $OUT/shard_balance.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err
python3 - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline ms', d['ms_per_step'], 'median sync ms', d.get('ms_per_step_median_synchronised'))
print('roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'valu_issue_frac', 'shader_clock_mhz', 'traffic')})
print('roofline_f64', {k: (d.get('roofline_f64') or {}).get(k) for k in ('frac', 'avg_launch_us', 'ms_per_step')})
print('roofline_f64_mid', {k: (d.get('roofline_f64_mid_training') or {}).get(k) for k in ('frac', 'avg_launch_us', 'ms_per_step')})
for e in d.get('extra', []):
    print('  %-110s %.4f ms' % (e['workload'][:110], e['ms_per_step']))
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
