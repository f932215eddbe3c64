#!/bin/bash
# after the scalar-side trims: SPD + configs + round-2 parity, rocprof stats of the headline and config-5 cases
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02t
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_round2_gpu.py tests/test_c_abi.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
show() { python3 - $1 <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pdist' in r['Name'] or 'gram' in r['Name']: print('  ', r['Name'][9:75], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3), 'min %.1f' % (float(r['MinNs']) / 1e3))
PY
}
for CASE in "pdist 3 5000 f32 0.1" "pdist 3 5000 f32 0.35" "pdist 4 16384 f32 0.1" "loss 4 16384 f32" "pdist 3 5000 f64 0.1" "vec 11 4039 f32 lorentz"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME} -o s -- $C $CASE 40 > /dev/null 2>&1
  echo "== $CASE"; show $OUT/${NAME}/s_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['per_rank'][0]['fwd_kernel_us'], d['per_rank'][0]['bwd_kernel_us'], d['roofline']['frac'])
for e in d['extra']: print(e['workload'][:60], e['ms_per_step'])
"
