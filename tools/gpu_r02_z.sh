#!/bin/bash
# dispatch-attached kernel timing in bench.py vs rocprofv3 of the same command
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_c_abi.py -m gpu -x -q -k "not fuzz" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('bench: step', d['ms_per_step']*1e3, 'fwd', d['per_rank'][0]['fwd_kernel_us'], 'bwd', d['per_rank'][0]['bwd_kernel_us'], 'frac', d['roofline']['frac'], 'eager ms', d['eager_ms_per_step'])
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 /root/repo/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $OUT/bench_prof.json 2>/dev/null
python3 - $OUT <<'PY'
import csv, glob, sys, json
for f in glob.glob(sys.argv[1] + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pdist' in r['Name']: print('rocprof:', r['Name'][9:50], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
d = json.loads(open(sys.argv[1] + '/bench_prof.json').read().strip().splitlines()[-1])
print('bench under rocprof: fwd', d['per_rank'][0]['fwd_kernel_us'], 'bwd', d['per_rank'][0]['bwd_kernel_us'])
PY
