#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/r02l_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/r02l_pytest.log
python3 - <<'PY'
import json, subprocess, sys
r = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--steps', '30', '--warmup', '10'], capture_output=True, text=True)
open('gpurun_out/r02l_bench.json', 'w').write(r.stdout)
d = json.loads(r.stdout)
print('step %.1f us  %.1f G pairs/s' % (d['ms_per_step'] * 1e3, d['value'] / 1e9), d['per_rank'][0])
for e in d['extra']:
    print('  %-70s step %.1f us fwd %s bwd %s' % (e['workload'][:70], e['ms_per_step'] * 1e3, e.get('fwd_kernel_us'), e.get('bwd_kernel_us')))
PY
