#!/bin/bash
# micro-fixes (saddr loads/stores, raw max, bitwise gate) + 4x4 invariants-only forward: parity, then the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
timeout 1500 python -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_round2_gpu.py -m gpu -x -q > gpurun_out/r02r/pytest.log 2>&1
tail -5 gpurun_out/r02r/pytest.log
timeout 600 python bench.py > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02r/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k,v in d.get('extra',{}).items(): print(k, v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in ('ms_per_step','value','fwd_us','bwd_us','kernels_us')})
PY
