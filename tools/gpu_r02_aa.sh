#!/bin/bash
# backward grid sized by rows per workgroup: parity, one rank's kernel times for N-way shards, the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aa
timeout 1200 python -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_round2_gpu.py -m gpu -x -q > gpurun_out/r02aa/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r02aa/pytest.log
python3 tools/shard_kernel_times.py 2>/dev/null > gpurun_out/r02aa/shard_kernel_times.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r02aa/shard_kernel_times.json'))
for k, v in d.items():
    print(k)
    for kk, vv in v.items(): print('  ', kk, vv)
PY
bash tools/gpu_shard_stats.sh
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02aa/bench.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('gpurun_out/r02aa/bench.json').read().strip().splitlines()[-1])
print('bench: step', d['ms_per_step']*1e3, 'fwd', d['per_rank'][0]['fwd_kernel_us'], 'bwd', d['per_rank'][0]['bwd_kernel_us'])
for e in d['extra']: print(e['workload'][:60], round(e['ms_per_step']*1e3,1), e.get('fwd_kernel_us'), e.get('bwd_kernel_us'))
"
