#!/bin/bash
# round 2, GPU call B: hardware check of the new reduction, parity of the rewritten kernels, A/B of the backward variants
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
tools/micro/t_reduce > $OUT/r02b_reduce.log 2>&1; echo "t_reduce rc=$?"; tail -3 $OUT/r02b_reduce.log
timeout 900 python3 -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_c_abi.py -m gpu -x -q > $OUT/r02b_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r02b_pytest.log
for V in "" a1 a2 a2w7 a1w8; do
  if [ -n "$V" ]; then export MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_$V.so; else unset MM_MANIFOLDS_LIB; fi
  timeout 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > $OUT/r02b_bench_${V:-main}.json 2>/dev/null
  python3 - "$OUT/r02b_bench_${V:-main}.json" "${V:-main}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['per_rank'][0]
print(sys.argv[2], 'step %.1f us  fwd %.1f  bwd %.1f |' % (d['ms_per_step'] * 1e3, r['fwd_kernel_us'], r['bwd_kernel_us']),
      ' | '.join('%s: fwd %.1f bwd %.1f' % (e['workload'][:28], e.get('fwd_kernel_us') or 0, e.get('bwd_kernel_us') or 0) for e in d['extra'][:5]),
      '| cfg5 %.0f us' % (d['extra'][-1]['ms_per_step'] * 1e3))
PY
done
