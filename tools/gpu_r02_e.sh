#!/bin/bash
# round 2, GPU call E: balanced persistent backward — parity, timing, timeline, grid-size sweep
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_c_abi.py -m gpu -x -q > $OUT/r02e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r02e_pytest.log
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['per_rank'][0]
print(sys.argv[2], 'step %.1f us  fwd %.1f  bwd %.1f |' % (d['ms_per_step'] * 1e3, r['fwd_kernel_us'], r['bwd_kernel_us']),
      ' | '.join('%s: fwd %.1f bwd %.1f' % (e['workload'][:28], e.get('fwd_kernel_us') or 0, e.get('bwd_kernel_us') or 0) for e in d.get('extra', [])[:5]),
      ('| cfg5 %.0f us' % (d['extra'][-1]['ms_per_step'] * 1e3)) if d.get('extra') else '')
PY
}
timeout 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > $OUT/r02e_bench_main.json 2>/dev/null; summ $OUT/r02e_bench_main.json main
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_a1.so timeout 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > $OUT/r02e_bench_a1.json 2>/dev/null; summ $OUT/r02e_bench_a1.json a1
for G in 1024 1280 1536 1792 2048 3072 3584; do
  MM_SPD_BWD_GRID=$G timeout 300 python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 10 > $OUT/r02e_bench_g$G.json 2>/dev/null; summ $OUT/r02e_bench_g$G.json grid$G
done
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_stamp.so python3 tools/stamp_timeline.py 5000 2>/dev/null | tee $OUT/r02e_timeline.txt
