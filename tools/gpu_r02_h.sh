#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['per_rank'][0]
print(sys.argv[2], 'step %.1f us  fwd %.1f  bwd %.1f' % (d['ms_per_step'] * 1e3, r['fwd_kernel_us'], r['bwd_kernel_us']))
PY
}
for V in t64 t32; do
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_$V.so timeout 300 python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 10 > $OUT/r02h_bench_$V.json 2>/dev/null; summ $OUT/r02h_bench_$V.json $V
done
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_t64stamp.so python3 tools/stamp_timeline.py 5000 2>/dev/null | tee $OUT/r02h_timeline_t64.txt
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_stamp.so python3 tools/stamp_timeline.py 5000 2>/dev/null | tee $OUT/r02h_timeline_main.txt
