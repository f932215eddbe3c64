#!/bin/bash
# round 4, session I: the mixed-manifold pair kernel with the index indirection as a template argument (one batch of prologue
# loads instead of six serial round trips): parity suites, workgroup timeline, rocprofv3 A/B against the build with the
# factor kinds fixed at compile time (libmm_pkfix.so: what do run-time kinds cost the row loop?); linalg.fast tests.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_linalg_fast.py tests/test_minibatch_golden.py tests/test_configs_gpu.py tests/test_vec_gpu.py tests/test_fused_step_gpu.py tests/test_vec_forms_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
V=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants
MM_MANIFOLDS_LIB=$V/libmm_pstamp.so python3 tools/product_timeline.py 1025 2>&1 | grep -v amdgpu.ids | tee $OUT/product_timeline.txt
export MM_AB_ROUNDS=3
export MM_AB_CASES="product 1025 f32"
cd /tmp && export TMPDIR=/tmp
for ROUND in 1 2 3; do
  for LIB in main pkfix; do
    if [ "$LIB" = main ]; then unset MM_MANIFOLDS_LIB; else export MM_MANIFOLDS_LIB=$V/libmm_$LIB.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod_${LIB}_$ROUND -o s -- python3 /root/repo/tools/profile_case.py product 1025 f32 60 > /dev/null 2>&1
    python3 - $OUT/prod_${LIB}_$ROUND/s_kernel_stats.csv "$LIB round $ROUND: product n=1025 f32" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'product_' in r['Name'] and int(r['Calls']) > 10:
        out.append('%s avg %.2f min %.2f (x%s)' % (r['Name'].split('(')[0].replace('void mm::', '')[:52], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
  done
done | tee $OUT/product_ab.txt
unset MM_MANIFOLDS_LIB
cd $GRAFT_REPO_ROOT
for C in c4_csphd_product_step_f32_native_graph c4_csphd_minibatch512_step_f32_graph; do python tools/bench_configs.py --only $C 2>/dev/null | tr -d '\n ' | cut -c1-700; echo; done | tee $OUT/c4_steps.txt
