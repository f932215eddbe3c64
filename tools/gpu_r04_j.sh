#!/bin/bash
# round 4, session J: factor kinds as template arguments in both forms of the mixed-manifold pair kernel: product parity
# suites (default, each form forced, run-time kinds forced), config-4 / product step timings, linalg.fast tests.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_linalg_fast.py tests/test_minibatch_golden.py tests/test_configs_gpu.py tests/test_vec_gpu.py tests/test_fused_step_gpu.py tests/test_vec_forms_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
cd /tmp && export TMPDIR=/tmp
for ROUND in 1 2; do
  for RT in 0 1; do
    for CASE in "product 1025 f32" "product 1025 f64" "product 5000 f32"; do
      NAME=$(echo $CASE | tr ' ' '_')_rt${RT}_$ROUND
      MM_PRODUCT_RT_KINDS=$RT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME -o s -- python3 /root/repo/tools/profile_case.py $CASE 60 > /dev/null 2>&1
      python3 - $OUT/$NAME/s_kernel_stats.csv "run-time kinds=$RT round $ROUND: $CASE" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'product_' in r['Name'] and int(r['Calls']) > 10:
        out.append('%s avg %.2f min %.2f (x%s)' % (r['Name'].split('(')[0].replace('void mm::', '')[:56], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
    done
  done
done | tee $OUT/product_ab.txt
cd $GRAFT_REPO_ROOT
for C in c4_csphd_product_step_f32_native_graph c4_csphd_minibatch512_step_f32_graph c4_csphd_product_step_f64_fused_graph c4_product_n5000_step_f32_fused_graph; do python tools/bench_configs.py --only $C 2>/dev/null | tr -d '\n ' | cut -c1-700; echo; done | tee $OUT/c4_steps.txt
