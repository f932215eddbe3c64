#!/bin/bash
# round 4, session B: node minibatches inside the single factors' pair kernels (tests, launch counts, step times), and the
# fp64 / fp32 pair kernels of the main build against the round-3 kernels (`mat`).
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_minibatch_golden.py tests/test_spd_gpu.py tests/test_fused_step_gpu.py tests/test_c_abi.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -3 $OUT/pytest_a.log
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
export MM_AB_CASES="pdist 3 5000 f64 0.1;pdist 3 5000 f64 0.35;pdist 3 5000 f32 0.1;pdist 3 5000 f32 0.35;pdist 3 5000 f32 0.6"
export MM_AB_ROUNDS=2
bash tools/gpu_ab_libs.sh mat main 2>&1 | tee $OUT/ab.txt
python tools/bench_configs.py --only minibatch512 > $OUT/minibatch.json 2> $OUT/minibatch.err
cat $OUT/minibatch.json
cd /tmp && export TMPDIR=/tmp
for C in c5_spd4_minibatch512_step_n16384_f32_native_graph lorentz24_minibatch512_step_n4039_f32_native_graph; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$C -o s -- python3 /root/repo/tools/bench_configs.py --only $C > /dev/null 2>&1
  python3 - $OUT/trace_$C/s_kernel_stats.csv $C <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -int(r['Calls']))
print(sys.argv[2])
for r in rows[:8]:
    print('   %6s calls  avg %8.1f us  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
done
