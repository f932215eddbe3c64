#!/bin/bash
# round 4, session D: the product pair kernel's workgroup timeline, alignment microbenchmark with counters, shard balance
# (rounds), the vector subset kernel's rows-per-workgroup sweep, layer-F1 known answers.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_layer_f1_known_answers.py tests/test_metrics.py tests/test_minibatch_golden.py tests/test_configs_gpu.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -2 $OUT/pytest_a.log
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_pstamp.so python3 tools/product_timeline.py 1025 2>&1 | tee $OUT/product_timeline.txt
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
for R in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod_$R -o s -- $C product 1025 f32 60 > /dev/null 2>&1
  python3 - $OUT/prod_$R/s_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'product_' in r['Name'] and int(r['Calls']) > 10:
        print('product n=1025 f32 |', r['Name'].split('(')[0][:60], 'avg %.1f min %.1f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
# alignment microbenchmark: times, then FETCH_SIZE / WRITE_SIZE per kernel
cd $GRAFT_REPO_ROOT/tools/micro && hipcc -O3 --offload-arch=gfx950 -Wno-unused-value pair_align.hip -o /tmp/pair_align && /tmp/pair_align | tee $OUT/pair_align.txt
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pa_fetch -o p -- /tmp/pair_align > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pa_write -o p -- /tmp/pair_align > /dev/null 2>&1
python3 - $OUT <<'PY' | tee -a $OUT/pair_align.txt
import csv, glob, sys, collections
for kind, unit in (('fetch', 2.0), ('write', 1.0)):
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + f'/pa_{kind}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{kind.upper()}_SIZE {k:40s} {sum(v) / len(v) * 1024 * unit / 1e6:8.1f} MB per launch' + (' (x2: gfx950 counts half)' if kind == 'fetch' else ''))
PY
cd $GRAFT_REPO_ROOT
for K in 0 100000; do MM_SHARD_K=$K MM_SPD4_BWD_TWO_COLS=1 python3 tools/shard_balance.py 8 16384 3; done 2>/dev/null | tee $OUT/shard_balance.txt
MM_SHARD_K=0 python3 tools/shard_balance.py 8 16384 3 2>/dev/null | tee -a $OUT/shard_balance.txt
for ROWS in 8 16 32 64; do
  echo "MM_VEC_SUBSET_ROWS=$ROWS"; MM_VEC_SUBSET_ROWS=$ROWS python tools/bench_configs.py --only lorentz24_minibatch512 2>/dev/null | grep step_us
done | tee $OUT/vec_subset_rows.txt
cd /tmp
MM_VEC_SUBSET_ROWS=16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_l24 -o s -- python3 /root/repo/tools/bench_configs.py --only lorentz24_minibatch512 > /dev/null 2>&1
python3 - $OUT/trace_l24/s_kernel_stats.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -int(r['Calls']))
for r in rows[:6]:
    print('   %6s calls  avg %8.1f us  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:100]))
PY
