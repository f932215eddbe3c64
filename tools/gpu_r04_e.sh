#!/bin/bash
# round 4, session E: product pair kernel after the prologue / loss-reduction changes (+ its timeline), small-launch timelines
# of the SPD backward (a rank's shard of 8, n = 1000, SPD(4) n = 2274) and their grid sweeps, alignment microbenchmark v2.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_minibatch_golden.py tests/test_configs_gpu.py tests/test_vec_gpu.py tests/test_fused_step_gpu.py tests/test_round2_gpu.py -m gpu -x -q > $OUT/pytest_a.log 2>&1
echo "pytest(a) rc=$?"; tail -2 $OUT/pytest_a.log
V=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants
MM_MANIFOLDS_LIB=$V/libmm_pstamp.so python3 tools/product_timeline.py 1025 2>&1 | grep -v amdgpu.ids | tee $OUT/product_timeline.txt
for CASE in "5000 --warm --shard=0/8" "5000 --warm --shard=7/8" "1000 --warm" "2274 --warm --d=4" "5000 --warm"; do
  echo "== stamp_timeline $CASE"; MM_MANIFOLDS_LIB=$V/libmm_stamp.so python3 tools/stamp_timeline.py $CASE 2>&1 | grep -v amdgpu.ids
done | tee $OUT/small_launch_timelines.txt
cd /tmp && export TMPDIR=/tmp
C="python3 /root/repo/tools/profile_case.py"
stat() {  # dir label pattern
  python3 - $1/s_kernel_stats.csv "$2" "$3" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r['Name'] and int(r['Calls']) > 10:
        out.append('%s avg %.1f min %.1f (x%s)' % (r['Name'].split('(')[0].replace('void mm::', '')[:52], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))
print(sys.argv[2], '|', '; '.join(out))
PY
}
for R in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod_$R -o s -- $C product 1025 f32 60 > /dev/null 2>&1
  stat $OUT/prod_$R "product n=1025 f32 round $R" product_
done | tee $OUT/product.txt
for G in 256 384 512 768 1024; do
  MM_SPD_BWD_GRID=$G rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sg8_$G -o s -- python3 /root/repo/tools/shard_case.py 8 0 > /dev/null 2>&1
  stat $OUT/sg8_$G "SPD(3) n=5000 rank 0 of 8, grid $G" pdist_bwd
done | tee $OUT/grid_sweeps.txt
for G in 0 512 768 1024 1280 2048; do
  ( [ $G -gt 0 ] && export MM_SPD_BWD_GRID=$G; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s4_$G -o s -- $C pdist 4 2274 f32 0.1 60 > /dev/null 2>&1 )
  stat $OUT/s4_$G "SPD(4) n=2274, grid $G (0 = default)" pdist_bwd
done | tee -a $OUT/grid_sweeps.txt
for G in 0 256 512 768 1024; do
  ( [ $G -gt 0 ] && export MM_SPD_BWD_GRID=$G; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s3_$G -o s -- $C pdist 3 1000 f32 0.1 60 > /dev/null 2>&1 )
  stat $OUT/s3_$G "SPD(3) n=1000, grid $G (0 = default)" pdist_
done | tee -a $OUT/grid_sweeps.txt
cd $GRAFT_REPO_ROOT
python tools/bench_configs.py --only lorentz24_minibatch512 2>/dev/null | grep step_us
cd tools/micro && hipcc -O3 --offload-arch=gfx950 -Wno-unused-value pair_align.hip -o /tmp/pair_align && /tmp/pair_align | tee $OUT/pair_align.txt
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pa_fetch -o p -- /tmp/pair_align > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pa_write -o p -- /tmp/pair_align > /dev/null 2>&1
python3 - $OUT <<'PY' | tee -a $OUT/pair_align.txt
import csv, glob, sys, collections
for kind, unit in (('fetch', 2.0), ('write', 1.0)):
    acc = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + f'/pa_{kind}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        print(f'{kind.upper()}_SIZE {k:40s} {sum(v) / len(v) * 1024 * unit / 1e6:8.1f} MB per launch' + (' (x2: gfx950 counts half)' if kind == 'fetch' else ''))
PY
