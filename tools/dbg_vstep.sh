cd /root/repo
timeout 2400 python3 -m pytest tests/test_fused_step_gpu.py tests/test_comm_gpu.py tests/test_vec_gpu.py tests/test_configs_gpu.py tests/test_radam.py -m gpu -x -q > gpurun_out/vstep_tests.log 2>&1; grep -v "^$" gpurun_out/vstep_tests.log | tail -8
cd /tmp && export TMPDIR=/tmp
for F in fused unfused; do
for C in "vstep 11 4039 f32 lorentz"; do
NAME=$(echo $C | tr ' .' '__')_$F
if [ $F = unfused ]; then export MM_VEC_STEP_UNFUSED=1; else unset MM_VEC_STEP_UNFUSED; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/vsym_$NAME -o s -- python3 /root/repo/tools/profile_case.py $C 200 > /dev/null 2>&1
echo "== $C $F"; python3 - $(find /root/repo/gpurun_out/vsym_$NAME -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print('%-60s calls %5s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done; done
