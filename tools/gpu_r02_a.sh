#!/bin/bash
# round 2, GPU call A: baseline of HEAD — full GPU test suite, bench (N=1), FETCH_SIZE calibration, kernel stats + PMC.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/r02a_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r02a_pytest.log
tail -5 $OUT/r02a_pytest.log
timeout 600 python3 bench.py > $OUT/r02a_bench.json 2> $OUT/r02a_bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/r02a_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r02a_cal_fetch -o p -- $GRAFT_REPO_ROOT/tools/micro/load_bw > $OUT/r02a_cal.log 2>&1
cat $OUT/r02a_cal.log | grep -E "linear|pairs"
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r02a_cal_fetch/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print('CAL', k, 'FETCH_SIZE mean', sum(v) / len(v), 'n', len(v))
PY
bash tools/gpu_profile.sh r02a > $OUT/r02a_profile.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/r02a_pmc_sq gpurun_out/r02a_pmc_fetch gpurun_out/r02a_pmc_write | cut -c1-600
find gpurun_out/r02a_stats -name "*kernel_stats.csv" -exec head -8 {} \;
