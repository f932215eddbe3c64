#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_spd_gpu.py tests/test_configs_gpu.py tests/test_c_abi.py -m gpu -x -q > $OUT/r02k_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/r02k_pytest.log
summ() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d['per_rank'][0]
print(sys.argv[2], 'step %.1f us  fwd %.1f  bwd %.1f |' % (d['ms_per_step'] * 1e3, r['fwd_kernel_us'], r['bwd_kernel_us']),
      ' | '.join('%s: fwd %.1f bwd %.1f' % (e['workload'][:28], e.get('fwd_kernel_us') or 0, e.get('bwd_kernel_us') or 0) for e in d.get('extra', [])[:5]),
      ('| cfg5 %.0f us' % (d['extra'][-1]['ms_per_step'] * 1e3)) if d.get('extra') else '')
PY
}
timeout 300 python3 bench.py --no-cpu-baseline --steps 30 --warmup 10 > $OUT/r02k_bench_main.json 2>/dev/null; summ $OUT/r02k_bench_main.json main
for G in 768 1024 1280; do
  MM_SPD_BWD_GRID=$G timeout 300 python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 10 > $OUT/r02k_bench_g$G.json 2>/dev/null; summ $OUT/r02k_bench_g$G.json grid$G
done
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_stamp.so python3 tools/stamp_timeline.py 5000 2>/dev/null | tee $OUT/r02k_timeline.txt
cd /tmp && export TMPDIR=/tmp
P="python3 /root/repo/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-prof"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/r02k_pmc_a -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/r02k_pmc_b -o p -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02k_stats -o s -- python3 /root/repo/bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/r02k_pmc_a gpurun_out/r02k_pmc_b | grep -E "bwd|fwd" | cut -c1-600
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r02k_stats/s_kernel_stats.csv')):
    if 'spd_' in r['Name']: print(r['Name'][9:45], r['Calls'], 'avg %.1f us min %.1f max %.1f' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
