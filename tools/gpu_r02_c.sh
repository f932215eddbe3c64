#!/bin/bash
# round 2, GPU call C: where does the backward's time go — kernel trace (+ register counts) per variant, PMC of the main build, tile-height sweep
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-extra"
for V in main a1 a1w8; do
  if [ "$V" != main ]; then export MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_$V.so; else unset MM_MANIFOLDS_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02c_${V}_stats -o s -- $B --steps 20 --warmup 5 > $OUT/r02c_${V}.log 2>&1
  echo "== $V"; grep -E "spd_pdist_(bwd|fwd)" $OUT/r02c_${V}_stats/s_kernel_stats.csv | cut -d, -f2-8 | head -3
  python3 - $OUT/r02c_${V}_stats/s_kernel_trace.csv <<'PY'
import csv, sys
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:40]
    if 'spd_pdist' in k and k not in seen:
        seen.add(k)
        print(k, {c: r[c] for c in r if c in ('VGPR_Count', 'SGPR_Count', 'Accum_VGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Workgroup_Size', 'Grid_Size')})
PY
done
unset MM_MANIFOLDS_LIB
for TI in 8 16 32; do
  export MM_SPD_BWD_TI=$TI
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02c_ti${TI}_stats -o s -- $B --steps 20 --warmup 5 --no-prof > /dev/null 2>&1
  echo "== TI=$TI"; grep -E "spd_pdist_bwd" $OUT/r02c_ti${TI}_stats/s_kernel_stats.csv | cut -d, -f2-8 | head -2
done
unset MM_SPD_BWD_TI
P="$B --steps 3 --warmup 1 --no-prof"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/r02c_pmc_a -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/r02c_pmc_b -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum GRBM_COUNT --output-format csv -d $OUT/r02c_pmc_c -o p -- $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/r02c_pmc_a gpurun_out/r02c_pmc_b gpurun_out/r02c_pmc_c | cut -c1-700
