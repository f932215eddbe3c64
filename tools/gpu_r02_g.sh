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
for V in p0 p2; do
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_$V.so timeout 300 python3 bench.py --no-cpu-baseline --no-extra --steps 30 --warmup 10 > $OUT/r02g_bench_$V.json 2>/dev/null; summ $OUT/r02g_bench_$V.json $V
done
MM_MANIFOLDS_LIB=$GRAFT_REPO_ROOT/matrix-manifolds_amd/lib/variants/libmm_p0stamp.so python3 tools/stamp_timeline.py 5000 2>/dev/null | tee $OUT/r02g_timeline_p0.txt
cd /tmp && export TMPDIR=/tmp
P="python3 /root/repo/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-prof"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/r02g_pmc_a -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/r02g_pmc_b -o p -- $P > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/r02g_pmc_a gpurun_out/r02g_pmc_b | grep bwd | cut -c1-700
