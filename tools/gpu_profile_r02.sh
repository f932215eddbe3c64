#!/bin/bash
# Round-2 evidence set (run on the GPU box via gpurun; copies of the summaries go to profiles/):
#   kernel-trace stats of `python3 bench.py` (headline), PMC passes of the same command (SQ / FETCH_SIZE / WRITE_SIZE in
#   separate passes), per-case kernel stats + PMC for fp64 and the SPD(4) sizes of BASELINE config 5, the Lorentz(11)
#   case, the workgroup timeline of the backward, and profiles/pmc_head.json stamped with the kernel-source hash.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o s -- $B --steps 20 --warmup 5 > $OUT/bench_stats.log 2>&1
P="$B --steps 3 --warmup 1 --no-prof"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/bench_pmc_sq -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_pmc_fetch -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/bench_pmc_write -o p -- $P > /dev/null 2>&1
C="python3 /root/repo/tools/profile_case.py"
for CASE in "pdist 3 5000 f64 0.1" "pdist 3 5000 f32 0.35" "pdist 4 2274 f32 0.1" "pdist 4 16384 f32 0.1" "loss 4 16384 f32" "vec 11 4039 f32 lorentz"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/case_${NAME}_stats -o s -- $C $CASE 40 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/case_${NAME}_pmc_sq -o p -- $C $CASE 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/case_${NAME}_pmc_fetch -o p -- $C $CASE 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/case_${NAME}_pmc_write -o p -- $C $CASE 3 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_stamp.py gpurun_out/$TAG/bench_pmc_sq gpurun_out/$TAG/bench_pmc_fetch gpurun_out/$TAG/bench_pmc_write --source "profiles/${TAG}_bench_pmc_summary.txt (rocprofv3 --pmc, separate passes; python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-prof)" > /dev/null
cp profiles/pmc_head.json gpurun_out/$TAG/pmc_head.json
python3 tools/summarize_r02.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt 2>&1
cat gpurun_out/$TAG/summary.txt
timeout 600 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 600 gpurun_out/$TAG/bench.json
