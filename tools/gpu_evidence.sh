#!/bin/bash
# Round evidence set (run on the GPU box through gpurun; the summaries are then copied to profiles/<TAG>_*):
#   tools/gpu_evidence.sh TAG
#   1. the whole `-m gpu` suite + smoke
#   2. rocprofv3 kernel stats of `python3 bench.py` (headline) and PMC passes of the same command (SQ / FETCH_SIZE /
#      WRITE_SIZE in SEPARATE passes, no trace domains alongside --pmc) -> profiles/pmc_head.json stamped with the kernel-source hash
#   3. per-case kernel stats + PMC: fp64, mid-training spread, SPD(4) sizes of BASELINE config 5 (init and mid-training),
#      SPD(6) / SPD(9), Lorentz(11) with the MFMA counters, the mixed-manifold pair kernel at n = 1025 and n = 5000
#   4. the warm-regime shader clock from in-kernel stamps (lib/variants/libmm_stamp.so, if built)
#   5. a default `python3 bench.py` run, tools/bench_configs.py, tools/shard_kernel_times.py, the eager-path host profile
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o s -- $B --steps 20 --warmup 5 > $OUT/bench_stats.log 2>&1
P="$B --steps 3 --warmup 1 --no-prof"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/bench_pmc_sq -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_pmc_fetch -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/bench_pmc_write -o p -- $P > /dev/null 2>&1
C="python3 /root/repo/tools/profile_case.py"
for CASE in "pdist 3 5000 f64 0.1" "pdist 3 5000 f32 0.35" "pdist 3 5000 f64 0.35" "pdist 4 2274 f32 0.1" "pdist 4 16384 f32 0.1" "pdist 4 16384 f32 0.35" "loss 4 16384 f32" "vec 11 4039 f32 lorentz" "pdist 6 2000 f32 0.1" "pdist 6 2000 f32 0.35" "pdist 6 2000 f64 0.1" "pdist 9 2000 f32 0.1" "step 3 5000 f32" "vstep 11 4039 f32 lorentz"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/case_${NAME}_stats -o s -- $C $CASE 40 > /dev/null 2>&1
done
for CASE in "pdist 3 5000 f32 0.35" "pdist 3 5000 f64 0.1" "pdist 3 5000 f64 0.35" "pdist 4 16384 f32 0.1" "loss 4 16384 f32" "vec 11 4039 f32 lorentz" "pdist 6 2000 f32 0.1"; do
  NAME=$(echo $CASE | tr ' .' '__')
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/case_${NAME}_pmc_sq -o p -- $C $CASE 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/case_${NAME}_pmc_fetch -o p -- $C $CASE 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/case_${NAME}_pmc_write -o p -- $C $CASE 3 > /dev/null 2>&1
done
# matrix-core utilisation of the Gram kernels (north_star: "MFMA utilisation against gfx950 peak"): the forward is the default
# path; the matrix-core backward is forced (the default backward up to m = 16 is the symmetric VALU kernel)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/case_vec_11_4039_f32_lorentz_pmc_mfma -o p -- $C vec 11 4039 f32 lorentz 3 > /dev/null 2>&1
export MM_VEC_BWD=gram
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/case_vecgram_11_4039_f32_lorentz_stats -o s -- $C vec 11 4039 f32 lorentz 40 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/case_vecgram_11_4039_f32_lorentz_pmc_mfma -o p -- $C vec 11 4039 f32 lorentz 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/case_vecgram_11_4039_f32_lorentz_pmc_fetch -o p -- $C vec 11 4039 f32 lorentz 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/case_vecgram_11_4039_f32_lorentz_pmc_write -o p -- $C vec 11 4039 f32 lorentz 3 > /dev/null 2>&1
unset MM_VEC_BWD
# the mixed-manifold pair kernel (config 4) at csphd size and at n = 5000
for N in 1025 5000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/case_product_${N}_stats -o s -- $C product $N f32 40 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/case_product_${N}_pmc_sq -o p -- $C product $N f32 3 > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/case_product_${N}_pmc_fetch -o p -- $C product $N f32 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/case_product_${N}_pmc_write -o p -- $C product $N f32 3 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_stamp.py gpurun_out/$TAG/bench_pmc_sq gpurun_out/$TAG/bench_pmc_fetch gpurun_out/$TAG/bench_pmc_write --source "profiles/${TAG}_bench_pmc_summary.txt (rocprofv3 --pmc, separate passes; python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --no-prof)" > /dev/null
cp profiles/pmc_head.json $OUT/pmc_head.json
python3 tools/summarize_round.py gpurun_out/$TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
if [ -f matrix-manifolds_amd/lib/variants/libmm_stamp.so ]; then
  MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_stamp.so python3 tools/stamp_timeline.py 5000 > $OUT/timeline_cold.txt 2>&1
  MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_stamp.so python3 tools/stamp_timeline.py 5000 --warm > $OUT/timeline_warm.txt 2>&1
  head -3 $OUT/timeline_cold.txt $OUT/timeline_warm.txt
fi
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
timeout 1200 python3 tools/bench_configs.py > $OUT/configs.json 2> $OUT/configs.err; echo "configs rc=$?"
# node minibatches through the one-call step: the launches of a step (kernel trace of two of the cases)
cd /tmp
for C5 in c5_spd4_minibatch512_step_n16384_f32_native_graph lorentz24_minibatch512_step_n4039_f32_native_graph; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/case_${C5}_stats -o s -- python3 /root/repo/tools/bench_configs.py --only $C5 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
if [ -f matrix-manifolds_amd/lib/variants/libmm_pstamp.so ]; then
  MM_MANIFOLDS_LIB=/root/repo/matrix-manifolds_amd/lib/variants/libmm_pstamp.so python3 tools/product_timeline.py 1025 > $OUT/product_timeline.txt 2>&1
fi
python3 tools/shard_kernel_times.py > $OUT/shard_kernel_times.json 2> $OUT/shard.err
python3 tools/eager_profile.py euclid > $OUT/eager_euclid.txt 2>&1
python3 tools/eager_profile.py spd > $OUT/eager_spd.txt 2>&1
MM_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_gloo2_dryrun.json 2> $OUT/bench_gloo2.err; echo "gloo2 rc=$?"
