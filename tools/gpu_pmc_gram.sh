#!/bin/bash
# MFMA evidence for the Gram forward of the Lorentz / facebook-size config (n=4039, m=11)
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/tools/bench_configs.py --only c2_facebook_lorentz11_f32"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/gram_pmc -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/gram_pmc_w -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/gram_pmc_f -o p -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gram_stats -o s -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/gram_pmc gpurun_out/gram_pmc_w gpurun_out/gram_pmc_f; head -8 gpurun_out/gram_stats/s_kernel_stats.csv | cut -c1-200
