#!/bin/bash
# Counter evidence for the mixed-manifold pair kernel (csphd product step, n = 1025): VALU / LDS / HBM bytes.
# (counters in their own passes, no trace domains alongside --pmc)
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/tools/bench_configs.py --only c4_csphd_product_step_f32_fused"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/prod_pmc -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prod_pmc_w -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prod_pmc_f -o p -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/prod_pmc gpurun_out/prod_pmc_w gpurun_out/prod_pmc_f
