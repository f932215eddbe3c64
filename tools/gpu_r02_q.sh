#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_vec_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $OUT/r02q_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r02q_pytest.log
timeout 600 python3 -m pytest tests/test_round2_gpu.py -m gpu -x -q -k "fuzz_pdist or fuzz_product or fuzz_graph" > $OUT/r02q_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/r02q_fuzz.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lor.py <<'PY'
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/matrix-manifolds_amd')
import torch
from graphembed import manifolds as M
torch.manual_seed(0)
for name, man, m in (('lorentz11', M.Lorentz(11), 11), ('sphere6', M.Sphere(6), 6), ('euclidean10', M.Euclidean(10), 10)):
    n = 4039
    x = man.rand(n, out=torch.empty(0, device='cuda')).requires_grad_()
    g = torch.randn(n * (n - 1) // 2, device='cuda')
    for _ in range(12):
        d2 = man.pdist(x, squared=True)
        gr, = torch.autograd.grad(d2, x, g)
    torch.cuda.synchronize()
PY
for ORD in 1 0; do
  MM_GRAM_BWD_ORDERED=$ORD rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02q_ord${ORD}_stats -o s -- python3 /tmp/lor.py > /dev/null 2>&1
  echo "== MM_GRAM_BWD_ORDERED=$ORD"; python3 - $OUT/r02q_ord${ORD}_stats/s_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'vec_gram' in r['Name']: print('  ', r['Name'][9:60], r['Calls'], 'avg %.1f us' % (float(r['AverageNs']) / 1e3))
PY
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r02q_pmc_fetch -o p -- python3 /tmp/lor.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/r02q_pmc_write -o p -- python3 /tmp/lor.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/summarize_pmc.py gpurun_out/r02q_pmc_fetch gpurun_out/r02q_pmc_write | grep "vec_gram_bwd" | cut -c1-400
