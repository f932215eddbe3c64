import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/matrix-manifolds_amd')
import torch, bench
world, rank = int(sys.argv[1]), int(sys.argv[2])
wl = bench.PdistWorkload(3, 5000, torch.float32, 0.1, world, rank, torch.device('cuda', 0))
for _ in range(400):
    wl.kernels()
torch.cuda.synchronize()
