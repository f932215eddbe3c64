import sys, os
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/matrix-manifolds_amd'); sys.path.insert(0, ROOT+'/tests')
import torch
import test_fused_step_gpu as t
from graphembed.native_step import NativeTrainStep
from graphembed.objectives import StressLoss
d=int(sys.argv[1]) if len(sys.argv)>1 else 2
emb,target=t._setup(d,131,torch.float32)
step=NativeTrainStep(emb, StressLoss(), target, t._opts(emb,'rsgd','rsgd'))
print('stepping', flush=True)
l=step(epoch=0,alpha=1.0)
torch.cuda.synchronize()
print('loss', l.item(), flush=True)
l=step(epoch=1,alpha=1.0)
torch.cuda.synchronize()
print('loss', l.item(), flush=True)
