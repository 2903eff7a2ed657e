import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from scflow_amd import ops
DEV='cuda:0'
n=32
x=torch.randn((n,324,32,32),device=DEV); w=torch.randn((256,324,1,1),device=DEV)*0.05; b=torch.zeros(256,device=DEV)
pc=ops.PackedConv.from_weight(w,b,padding=0)
out=torch.empty((n,256,32,32),device=DEV)
for force in (0,1,0,1):
    ops.tune('dma_force_ksplit', force)
    for _ in range(50): ops.conv2d(pc,x,out=out,act=ops.ACT_RELU)
    with ops.record_conv_kernels() as ran: ops.conv2d(pc,x,out=out,act=ops.ACT_RELU)
    ts=sorted(ops.time_first_kernel(lambda: ops.conv2d(pc,x,out=out,act=ops.ACT_RELU)) for _ in range(9))
    print('324->256 1x1 N32 force_ksplit', force, ran, f'{ts[4]:.1f} us')
ops.tune('dma_force_ksplit', 0)
