import sys, os
sys.path.insert(0, '/root/repo')
import torch
from scflow_amd import ops
torch.manual_seed(0)
for (n,h,w) in [(32,32,32),(3,32,32),(2,60,80),(5,24,40)]:
    f1 = torch.randn(n,64,h,w,device='cuda'); f2 = torch.randn(n,64,h,w,device='cuda')
    for mask in (0, ops.pyramid_layout(h,w,4,4)):
        pyr = ops.corr_build(f1,f2,4,tiled_levels=mask) if mask else ops.corr_build(f1,f2,4)
        flow = torch.randn(n,2,h,w,device='cuda')*4
        flow[0,:,0,0] = 1e6
        outs=[]
        for m in (1,2,3,4,5):
            ops.tune('lookup_pipe', m)
            outs.append(ops.corr_lookup(pyr, flow, 4, tiled_levels=mask) if mask else ops.corr_lookup(pyr, flow, 4))
        ops.tune('lookup_pipe', 0)
        torch.cuda.synchronize()
        print((n,h,w), 'mask', mask, 'v9(2)==v8', torch.equal(outs[0],outs[1]), 'v9(3)==v8', torch.equal(outs[0],outs[2]), "gpb2==v8", torch.equal(outs[0],outs[3]), "gpb4==v8", torch.equal(outs[0],outs[4]))
