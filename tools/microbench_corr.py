import torch, time, sys
sys.path.insert(0,'/root/repo')
from scflow_amd import ops
dev='cuda:0'
for N in (1, 32):
    h=w=32
    f1=torch.randn(N,256,h,w,device=dev); f2=torch.randn(N,256,h,w,device=dev)
    pyr=ops.corr_build(f1,f2,4)
    flow=torch.randn(N,2,h,w,device=dev)*3
    out=ops.corr_lookup(pyr,flow,4)
    torch.cuda.synchronize()
    for name,fn in (('lookup',lambda: ops.corr_lookup(pyr,flow,4,out=out)),('build',lambda: ops.corr_build(f1,f2,4,out=pyr))):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        for _ in range(5): fn()
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e)/50
        if name=='lookup': print(f'N={N} lookup {ms*1e3:.1f} us  {2904*N*h*w/ms/1e6:.1f} GB/s algorithmic')
        else: print(f'N={N} build {ms*1e3:.1f} us  {2*256*(h*w)**2*N/ms/1e9:.2f} TFLOP/s')
