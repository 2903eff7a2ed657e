import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
from scflow_amd import ops, _lib
DEV = 'cuda:0'
g = torch.Generator().manual_seed(1)
n, h, w, ch, cx = 2, 8, 8, 128, 256
hx = torch.randn((n, ch + cx, h, w), generator=g)
wzr = (torch.randn((2 * ch, ch + cx, 1, 5), generator=g) * 0.03).to(DEV)
bzr = (torch.randn((2 * ch,), generator=g) * 0.1).to(DEV)
wq = (torch.randn((ch, ch + cx, 1, 5), generator=g) * 0.03).to(DEV)
bq = (torch.randn((ch,), generator=g) * 0.1).to(DEV)
wzr2 = (torch.randn((2 * ch, ch + cx, 5, 1), generator=g) * 0.03).to(DEV)
wq2 = (torch.randn((ch, ch + cx, 5, 1), generator=g) * 0.03).to(DEV)
packs = [(ops.PackedConv.from_weight(wzr, bzr, padding=(0, 2)), ops.PackedConv.from_weight(wq, bq, padding=(0, 2))),
         (ops.PackedConv.from_weight(wzr2, bzr, padding=(2, 0)), ops.PackedConv.from_weight(wq2, bq, padding=(2, 0)))]
a, b = hx.to(DEV), hx.to(DEV)
za, ra = torch.empty((n, ch, h, w), device=DEV), torch.empty((n, ch, h, w), device=DEV)
zb, rb = torch.empty_like(za), torch.empty_like(za)
za.fill_(float('nan')); ra.fill_(float('nan')); zb.fill_(float('nan')); rb.fill_(float('nan'))
ops.sepconv_gru(packs, a, ch, za, ra)
ops.conv_timing(True)
ops.sepconv_gru(packs, b, ch, zb, rb)
ev = ops.conv_timing(False)
print('nan in h (C entry):', int(torch.isnan(a[:, :ch]).sum()), ' (launch by launch):', int(torch.isnan(b[:, :ch]).sum()),
      'nan z', int(torch.isnan(za).sum()), int(torch.isnan(zb).sum()), 'nan rh', int(torch.isnan(ra).sum()), int(torch.isnan(rb).sum()))
c2 = hx.to(DEV)
ops.sepconv_gru(packs, c2, ch, za, ra)      # reuse dirty scratch
print('reuse scratch: diff vs first', float((c2[:, :ch] - a[:, :ch]).abs().max()))
print('z diff', float((za - zb).abs().max()), 'rh diff', float((ra - rb).abs().max()), 'h diff', float((a[:, :ch] - b[:, :ch]).abs().max()))
print('plans', [(p.plans, q.plans) for p, q in packs])
lib = _lib.load()
for pzr, pq in packs:
    for pc, c0, c1 in ((pzr, 384, 0), (pq, 128, 256)):
        for kc, wp in ((8, pc.wp), (32, pc.wp_alt)):
            d = _lib.ConvDesc()
            d.in0, d.C0, d.C1, d.in0_nstride, d.in1_nstride = a.data_ptr(), c0, c1, 384 * 64, 384 * 64
            d.in1 = a.data_ptr() if c1 else None
            d.N, d.H, d.W = n, h, w
            d.wp, d.Mld, d.Cout, d.KC = wp.data_ptr(), pc.mld, pc.cout, kc
            d.KH, d.KW, d.stride, d.pad_h, d.pad_w = pc.kh, pc.kw, 1, pc.pad_h, pc.pad_w
            d.out, d.out_nstride, d.out_div = za.data_ptr(), ch * 64, 1.0
            d.wp_a4, d.a4_groups, d.a4_mld = pc.wp4.data_ptr(), pc.g4, pc.mld
            info = (C.c_int32 * 4)()
            rc = lib.scf_conv2d_query(C.byref(d), info)
            print(pc.kh, pc.kw, 'c0', c0, 'KC', kc, 'rc', rc, list(info))
