"""Timeline of the Winograd kernel's chunks (library built with -DSCF_WINO_LAB): per-phase shader-clock
durations in steady state, and how the two blocks that share a CU interleave."""
import os, sys, ctypes as C, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libscflow_hip_exp.so')
from scflow_amd import ops
DEV = 'cuda:0'
lib = _lib.load()
lib.scf_wino_trace_set.argtypes = [C.c_void_p, C.c_int]
cases = [('128->512 @32 N32', 32, 128, 512, 32, 32), ('256->192 @32 N32', 32, 256, 192, 32, 32)]
NB = 2048
ops.set_conv_winograd(True)
os.environ['SCF_WINO_LAB'] = '32'
for name, n, cin, cout, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, 3, 3), device=DEV) * (1.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), device=DEV) * 0.1
    pc = ops.PackedConv.from_weight(w, b, padding=1)
    out = torch.empty((n, cout, H, W), device=DEV)
    for _ in range(3):
        ops.conv2d(pc, x, out=out, act=ops.ACT_RELU)
    tr = torch.zeros((NB, 4, 128), dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    lib.scf_wino_trace_set(C.c_void_p(tr.data_ptr()), NB)
    us = ops.time_first_kernel(lambda: ops.conv2d(pc, x, out=out, act=ops.ACT_RELU))
    torch.cuda.synchronize()
    lib.scf_wino_trace_set(None, 0)
    t = tr.cpu().numpy().astype('int64') & 0xffffffff
    print(f'== {name}: kernel {us:.1f} us (traced build)')
    names = ['DMA issue', 'MFMA g1 (+operand wait)', 'MFMA g2 (+transform s0)', 'MFMA g3', 'MFMA g4 (+transform s1)', 'vmcnt wait', 'barrier']
    acc = [[] for _ in names]
    tot, pro, epi = [], [], []
    for bi in range(NB):
        for wv in range(4):
            row = t[bi, wv]
            if not row[1]:
                continue
            pro.append((row[1] - row[0]) & 0xffffffff)
            for c in range(4, 14):
                base = 2 + 8 * c
                prev = row[2 + 8 * (c - 1) + 6]
                st = [row[base + k] for k in range(7)]
                if not all(st) or not prev:
                    continue
                d = [(st[0] - prev) & 0xffffffff] + [(st[k] - st[k - 1]) & 0xffffffff for k in range(1, 7)]
                for k in range(7):
                    acc[k].append(d[k])
                tot.append((st[6] - prev) & 0xffffffff)
            if row[123]:
                epi.append((row[123] - row[122]) & 0xffffffff)
    import numpy as np
    fr = []
    for bi in range(NB):
        row = t[bi, 0]
        if row[120] and row[118]:
            drt = (row[120] - row[118]) & 0xffffffff
            dcy = (row[121] - row[119]) & 0xffffffff
            if drt > 100:
                fr.append(dcy / drt * 100.0)
    print(f'   shader clock over the blocks: mean {np.mean(fr):.0f} MHz (min {np.min(fr):.0f}, max {np.max(fr):.0f}); block duration {np.mean([((t[bi,0,120]-t[bi,0,118]) & 0xffffffff) for bi in range(NB) if t[bi,0,120]]) / 100.0:.1f} us')
    print(f'   prologue {np.mean(pro):.0f} cycles; per chunk {np.mean(tot):.0f} (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}); '
          f'output transform to exchange {np.mean(epi):.0f}')
    for k, nm in enumerate(names):
        a = np.array(acc[k])
        print(f'     {nm:26s} mean {a.mean():7.0f}  p10 {np.percentile(a, 10):6.0f}  p50 {np.percentile(a, 50):6.0f}  p90 {np.percentile(a, 90):6.0f}')
    # two blocks of one CU: print the raw chunk timeline of the first pair found
    cus = collections.defaultdict(list)
    for bi in range(min(NB, 512)):
        hw, xcc = int(t[bi, 0, 126]), int(t[bi, 0, 127]) & 0xf
        cus[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)].append(bi)
    for key, blks in list(cus.items())[:1]:
        print('   CU', key, 'blocks', blks)
        t00 = min(int(t[bi, 0, 0]) for bi in blks)
        for bi in blks[:2]:
            for wv in (0, 1):
                row = t[bi, wv]
                simd = (int(row[126]) >> 4) & 3
                s = f'     block {bi} wave {wv} simd {simd}: entry {int(row[0]) - t00} prologue done {int(row[1]) - t00} |'
                for c in range(4, 8):
                    base = 2 + 8 * c
                    s += f' c{c}: ' + ' '.join(str(int(row[base + k]) - t00) for k in (0, 1, 4, 6))
                print(s)
