"""Timelines of ALL blocks of one full-grid convolution launch (library built with -DSCF_CONV_LAB):
which blocks share a CU (HW_ID / XCC_ID), and for how much of the kernel neither of the two waves
that share a SIMD is inside its MFMA phase (matrix pipe necessarily idle)."""
import sys, os, ctypes as C, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scflow_amd import ops, _lib
DEV = 'cuda:0'
lib = _lib.load()
lib.scf_conv_trace_set.argtypes = [C.c_void_p, C.c_int]
cases = [('heads 128->512 3x3', 32, 128, 512, (3, 3), 1, 1, 32, 32),
         ('GRU zr 384->256 5x1', 32, 384, 256, (5, 1), 1, (2, 0), 32, 32),
         ('enc 64->64 3x3 @128', 64, 64, 64, (3, 3), 1, 1, 128, 128)]
NB = 4096
for name, n, cin, cout, k, stride, pad, H, W in cases:
    x = torch.randn((n, cin, H, W), device=DEV)
    w = torch.randn((cout, cin, *k), device=DEV) * 0.05
    b = torch.randn((cout,), device=DEV)
    pc = ops.PackedConv.from_weight(w, b, stride=stride, padding=pad)
    for _ in range(3):
        ops.conv2d(pc, x, act=ops.ACT_RELU)
    tr = torch.zeros((NB, 4, 128), dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(C.c_void_p(tr.data_ptr()), NB)
    us = ops.time_first_kernel(lambda: ops.conv2d(pc, x, act=ops.ACT_RELU))
    torch.cuda.synchronize()
    lib.scf_conv_trace_set(None, 0)
    t = tr.cpu().numpy()
    used = [bi for bi in range(NB) if t[bi, 0, 0]]
    t0 = min(int(t[bi, 0, 0]) for bi in used)
    fl = 2.0 * n * cout * cin * k[0] * k[1] * (H // stride) * (W // stride)
    print(f'== {name}: kernel {us:.1f} us  {fl / us * 1e-6:.1f} TFLOP/s, {len(used)} blocks traced')
    fr = [(int(t[bi, 0, 126]) - int(t[bi, 0, 125])) / max(1, (int(t[bi, 0, 3]) - int(t[bi, 0, 0]))) * 100.0 for bi in used if t[bi, 0, 3]]
    print(f'   shader clock during the blocks (s_memtime ticks per us): min {min(fr):.0f} mean {sum(fr) / len(fr):.0f} max {max(fr):.0f} MHz')
    cus = collections.defaultdict(list)
    for bi in used:
        hw = int(t[bi, 0, 127])
        hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
        key = (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15)
        cus[key].append((bi, (hwid >> 16) & 15))
    print('   CUs seen', len(cus), ' blocks per CU (first 8):', [len(v) for v in list(cus.values())[:8]])
    # matrix-pipe idle bound per SIMD: time inside [first mfma start, last mfma end] where no wave of that SIMD is in an MFMA phase
    tot_idle = tot_span = 0.0
    shown = 0
    for key, blks in cus.items():
        for wv in range(4):
            iv = []
            for bi, tg in blks:
                row = t[bi, wv]
                c = 0
                while 4 + 4 * c + 3 < 125 and row[4 + 4 * c + 3]:
                    iv.append(((int(row[4 + 4 * c + 2]) - t0) * 0.01, (int(row[4 + 4 * c + 3]) - t0) * 0.01))
                    c += 1
            if not iv:
                continue
            iv.sort()
            lo, hi = iv[0][0], max(e for _, e in iv)
            busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for a, e in iv[1:]:
                if a > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = a, e
                else:
                    cur_e = max(cur_e, e)
            busy += cur_e - cur_s
            tot_idle += (hi - lo) - busy
            tot_span += hi - lo
        if shown < 2:
            shown += 1
            print('   CU', key, 'blocks (id, tg_id):', blks)
            for bi, tg in blks[:4]:
                row = t[bi, 0]
                c = 0
                s = f'     block {bi} tg {tg} wave0: start {(int(row[0]) - t0) * 0.01:.2f} |'
                while 4 + 4 * c + 3 < 125 and row[4 + 4 * c + 3] and c < 10:
                    a, bb, cc, d = ((int(row[4 + 4 * c + i]) - t0) * 0.01 for i in range(4))
                    s += f' c{c}: [{cc:.1f}-{d:.1f}]'
                    c += 1
                print(s)
    print(f'   no-wave-in-MFMA-phase fraction of the SIMD spans: {tot_idle / tot_span:.3f}')
