"""K split across blocks (scf_conv_desc.k_slices) on the pose head's three stride-2 convolutions: parity of the summed
partial tensors against the unsliced launch and torch fp64, and launch time per slice count.
    python tools/lab/kslice_sweep.py [batch ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import torch.nn.functional as F

from scflow_amd import ops


def time_us(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ops.conv_timing(True)
    for _ in range(reps):
        fn()
    ev = ops.conv_timing(False)
    us = sorted(e[0] for e in ev)
    return us[len(us) // 2]


def main():
    batches = [int(a) for a in sys.argv[1:]] or [32, 8, 1]
    torch.manual_seed(0)
    for n in batches:
        for (cin, cout, hw) in [(224, 128, 32), (128, 128, 16), (128, 128, 8)]:
            x = torch.randn(n, cin, hw, hw, device='cuda').abs()
            w = torch.randn(cout, cin, 3, 3, device='cuda') * (1.0 / (cin * 9)) ** 0.5
            pc = ops.PackedConv.from_weight(w, None, stride=2, padding=1)
            want = F.conv2d(x.double(), w.double(), None, stride=2, padding=1)
            base = ops.conv2d(pc, x)
            line = f'N={n:2d} {cin}->{cout} 3x3/s2 @{hw // 2}x{hw // 2}:'
            for S in (1, 2, 3, 4, 6, 8):
                try:
                    if S == 1:
                        got = base
                        t = time_us(lambda: ops.conv2d(pc, x))
                    else:
                        parts = ops.conv2d(pc, x, kslices=S)
                        got = parts[0].clone()
                        for s_ in range(1, S):
                            got += parts[s_]
                        t = time_us(lambda: ops.conv2d(pc, x, kslices=S))
                    err = float((got.double() - want).abs().max())
                    dif = float((got - base).abs().max())
                    line += f'  S={S}: {t:6.1f} us (err {err:.1e}, vs S=1 {dif:.1e})'
                except Exception as e:      # unsupported slice count for this layer
                    line += f'  S={S}: {type(e).__name__}'
            print(line, flush=True)
    # GroupNorm on partial tensors == GroupNorm on their sum
    x = torch.randn(4, 224, 32, 32, device='cuda')
    w = torch.randn(128, 224, 3, 3, device='cuda') * 0.02
    pc = ops.PackedConv.from_weight(w, None, stride=2, padding=1)
    g, b = torch.rand(128, device='cuda') + 0.5, torch.randn(128, device='cuda')
    parts = ops.conv2d(pc, x, kslices=4)
    a = ops.group_norm_relu(parts, g, b, 32)
    ref = ops.group_norm_relu(((parts[0] + parts[1]) + parts[2]) + parts[3], g, b, 32)
    print('group_norm_relu(parts) == group_norm_relu(sum in order):', torch.equal(a, ref))


if __name__ == '__main__':
    main()
