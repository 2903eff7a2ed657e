"""r6: randomised sweep of scf_conv2d over shapes / epilogues against a CPU fp64 convolution -- looks for dispatch corner cases the
hand-picked test shapes miss.  Every case prints the kernel family that ran; failures print the case so that it can be pinned as a
test.   python tools/lab/conv_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from scflow_amd import ops
DEV = 'cuda:0'
def run(cases: int, seed: int, verbose: bool = True) -> int:
    """-> number of failing cases"""
    _print = print if verbose else (lambda *a, **k: None)
    rs = random.Random(seed)
    torch.set_num_threads(16)
    KS = [((1, 1), (0, 0)), ((3, 3), (1, 1)), ((3, 3), (1, 1)), ((1, 5), (0, 2)), ((5, 1), (2, 0)), ((7, 7), (3, 3)), ((5, 5), (2, 2)), ((3, 3), (0, 0)), ((3, 1), (1, 0))]
    bad, fam = 0, {}
    for ci in range(cases):
        k, p = rs.choice(KS)
        stride = rs.choice([1, 1, 1, 2])
        cin = rs.choice([1, 2, 3, 4, 8, 16, 24, 30, 32, 40, 60, 64, 72, 96, 100, 128, 130, 224, 256, 324])
        cout = rs.choice([1, 2, 3, 4, 8, 20, 32, 40, 63, 64, 96, 126, 128, 192, 256])
        if rs.random() < 0.5:
            H, W = rs.choice([(8, 8), (16, 16), (32, 32), (64, 64), (60, 80), (30, 40), (15, 20)])
        else:
            H, W = rs.randint(max(4, k[0]), 70), rs.randint(max(4, k[1]), 90)
        budget = 2.5e9      # flops per case (CPU fp64 reference stays in seconds)
        nmax = max(1, int(budget / (2.0 * cin * k[0] * k[1] * cout * H * W / stride ** 2)))
        n = rs.choice([1, 2, 3, 5, 8, 17, 32, 64])
        n = max(1, min(n, nmax))
        two = cin >= 16 and rs.random() < 0.25
        c0 = rs.choice([c for c in (8, 16, 32, 64, 96, 128, 192) if c < cin] or [0]) if two else 0
        bias, relu, res = rs.random() < 0.8, rs.random() < 0.6, rs.random() < 0.2
        bn = rs.random() < 0.15
        g = torch.Generator().manual_seed(seed * 100003 + ci)
        x = torch.randn((n, cin, H, W), generator=g)
        w = torch.randn((cout, cin, *k), generator=g) * (1.0 / (cin * k[0] * k[1])) ** 0.5
        b = torch.randn((cout,), generator=g) * 0.1 if bias else None
        ho, wo = (H + 2 * p[0] - k[0]) // stride + 1, (W + 2 * p[1] - k[1]) // stride + 1
        if ho < 1 or wo < 1:
            continue
        r = torch.randn((n, cout, ho, wo), generator=g) if res else None
        bnp = None
        want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=p)
        if bn:
            ga, be, mu, var = (torch.randn((cout,), generator=g) * 0.2 + 1, torch.randn((cout,), generator=g) * 0.1,
                               torch.randn((cout,), generator=g) * 0.1, torch.rand((cout,), generator=g) * 0.5 + 0.5)
            bnp = [t.to(DEV) for t in (ga, be, mu, var)]
            want = (want - mu.double()[None, :, None, None]) / torch.sqrt(var.double()[None, :, None, None] + 1e-5) * ga.double()[None, :, None, None] + be.double()[None, :, None, None]
        if res:
            want = want + r.double()
        if relu:
            want = torch.relu(want)
        tag = f'case {ci}: N{n} {cin}->{cout} {k[0]}x{k[1]}/s{stride} pad{p} @{H}x{W} c0={c0} bias={int(bias)} bn={int(bn)} res={int(res)} relu={int(relu)}'
        try:
            pc = ops.PackedConv.from_weight(w.to(DEV), None if b is None else b.to(DEV), stride=stride, padding=p, bn=bnp)
            xd = x.to(DEV)
            x0, x1 = (xd[:, :c0], xd[:, c0:]) if c0 else (xd, None)
            with ops.record_conv_kernels() as ran:
                got = ops.conv2d(pc, x0, x1, res=None if r is None else r.to(DEV), act=ops.ACT_RELU if relu else ops.ACT_NONE)
            torch.cuda.synchronize()
        except Exception as exc:
            print('RAISED', tag, repr(exc)[:200], flush=True)
            bad += 1
            continue
        name = ran[0][1] if ran else '?'
        fam[name] = fam.get(name, 0) + 1
        scale = float((x.abs().double().mean() * w.abs().double().sum(dim=(1, 2, 3)).max()))        # ~ sum |w||x| per output
        err = float((got.cpu().double() - want).abs().max())
        lim = 40 * 1.19e-7 * max(scale, 1.0) + 1e-6
        ok = err <= lim and bool(torch.isfinite(got).all())
        if not ok:
            bad += 1
        print(f'{"ok  " if ok else "FAIL"} {tag} [{name}] err {err:.2e} (limit {lim:.2e})', flush=True)
    print('families:', fam)
    print('FUZZ', 'FAILED' if bad else 'ok', bad, 'of', cases)
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
