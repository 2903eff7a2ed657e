"""r6: randomised sweep of the ConvGRU cell (scf_sepconv_gru / scf_sepconv_gru_ctx through scflow_amd.modules.ConvGRU) against the
oracle: state / input widths other than the refiner's 128 / 256, map sizes, both GRU types, with and without the hoisted context
term.   python tools/lab/gru_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from scflow_amd.modules import ConvGRU
DEV = 'cuda:0'


def run(cases: int, seed: int, verbose: bool = True) -> int:
    """-> number of failing cases"""
    _print = print if verbose else (lambda *a, **k: None)
    rs = random.Random(seed)
    torch.set_num_threads(16)
    bad = 0
    for ci in range(cases):
        hc = rs.choice([32, 64, 96, 128, 128])
        cc = rs.choice([0, 32, 64, 128])
        xc = cc + rs.choice([32, 64, 128, 130, 66])
        kind = rs.choice(['SeqConv', 'SeqConv', 'Conv'])
        if rs.random() < 0.5:
            h, w = rs.choice([(8, 8), (16, 16), (32, 32), (60, 80), (30, 40), (12, 20)])
        else:
            h, w = rs.randint(4, 48), rs.randint(4, 64)
        n = rs.choice([1, 2, 3, 8, 32])
        while n > 1 and n * h * w * (hc + xc) * hc * 30 > 4e9:
            n //= 2
        g = torch.Generator().manual_seed(seed * 4099 + ci)
        gru = ConvGRU(hc, xc, kind)
        sd = {}
        for k_, v in gru.state_dict().items():
            fan = v[0].numel() if v.dim() > 1 else 1
            sd[k_] = (torch.randn(v.shape, generator=g) * ((1.0 / fan) ** 0.5 if v.dim() > 1 else 0.1))
        gru.load_state_dict(sd, strict=True)
        gru = gru.to(DEV)
        h0 = torch.tanh(torch.randn((n, hc, h, w), generator=g))
        x = torch.randn((n, xc, h, w), generator=g)
        want = oracle.sepconv_gru(h0, x, {'gru.' + k_: v for k_, v in sd.items()}, 'gru.')
        use_ctx = cc > 0 and rs.random() < 0.6
        tag = f'case {ci}: {kind} N{n} h{hc} x{xc} (ctx {cc if use_ctx else 0}) @{h}x{w}'
        try:
            hx = torch.cat([h0, x], 1).to(DEV)
            if use_ctx:
                ctx = gru.context_terms(hx[:, hc:hc + cc])
                got = gru.forward_inplace(hx, ctx, cc)
            else:
                got = gru.forward_inplace(hx)
            torch.cuda.synchronize()
        except Exception as exc:
            print('RAISED', tag, repr(exc)[:200], flush=True)
            bad += 1
            continue
        err = float((got.cpu() - want).abs().max())
        ok = err <= 2e-5 and bool(torch.isfinite(got).all())
        bad += 0 if ok else 1
        _print(f'{"ok  " if ok else "FAIL"} {tag}: max |dh| {err:.2e}', flush=True)
        if not ok and not verbose:
            print('FAIL', tag, f'{err:.2e}', flush=True)
    print('FUZZ', 'FAILED' if bad else 'ok', bad, 'of', cases)
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
