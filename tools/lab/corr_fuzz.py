"""r6: randomised sweep of the correlation pair (scf_corr_build_ex + scf_corr_lookup_ex) against the CPU oracle: map sizes,
channel counts, radii, level counts, layouts (row-major / the preferred tiling / a random legal mask), flows with out-of-range and
integer-valued entries.   python tools/lab/corr_fuzz.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from scflow_amd import ops
DEV = 'cuda:0'
def run(cases: int, seed: int, verbose: bool = True) -> int:
    """-> number of failing cases"""
    _print = print if verbose else (lambda *a, **k: None)
    rs = random.Random(seed)
    torch.set_num_threads(16)
    bad = 0
    for ci in range(cases):
        if rs.random() < 0.5:
            h, w = rs.choice([(8, 8), (16, 16), (32, 32), (12, 20), (30, 40), (24, 32), (15, 20), (60, 80)])
        else:
            h, w = rs.randint(4, 48), rs.randint(4, 64)
        L = rs.randint(1, 4)
        while L > 1 and (min(h, w) >> (L - 1)) < 1:
            L -= 1
        r = rs.choice([1, 2, 3, 4, 4, 4, 5, 6, 7])
        c = rs.choice([1, 2, 4, 7, 16, 32, 64, 96, 100, 128, 256])
        n = rs.choice([1, 2, 3, 5])
        while n > 1 and n * (h * w) ** 2 * 4 * 1.34 > 400e6:
            n -= 1
        if (h * w) ** 2 * 4 * 1.34 > 400e6:
            continue
        pref = ops.pyramid_layout(h, w, r, L)
        mask = rs.choice([0, pref, pref])
        g = torch.Generator().manual_seed(seed * 7919 + ci)
        f1, f2 = torch.randn((n, c, h, w), generator=g), torch.randn((n, c, h, w), generator=g)
        flow = torch.randn((n, 2, h, w), generator=g) * rs.choice([0.5, 3.0, 12.0])
        flow[0, :, 0, 0] = torch.tensor([-3.0 * w, 2.0])
        flow[0, :, -1, -1] = torch.tensor([float(w), float(h)])
        flow[0, :, h // 2, w // 2] = torch.tensor([1.0, -2.0])
        tag = f'case {ci}: N{n} C{c} {h}x{w} r{r} L{L} tiled {mask:04b} (preferred {pref:04b})'
        try:
            pyr = ops.corr_build(f1.to(DEV), f2.to(DEV), L, tiled_levels=mask)
            got = ops.corr_lookup(pyr, flow.to(DEV), r, tiled_levels=mask)
            torch.cuda.synchronize()
        except Exception as exc:
            print('RAISED', tag, repr(exc)[:200], flush=True)
            bad += 1
            continue
        want_p = oracle.correlation_pyramid(f1, f2, L)
        want = oracle.corr_lookup(want_p, flow.clone(), r)
        e_l = float((got.cpu() - want).abs().max())
        e_p = 0.0
        for l in range(L):
            lv = pyr[l]
            if (mask >> l) & 1:
                lv = ops.untile_level(lv, h >> l, w >> l)
            e_p = max(e_p, float((lv.cpu() - want_p[l]).abs().max()))
        # pyramid: an fp32 dot product of c terms; lookup: the reference's coordinate round trip (x * 2 / (W - 1) - 1 and
        # back, corr_lookup.py:64-65) moves a tap by ~W * 6e-8 px, times the local slope of the volume (~ its magnitude)
        vmax = max(1.0, float(want_p[0].abs().max()))
        lim_p = 3e-5 * max(1.0, (c / 64.0) ** 0.5)
        lim = 8e-6 * vmax * max(1.0, max(h, w) / 32.0) ** 0.5 + lim_p
        ok = e_l <= lim and e_p <= lim_p and bool(torch.isfinite(got).all())
        bad += 0 if ok else 1
        print(f'{"ok  " if ok else "FAIL"} {tag}: pyramid err {e_p:.2e}, lookup err {e_l:.2e} (limit {lim:.1e})', flush=True)
    print('FUZZ', 'FAILED' if bad else 'ok', bad, 'of', cases)
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
