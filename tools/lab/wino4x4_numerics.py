"""CPU study (no GPU): forward error of Winograd F(4x4, 3x3) in fp32 for several point sets, in the unit the GPU
stress tests use (eps * sum|w||x|, eps = 2^-24, max over outputs), on the same operand classes
(tests/test_gpu_ops.py: dc50 / relu / wide_weights / dc50_wide) next to F(2x2, 3x3) and a plain fp32 fma chain.
    python tools/lab/wino4x4_numerics.py
Transforms are evaluated as fp32 matrix products (B^T d B, A^T m A), the weight transform in fp64 rounded once
(what the C packers do), the channel contraction in fp32."""
from fractions import Fraction as Fr
import numpy as np


def cook_toom(points, m, r):
    """-> (AT m x n, G n x r, BT n x n) as float64 arrays, n = m + r - 1, last point = infinity"""
    n = m + r - 1
    a = [Fr(p) for p in points]
    assert len(a) == n - 1
    def ev(cols):
        rows = [[ai ** j for j in range(cols)] for ai in a]
        rows.append([Fr(0)] * (cols - 1) + [Fr(1)])
        return rows
    C = ev(n)
    # exact inverse
    M = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(C)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        pv = M[c][c]
        M[c] = [v / pv for v in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                f = M[i][c]
                M[i] = [vi - f * vc for vi, vc in zip(M[i], M[c])]
    Cinv = [row[n:] for row in M]
    f = []
    for i in range(n - 1):
        v = Fr(1)
        for j in range(n - 1):
            if j != i:
                v *= a[i] - a[j]
        f.append(v)
    f.append(Fr(1))
    BT = [[f[i] * Cinv[j][i] for j in range(n)] for i in range(n)]
    G = [[v / f[i] for v in row] for i, row in enumerate(ev(r))]
    AT = [[ev(m)[j][i] for j in range(n)] for i in range(m)]
    to = lambda X: np.array([[float(v) for v in row] for row in X])
    return to(AT), to(G), to(BT)


def check_1d(AT, G, BT, m, r):
    rng = np.random.default_rng(0)
    g = rng.standard_normal(r); d = rng.standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)


def operands(kind, cin, cout, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((cin, H, W))
    w = rng.standard_normal((cout, cin, 3, 3)) * (1.0 / (cin * 9)) ** 0.5
    if kind in ('dc50', 'dc50_wide'):
        x = x + 50.0
    if kind == 'relu':
        x = np.abs(x)
    if kind in ('wide_weights', 'dc50_wide'):
        w = w * 10.0 ** (rng.random(w.shape) * 3.0 - 1.5)
    return x.astype(np.float32), w.astype(np.float32)


def direct64(x, w):
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + H, kx:kx + W])
    return out


def direct32(x, w):
    """an fp32 chain over K = cin * 9 (numpy matmul accumulates in fp32)"""
    cin, H, W = x.shape
    xp = np.zeros((cin, H + 2, W + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), np.float32)
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, ky, kx], xp[:, ky:ky + H, kx:kx + W], dtype=np.float32)
    return out


def winograd32(x, w, AT, G, BT, m):
    n = m + 2
    cin, H, W = x.shape
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((cin, th * m + 2, tw * m + 2), np.float32); xp[:, 1:H + 1, 1:W + 1] = x
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(np.float32)          # fp64, rounded once
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    tiles = np.stack([[xp[:, ty * m:ty * m + n, tx * m:tx * m + n] for tx in range(tw)] for ty in range(th)])  # th tw c n n
    V = np.einsum('ij,yxcjk->yxcik', BT32, tiles, dtype=np.float32)
    V = np.einsum('yxcik,lk->yxcil', V, BT32, dtype=np.float32)
    M = np.einsum('ocil,yxcil->yxoil', U, V, dtype=np.float32)
    Y = np.einsum('ij,yxojk->yxoik', AT32, M, dtype=np.float32)
    Y = np.einsum('yxoik,lk->yxoil', Y, AT32, dtype=np.float32)
    out = Y.transpose(2, 0, 3, 1, 4).reshape(w.shape[0], th * m, tw * m)
    return out[:, :H, :W]


def main():
    sets = {
        'F(2x2,3x3) 0,+-1': (2, [0, 1, -1]),
        'F(4x4,3x3) 0,+-1,+-2': (4, [0, 1, -1, 2, -2]),
        'F(4x4,3x3) 0,+-1,+-1/2': (4, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
        'F(4x4,3x3) 0,+-1,2,-1/2': (4, [0, 1, -1, 2, Fr(-1, 2)]),
        'F(4x4,3x3) 0,+-1,1/2,-2': (4, [0, 1, -1, Fr(1, 2), -2]),
        'F(4x4,3x3) 0,+-1,+-1/sqrt2~(+-3/4)': (4, [0, 1, -1, Fr(3, 4), Fr(-3, 4)]),
        'F(3x3,3x3) 0,+-1,2': (3, [0, 1, -1, 2]),
        'F(3x3,3x3) 0,+-1,1/2': (3, [0, 1, -1, Fr(1, 2)]),
        'F(3x3,3x3) 0,+-1,-1/2': (3, [0, 1, -1, Fr(-1, 2)]),
    }
    eps = 2.0 ** -24
    shapes = [(64, 64, 48, 48), (128, 96, 32, 32), (256, 64, 24, 24)]
    kinds = ['normal', 'dc50', 'relu', 'wide_weights', 'dc50_wide']
    print('max over outputs of |err| / (eps * sum|w||x|); columns:', ' '.join(kinds))
    rows = {}
    for name, (m, pts) in sets.items():
        AT, G, BT = cook_toom(pts, m, 3)
        check_1d(AT, G, BT, m, 3)
        rows[name] = []
    rows['direct fp32 chain'] = []
    for kind in kinds:
        worst = {k: 0.0 for k in rows}
        for si, (cin, cout, H, W) in enumerate(shapes):
            x, w = operands(kind, cin, cout, H, W, 100 + si)
            want = direct64(x, w)
            scale = direct64(np.abs(x), np.abs(w))
            worst['direct fp32 chain'] = max(worst['direct fp32 chain'], float((np.abs(direct32(x, w) - want) / (eps * scale)).max()))
            for name, (m, pts) in sets.items():
                AT, G, BT = cook_toom(pts, m, 3)
                got = winograd32(x, w, AT, G, BT, m)
                worst[name] = max(worst[name], float((np.abs(got - want) / (eps * scale)).max()))
        for k in rows:
            rows[k].append(worst[k])
    for k, v in rows.items():
        print(f'{k:42s}', ' '.join(f'{e:8.1f}' for e in v))
    for name in ('F(4x4,3x3) 0,+-1,+-1/2', 'F(4x4,3x3) 0,+-1,2,-1/2'):
        m, pts = sets[name]
        AT, G, BT = cook_toom(pts, m, 3)
        np.set_printoptions(linewidth=160, suppress=True)
        print(name, '\nBT=\n', BT, '\nG=\n', G, '\nAT=\n', AT)




def winograd32_mixed(x, w, T_rows, T_cols, mr, mc):
    """F(mr x mc, 3x3): different 1-D forms along rows and columns"""
    ATr, Gr, BTr = T_rows
    ATc, Gc, BTc = T_cols
    nr, nc = mr + 2, mc + 2
    cin, H, W = x.shape
    th, tw = (H + mr - 1) // mr, (W + mc - 1) // mc
    xp = np.zeros((cin, th * mr + 2, tw * mc + 2), np.float32); xp[:, 1:H + 1, 1:W + 1] = x
    U = np.einsum('ij,ocjk,lk->ocil', Gr, w.astype(np.float64), Gc).astype(np.float32)
    f = lambda a: a.astype(np.float32)
    tiles = np.stack([[xp[:, ty * mr:ty * mr + nr, tx * mc:tx * mc + nc] for tx in range(tw)] for ty in range(th)])
    V = np.einsum('ij,yxcjk->yxcik', f(BTr), tiles, dtype=np.float32)
    V = np.einsum('yxcik,lk->yxcil', V, f(BTc), dtype=np.float32)
    M = np.einsum('ocil,yxcil->yxoil', U, V, dtype=np.float32)
    Y = np.einsum('ij,yxojk->yxoik', f(ATr), M, dtype=np.float32)
    Y = np.einsum('yxoik,lk->yxoil', Y, f(ATc), dtype=np.float32)
    out = Y.transpose(2, 0, 3, 1, 4).reshape(w.shape[0], th * mr, tw * mc)
    return out[:, :H, :W]


def mixed():
    eps = 2.0 ** -24
    shapes = [(64, 64, 48, 48), (128, 96, 32, 32), (256, 64, 24, 24)]
    kinds = ['normal', 'dc50', 'relu', 'wide_weights', 'dc50_wide']
    forms = {'F(4x2) 0,+-1,2,-1/2 x 0,+-1': ((4, [0, 1, -1, 2, Fr(-1, 2)]), (2, [0, 1, -1])),
             'F(4x2) 0,+-1,1/2,-2 x 0,+-1': ((4, [0, 1, -1, Fr(1, 2), -2]), (2, [0, 1, -1])),
             'F(3x2) 0,+-1,1/2 x 0,+-1': ((3, [0, 1, -1, Fr(1, 2)]), (2, [0, 1, -1])),
             'F(4x3) 0,+-1,2,-1/2 x 0,+-1,1/2': ((4, [0, 1, -1, 2, Fr(-1, 2)]), (3, [0, 1, -1, Fr(1, 2)]))}
    for name, ((mr, pr), (mc, pc)) in forms.items():
        Tr, Tc = cook_toom(pr, mr, 3), cook_toom(pc, mc, 3)
        res = []
        for kind in kinds:
            worst = 0.0
            for si, (cin, cout, H, W) in enumerate(shapes):
                x, w = operands(kind, cin, cout, H, W, 100 + si)
                want = direct64(x, w)
                scale = direct64(np.abs(x), np.abs(w))
                got = winograd32_mixed(x, w, Tr, Tc, mr, mc)
                worst = max(worst, float((np.abs(got - want) / (eps * scale)).max()))
            res.append(worst)
        print(f'{name:42s}', ' '.join(f'{e:8.1f}' for e in res))


if __name__ == '__main__':
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == 'mixed':
        mixed()
    else:
        main()
