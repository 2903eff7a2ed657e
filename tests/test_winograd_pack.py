"""CPU: the Winograd weight packers of the C ABI (scf_pack_conv_weight_wino / _wino1d) against the algebra
the kernels implement -- the packed U, read back through the layout documented in include/scflow_hip.h,
combined with the kernels' input / output transforms in numpy, reproduces the convolution."""
import ctypes as C

import numpy as np
import pytest
import torch

from scflow_amd import _lib, ops


# independent restatement of the two packings (torch, fp64 einsum) -- the product packs through the C entry points
def _ref_pack_wino(weight):
    """(Cout, Cin, 3, 3) -> U = G g G^T in conv_wino.hip's layout (scf_pack_conv_weight_wino):
    [chunk][Cout / 32][4 i' + j][channel & 1][Cout % 32][channel >> 1 & 1] with the rows i of the transform
    domain stored in the order i' -> 0, 1, 3, 2 (conv_wino.hip gives each of the two rows-halves one wave), 4 channels per chunk,
    computed in double and rounded once; zero padded."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError('Winograd packing: 3x3 kernels')
    g = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
                     dtype=torch.float64, device=weight.device)
    u = torch.einsum('ia,ocab,jb->ijoc', g, weight.double(), g)[[0, 1, 3, 2]].reshape(16, cout, cin)   # rows stored 0, 1, 3, 2
    f, nchunk = (cout + 31) // 32, (cin + 3) // 4
    full = torch.zeros((16, f * 32, nchunk * 4), dtype=torch.float64, device=weight.device)
    full[:, :cout, :cin] = u
    # [xi][frag][m][chunk][s][kh] -> [chunk][frag][xi][kh][m][s]
    full = full.reshape(16, f, 32, nchunk, 2, 2).permute(3, 1, 0, 5, 2, 4)
    return full.contiguous().float().reshape(-1)


def _ref_pack_wino1d(weight):
    """(Cout, Cin, 1, 5) or (Cout, Cin, 5, 1) -> U = G g in conv_wino1d.hip's layout (scf_pack_conv_weight_wino1d):
    [chunk][Cout / 32][position i][channel & 1][Cout % 32][channel >> 1 & 3], 8 channels per chunk; G = the 6 x 5
    matrix of the points 0, 1, -1, 2, -2, infinity; computed in double and rounded once; zero padded."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((1, 5), (5, 1)):
        raise ValueError('F(2, 5) packing: 1x5 / 5x1 kernels')
    pts = [0.0, 1.0, -1.0, 2.0, -2.0]
    g = torch.zeros((6, 5), dtype=torch.float64)
    for i, a in enumerate(pts):
        nrm = 1.0
        for k, b in enumerate(pts):
            if k != i:
                nrm *= a - b
        g[i] = torch.tensor([a ** k / nrm for k in range(5)], dtype=torch.float64)
    g[5, 4] = 1.0
    wd = weight.reshape(cout, cin, 5).double()
    g = g.to(weight.device)
    u = torch.zeros((6, cout, cin), dtype=torch.float64, device=weight.device)
    for k in range(5):                   # the C packer's order of adds: bit-identical packings
        u = u + g[:, k, None, None] * wd[None, :, :, k]
    f, nchunk = (cout + 31) // 32, (cin + 7) // 8
    full = torch.zeros((6, f * 32, nchunk * 8), dtype=torch.float64, device=weight.device)
    full[:, :cout, :cin] = u
    # [i][frag][m][chunk][s][kh] -> [chunk][frag][i][kh][m][s]
    full = full.reshape(6, f, 32, nchunk, 4, 2).permute(3, 1, 0, 5, 2, 4)
    return full.contiguous().float().reshape(-1)


@pytest.fixture(scope='module')
def lib():
    return _lib.load()


def _unpack(packed, npos, cout, cin, kc):
    """[chunk][frag][pos][k-half][32][kc / 2] -> U[pos][cout][cin] (storage position, not transform index)"""
    f, nchunk, ks = (cout + 31) // 32, (cin + kc - 1) // kc, kc // 2
    a = packed.reshape(nchunk, f, npos, 2, 32, ks)
    u = np.zeros((npos, f * 32, nchunk * kc))
    for ch in range(nchunk):
        for s in range(ks):
            for kh in range(2):
                u[:, :, ch * kc + 2 * s + kh] = a[ch, :, :, kh, :, s].transpose(1, 0, 2).reshape(npos, f * 32)
    return u[:, :cout, :cin], u


def test_wino2d_packing_reproduces_the_convolution(lib):
    rng = np.random.default_rng(3)
    cout, cin = 40, 10
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)).astype(np.float32))
    n = lib.scf_pack_conv_weight_wino_size(cout, cin)
    assert n == ((cin + 3) // 4) * 2 * 16 * 128
    host = torch.empty(n)
    assert lib.scf_pack_conv_weight_wino(w.data_ptr(), cout, cin, host.data_ptr()) == 0
    assert torch.equal(host, _ref_pack_wino(w))           # the torch restatement: bit-identical
    assert torch.equal(host, ops.pack_conv_weight_wino(w))   # what PackedConv uses = the C packer
    u, full = _unpack(host.numpy().astype(np.float64), 16, cout, cin, 4)
    assert np.all(full[:, cout:, :] == 0) and np.all(full[:, :, cin:] == 0)      # zero padding
    # storage rows of the transform domain: 0, 1, 3, 2
    order = [0, 1, 3, 2]
    U = np.zeros((4, 4, cout, cin))
    for ip, i in enumerate(order):
        for j in range(4):
            U[i, j] = u[4 * ip + j]
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    d = rng.standard_normal((cin, 4, 4))
    V = np.einsum('ia,cab,jb->ijc', BT, d, BT)
    M = np.einsum('ijoc,ijc->ijo', U, V)
    Y = np.einsum('ai,ijo,bj->oab', AT, M, AT)
    want = np.zeros((cout, 2, 2))
    wd = w.numpy().astype(np.float64)
    for a in range(2):
        for b in range(2):
            want[:, a, b] = np.einsum('ocij,cij->o', wd, d[:, a:a + 3, b:b + 3])
    assert np.abs(Y - want).max() < 1e-5         # U is rounded to fp32 once; everything else is exact here


def test_wino1d_packing_reproduces_the_convolution(lib):
    rng = np.random.default_rng(4)
    cout, cin = 64, 20
    w = torch.from_numpy(rng.standard_normal((cout, cin, 1, 5)).astype(np.float32))
    n = lib.scf_pack_conv_weight_wino1d_size(cout, cin)
    assert n == ((cin + 7) // 8) * 2 * 1536
    host = torch.empty(n)
    taps = w.reshape(cout, cin, 5).contiguous()
    assert lib.scf_pack_conv_weight_wino1d(taps.data_ptr(), cout, cin, host.data_ptr()) == 0
    assert torch.equal(host, _ref_pack_wino1d(w))
    assert torch.equal(host, ops.pack_conv_weight_wino1d(w))
    assert torch.equal(host, ops.pack_conv_weight_wino1d(w.reshape(cout, cin, 5, 1)))     # 5x1: the same taps
    u, full = _unpack(host.numpy().astype(np.float64), 6, cout, cin, 8)
    assert np.all(full[:, :, cin:] == 0)
    # the kernel's transforms (conv_wino1d.hip): points 0, 1, -1, 2, -2, infinity
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1]], dtype=np.float64)
    d = rng.standard_normal((cin, 6))
    M = np.einsum('ioc,ic->io', u, np.einsum('ij,cj->ic', BT, d))
    y = AT @ M                                     # (2, cout)
    wd = taps.numpy().astype(np.float64)
    want = np.stack([np.einsum('ock,ck->o', wd, d[:, o:o + 5]) for o in range(2)])
    assert np.abs(y - want).max() < 1e-5


def test_wino1d4_packing_reproduces_the_convolution(lib):
    """F(4, 5): the C packer's U through conv_wino1d4.hip's input / output transforms, written out the way the
    kernel evaluates them, gives four outputs of the 5-tap convolution; so does its accumulator seeding of a
    pre-activation term."""
    rng = np.random.default_rng(14)
    cout, cin = 64, 18
    w = torch.from_numpy(rng.standard_normal((cout, cin, 1, 5)).astype(np.float32))
    n = lib.scf_pack_conv_weight_wino1d4_size(cout, cin)
    assert n == ((cin + 3) // 4) * 2 * 1024
    host = torch.empty(n)
    taps = w.reshape(cout, cin, 5).contiguous()
    assert lib.scf_pack_conv_weight_wino1d4(taps.data_ptr(), cout, cin, host.data_ptr()) == 0
    assert torch.equal(host, ops.pack_conv_weight_wino1d4(w))
    assert torch.equal(host, ops.pack_conv_weight_wino1d4(w.reshape(cout, cin, 5, 1)))
    u, full = _unpack(host.numpy().astype(np.float64), 8, cout, cin, 4)
    assert np.all(full[:, :, cin:] == 0)
    d = rng.standard_normal((cin, 8))
    d0, d1, d2, d3, d4, d5, d6, d7 = d.T
    t1, t2 = d2 + d6 - 4.25 * d4, d1 + d5 - 4.25 * d3
    t3, t4 = d6 + 0.25 * d2 - 1.25 * d4, 0.5 * d1 - 2.5 * d3 + 2 * d5
    t5, t6 = d6 + 4 * d2 - 5 * d4, 2 * d1 - 2.5 * d3 + 0.5 * d5
    V = np.stack([(d0 - d6) + 5.25 * (d4 - d2), t1 + t2, t1 - t2, t3 + t4, t3 - t4, t5 + t6, t5 - t6,
                  (d7 - d1) + 5.25 * (d3 - d5)])                      # (8, cin)
    M = np.einsum('ioc,ic->io', u, V)
    res = rng.standard_normal((4, cout))
    M[0] += res[0] - res[2]
    M[1] += 0.5 * (res[1] + res[2])
    M[2] += 0.5 * (res[2] - res[1])
    M[7] += res[3] - res[1]
    s12, d12, s34, d34, s56, d56 = M[1] + M[2], M[1] - M[2], M[3] + M[4], M[3] - M[4], M[5] + M[6], M[5] - M[6]
    y = np.stack([M[0] + s12 + s34 + s56, d12 + 2 * d34 + 0.5 * d56, s12 + 4 * s34 + 0.25 * s56,
                  d12 + 8 * d34 + 0.125 * d56 + M[7]])
    wd = taps.numpy().astype(np.float64)
    want = np.stack([np.einsum('ock,ck->o', wd, d[:, o:o + 5]) for o in range(4)]) + res
    assert np.abs(y - want).max() < 2e-5


def test_wino_packers_reject_bad_arguments(lib):
    buf = torch.empty(16)
    assert lib.scf_pack_conv_weight_wino(None, 4, 4, buf.data_ptr()) != 0
    assert lib.scf_pack_conv_weight_wino1d(buf.data_ptr(), 0, 4, buf.data_ptr()) != 0
    assert lib.scf_pack_conv_weight_wino1d4(buf.data_ptr(), 4, 0, buf.data_ptr()) != 0
    assert lib.scf_pack_conv_weight_wino_size(0, 4) == 0 and lib.scf_pack_conv_weight_wino1d_size(4, 0) == 0
    assert lib.scf_pack_conv_weight_wino1d4_size(0, 4) == 0
