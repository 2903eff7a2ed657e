"""Operator-level Python wrappers over the C ABI (include/scflow_hip.h).

torch is used for what it is good at here -- owning device memory and exposing
the current HIP stream; every arithmetic operation below runs in a
hand-written gfx950 kernel of libscflow_hip.so.  CPU tensors are rejected:
there is no fallback path.

Tensors may be *sample-strided* NCHW views (a channel slice of a larger
contiguous NCHW buffer): only the batch stride is free, which is how channel
concatenations are expressed without copies.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, CONV_GRU_Q, CONV_GRU_ZR,
                   CONV_PLAIN, ConvDesc)

Tensor = torch.Tensor

__all__ = ['record_conv_kernels', 'PackedConv', 'conv_desc', 'gru_passes', 'scflow_iteration', 'side_stream_handle', 'pyramid_layout', 'untile_level', 'level_storage_shape', 'sepconv_gru', 'pack_conv_weight', 'pack_conv_weight_f16x3', 'set_conv_precision', 'set_conv_winograd', 'get_conv_winograd', 'pack_conv_weight_wino', 'pack_conv_weight_wino1d', 'pack_conv_weight_wino1d4',
           'get_conv_precision', 'set_conv_kslices', 'conv_kslices', 'conv_kslices_for', 'constant', 'clear_constants', 'register_conv_workspace', 'choose_kc', 'conv2d', 'conv2d_pair', 'corr_build', 'corr_lookup',
           'instance_norm', 'group_norm_relu', 'linear', 'fc_splitk', 'fc_slices', 'pose_update', 'reproject_flow',
           'unproject_depth', 'linear_pair', 'resize_bilinear', 'convex_upsample', 'avgpool2x2', 'copy_channels',
           'ACT_NONE', 'ACT_RELU', 'ACT_SIGMOID', 'ACT_TANH', 'CONV_PLAIN', 'CONV_GRU_ZR',
           'CONV_GRU_Q']


# The two queries every launch makes -- current device, its current stream -- go straight to torch's C
# layer: torch.cuda.current_stream() builds a Stream object through four Python frames per call (~3 us),
# a fifth of the host cost of an eager batch-1 pass (tools/lab/host_profile.py).
_cur_dev = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


# ---- K-slice workspaces (scf_conv_workspace): opt-in.  With a workspace registered for a stream the library splits small-grid
#      convolutions launched on it into K slices + a combine launch.  Measured on this network (tools/lab/b1_autoslice_ab.py,
#      hipGraph replay): batch 1 3.09 vs 2.88 ms, batch 2 3.65 vs 3.42, batch 4 4.52 vs 4.47, batch 8 equal -- the combine
#      launch (~5.5 us) costs more than the shorter chains save (<= 7 us on the longest one), so nothing registers one by
#      default; ``register_conv_workspace()`` is for callers whose layers have longer chains. ----
_KWS = {}
_KWS_FLOATS = 1 << 20       # the rule slices only launches of <= 256 K-split blocks: N * Cout * Ho * Wo <= 2^18 floats, x 4 slices


def register_conv_workspace(enable: bool = True) -> None:
    """register (or clear) a 4 MB K-slice workspace for the CURRENT stream of the current device.
    The library keys its registry by the raw ``hipStream_t``: CLEAR the registration (``enable=False`` on that stream)
    before destroying a stream -- a new stream that is handed the same handle would otherwise inherit the workspace and
    could share it with another live stream."""
    dev = _cur_dev()
    handle = _raw_stream(dev)
    if enable:
        ws = torch.empty((_KWS_FLOATS,), dtype=torch.float32, device=f'cuda:{dev}')
        _lib.check(_lib.load().scf_conv_workspace(handle, ws.data_ptr(), _KWS_FLOATS), 'scf_conv_workspace')
        _KWS[(dev, handle)] = ws
    elif _KWS.pop((dev, handle), None) is not None:
        _lib.check(_lib.load().scf_conv_workspace(handle, None, 0), 'scf_conv_workspace')


def _stream() -> int:
    return _raw_stream(_cur_dev())


def _dev(t: Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.ScflowHipError(f'{name}: expected a tensor on the GPU (HIP path only, no CPU '
                                  'fallback)')
    if t.dtype != torch.float32:
        raise _lib.ScflowHipError(f'{name}: expected float32, got {t.dtype}')
    # launches go to the CURRENT device's current stream: a tensor on another GPU would be
    # touched through peer access (or fault) from the wrong device's stream
    if t.device.index != _cur_dev():
        raise _lib.ScflowHipError(
            f'{name} lives on cuda:{t.device.index} but the current device is '
            f'cuda:{_cur_dev()}: wrap the call in torch.cuda.device(tensor.device)')


def _dense(t: Tensor, name: str) -> int:
    _dev(t, name)
    if not t.is_contiguous():
        raise _lib.ScflowHipError(f'{name}: expected a contiguous tensor')
    return t.data_ptr()


def _nchw(t: Tensor, name: str) -> Tuple[int, int, int, int, int, int]:
    """(ptr, N, C, H, W, sample_stride) of a sample-strided NCHW tensor."""
    _dev(t, name)
    if t.dim() != 4:
        raise _lib.ScflowHipError(f'{name}: expected 4-D NCHW')
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ok = (sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1)
    if not ok:
        raise _lib.ScflowHipError(f'{name}: not a sample-strided NCHW view (strides {t.stride()})')
    if n == 1:
        sn = c * h * w
    return t.data_ptr(), n, c, h, w, sn


def _opt(t: Optional[Tensor], name: str) -> Optional[int]:
    return None if t is None else _dense(t, name)


# ------------------------------------------------- small-batch stream overlap
# At batch 1 every kernel of the path fills a fraction of the chip (a 32x32 map is 8-32 blocks on 256
# CUs), so independent branches are put on a second HIP stream and run side by side; hipGraph
# capture records the fork / join as graph dependencies.  Large batches keep one stream.
_SIDE = {}      # (device, main stream handle) -> that stream's side stream


# r6 (tools/lab/b1_branches.py: every subset of the four branches as hipGraph replays; tools/lab/graph_fork_penalty.py; tools/lab/
# b1_pairs.py; profiles/r6_b1_branch_subsets.txt, profiles/r6_b1_pairs.txt): on this runtime a replay pays ~1.2 us per NODE as soon
# as the graph holds a real parallel branch (0.33-0.36 ms at 285 nodes, whatever the number of fork / join edges; eager two-stream
# execution pays about the same in event waits), each branch then buys 0.16-0.17 ms back ('upsample': -0.05).  Since r6 the flow and
# mask branches and the context encoder ride in the main branch's launches instead (PAIR_BRANCHES: scf_conv2d_pair, two small-grid
# layers in one launch) and NO branch uses the side stream by default: batch 1 2.71 -> 2.56 ms, batch 2 3.25 -> 3.12, batch 4
# 4.33 -> 4.24 (identical results).  The streams stay available: ops.OVERLAP_BRANCHES = {'context', 'flow', 'mask'[, 'upsample']}, ops.PAIR_BRANCHES = set().
OVERLAP_BRANCHES = set()
OVERLAP_MAX_PIXELS = {'context': 4 * 256 * 256, 'flow': 4 * 256 * 256, 'mask': 4 * 256 * 256, 'upsample': 4 * 256 * 256}


# r6: branches whose convolutions ride in the main branch's launches (scf_conv2d_pair: two small-grid layers, one launch) instead
# of running on the side stream -- the C iteration only (SCFlowDecoder.c_iteration); takes precedence over OVERLAP_BRANCHES
PAIR_BRANCHES = {'context', 'flow', 'mask'}
# ... up to this many pixels per batch (None: any size -- scf_conv2d_pair itself merges a pair only where one launch needs fewer
# rounds of resident blocks than two; the flow branch gains at EVERY batch size: at batch 32 corr_net.1 (768 blocks) and flow_net.1
# (256) are 1.5 + 0.5 rounds apart and 2 full rounds together, 14.00 -> 13.89 ms per step; batch 16: 8.40 -> 8.17).  The two encoders
# walked together lose at batch 32 (they evict each other's activations), the mask branch has no mergeable pair there.
PAIR_MAX_PIXELS = {'context': 4 * 256 * 256, 'flow': None, 'mask': None}


def branch_mode(n: int, h: int, w: int, branch: str) -> int:
    """0 = in order, 1 = side stream, 2 = merged launches, for the C iteration's ``overlap_*`` fields."""
    if branch in PAIR_BRANCHES:
        lim = PAIR_MAX_PIXELS.get(branch, 4 * 256 * 256)
        return 2 if lim is None or n * h * w <= lim else 0
    lim = OVERLAP_MAX_PIXELS.get(branch, 4 * 256 * 256)
    return 1 if n * h * w <= lim and branch in OVERLAP_BRANCHES else 0


def small_work(n: int, h: int, w: int, branch: Optional[str] = None) -> bool:
    """whether a batch of n (h, w) images is small enough for branch-level concurrency to pay
    (``branch``: and that branch is enabled)."""
    lim = OVERLAP_MAX_PIXELS.get(branch, 4 * 256 * 256)
    return n * h * w <= lim and (branch is None or branch in OVERLAP_BRANCHES)


def fork_point() -> 'torch.cuda.Event':
    """record and RETURN an event at this point of the current stream; hand it to
    ``side_stream(after=...)`` to let a side branch start from here instead of from where the
    branch is opened.  The event travels explicitly (no module-level state): an event that is never
    consumed -- an exception between the two calls, a disabled branch -- cannot leak into a later,
    unrelated branch."""
    ev = torch.cuda.Event()
    ev.record()
    return ev


class side_stream:
    """``br = side_stream(enabled, after=ev); with br: <branch>`` enqueues the block on the second
    stream that belongs to (this device, the current stream), ordered after ``ev`` (a
    ``fork_point()`` of the current stream) or, without one, after everything enqueued so far;
    ``br.join()`` -- called once the OTHER branch has been enqueued on the main stream -- makes the
    main stream wait for it.  Tensors the branch writes into are allocated before the fork point;
    tensors it allocates itself must not escape it.  ``enabled=False``: plain in-order execution.
    One side stream per main stream: a user stream and e.g. GraphedRefiner's warm-up stream never
    share one."""

    def __init__(self, enabled: bool = True, after: Optional['torch.cuda.Event'] = None) -> None:
        self.enabled, self.after, self.ctx, self.side = enabled, after, None, None

    def __enter__(self):
        if not self.enabled:
            return self
        main = torch.cuda.current_stream()
        key = (torch.cuda.current_device(), main.cuda_stream)
        self.side = _SIDE.get(key)
        if self.side is None:
            self.side = _SIDE[key] = torch.cuda.Stream(device=key[0])
        ev, self.after = self.after, None
        if ev is not None:
            self.side.wait_event(ev)
        else:
            self.side.wait_stream(main)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled and self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self) -> None:
        if self.enabled and self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


# ------------------------------------------------------------------ conv
def choose_kc(cin: int, kh: int, kw: int, stride: int = 1) -> int:
    """channels staged per LDS round (scf_conv2d accepts 2, 8, 32): 32 for dense 1x1
    (a chunk must carry enough MFMA work to amortise its barriers), 2 for thin inputs or
    wide kernels (LDS / prefetch-register budget), else 8."""
    if cin < 8 or kh * kw >= 25:
        return 2
    if kh * kw == 1 and stride == 1 and cin >= 32:
        return 32
    return 8


def pack_conv_weight(weight: Tensor, kc: int) -> Tuple[Tensor, int]:
    """(Cout, Cin, KH, KW) -> packed (nchunk*T*KC, Mld) with
    row = (chunk*T + tap)*KC + channel_in_chunk, col = cout; zero rows for
    padded channels; Mld = Cout rounded up to 32 (see scf_conv2d)."""
    cout, cin, kh, kw = weight.shape
    t = kh * kw
    nchunk = (cin + kc - 1) // kc
    mld = (cout + 31) // 32 * 32
    w = torch.zeros((cout, nchunk * kc, t), dtype=torch.float32, device=weight.device)
    w[:, :cin] = weight.reshape(cout, cin, t).float()
    w = w.reshape(cout, nchunk, kc, t).permute(1, 3, 2, 0).reshape(nchunk * t * kc, cout)
    out = torch.zeros((nchunk * t * kc, mld), dtype=torch.float32, device=weight.device)
    out[:, :cout] = w
    return out.contiguous(), mld


def pack_conv_weight_f16x3(weight: Tensor) -> Tensor:
    """(Cout, Cin, KH, KW) fp32 -> split-fp16 packing for conv_f16x3.hip:
    w = hi + lo' * 2**-11 with hi = fp16(w), lo' = fp16((w - hi) * 2**11); cells of 8 channels,
    shape ((chunk16*T + tap)*2 + k8, plane hi|lo, Mld, 8) float16."""
    cout, cin, kh, kw = weight.shape
    t = kh * kw
    nchunk = (cin + 31) // 32 * 2       # 16-channel chunks, padded to a multiple of 32 channels
    mld = (cout + 31) // 32 * 32
    w = torch.zeros((cout, nchunk * 16, t), dtype=torch.float32, device=weight.device)
    w[:, :cin] = weight.reshape(cout, cin, t).float()
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    out = torch.zeros((nchunk, t, 2, 2, mld, 8), dtype=torch.float16, device=weight.device)
    for plane, x in enumerate((hi, lo)):
        out[:, :, :, plane, :cout] = x.reshape(cout, nchunk, 2, 8, t).permute(1, 4, 2, 0, 3)
    return out.reshape(nchunk * t * 2, 2, mld, 8).contiguous()


def choose_a4_groups(cin: int, kh: int, kw: int, stride: int) -> int:
    """8G channels per staged chunk of the LDS-DMA kernel (0: layer not eligible)."""
    t = kh * kw
    if stride not in (1, 2) or cin < 8 or t >= 25:
        return 0
    if t == 1:      # dense 1x1: 32-channel chunks (stride 2 = the ResNet shortcuts: staged as a dense 1x1 over
        return 4 if cin >= 64 else 0                        # every second row / column, conv_dma.hip)
    return 2 if t <= 5 else 1


def choose_a4s_groups(cin: int, kh: int, kw: int, stride: int) -> int:
    """8G channels per staged chunk of the SMALL-GRID a4 packing (0: not eligible): at batch 1
    one block runs per CU and the per-chunk fixed cost dominates, so chunks are 2-4x bigger than on
    full grids (and dense 1x1 layers join in)."""
    t = kh * kw
    if stride not in (1, 2) or cin < 16 or t >= 25:
        return 0
    return 4 if t <= 5 else 2


def choose_a4t_groups(cin: int, kh: int, kw: int, stride: int) -> int:
    """TINY-grid a4 packing (no more K-split blocks than CUs): 32-channel chunks for the 3x3 layers,
    whose small-grid packing has 16 (0: the layer has none)."""
    t = kh * kw
    return 4 if (stride in (1, 2) and cin >= 32 and 5 < t < 25) else 0


def pack_conv_weight_a4(weight: Tensor, groups: int) -> Tuple[Tensor, int]:
    """(Cout, Cin, KH, KW) -> [chunk][tap][g][h][Mld][4] (conv_dma.hip): channel
    chunk*8G + 8g + 2s + h at float s of cell (g, h); zero-padded channels and couts."""
    cout, cin, kh, kw = weight.shape
    t, kc = kh * kw, 8 * groups
    nchunk = (cin + kc - 1) // kc
    mld = (cout + 31) // 32 * 32
    w = torch.zeros((mld, nchunk * kc, t), dtype=torch.float32, device=weight.device)
    w[:cout, :cin] = weight.reshape(cout, cin, t).float()
    # channel index -> (chunk, g, s, h)
    w = w.reshape(mld, nchunk, groups, 4, 2, t).permute(1, 5, 2, 4, 0, 3)
    return w.contiguous().reshape(-1), mld


def pack_conv_weight_thin(weight: Tensor) -> Tensor:
    """(Cout <= 4, Cin, KH, KW) -> [Cin][KH*KW][CO] (conv_thin.hip), CO = Cout rounded up to
    1 / 2 / 4, padded to a multiple of 4 floats."""
    cout, cin, kh, kw = weight.shape
    co = 1 if cout <= 1 else 2 if cout <= 2 else 4
    w = torch.zeros((cin, kh * kw, co), dtype=torch.float32, device=weight.device)
    w[:, :, :cout] = weight.reshape(cout, cin, kh * kw).permute(1, 2, 0).float()
    flat = torch.zeros(((w.numel() + 3) // 4 * 4,), dtype=torch.float32, device=weight.device)
    flat[:w.numel()] = w.reshape(-1)
    return flat


def pack_conv_weight_taps(weight: Tensor) -> Tensor:
    """(Cout, Cin <= 4, KH, KW) -> [Kp][Mld] (conv_taps.hip): row k = ci * KH*KW + t, Kp = Cin*KH*KW rounded
    up to a multiple of 8, Mld = Cout rounded up to 32; zero padded."""
    cout, cin, kh, kw = weight.shape
    t = kh * kw
    kp = (cin * t + 7) // 8 * 8
    mld = (cout + 31) // 32 * 32
    out = torch.zeros((kp, mld), dtype=torch.float32, device=weight.device)
    out[:cin * t, :cout] = weight.reshape(cout, cin * t).t().float()
    return out.contiguous()


def pack_conv_weight_wino(weight: Tensor) -> Tensor:
    """(Cout, Cin, 3, 3) -> U = G g G^T in conv_wino.hip's layout, on the weight's device: the library's
    own host packer (``scf_pack_conv_weight_wino``: computed in double, rounded once; layout in
    include/scflow_hip.h) + one copy.  No vendor BLAS on the set-up path."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise ValueError('Winograd packing: 3x3 kernels')
    lib = _lib.load()
    host_w = weight.detach().to('cpu', torch.float32).contiguous()
    out = torch.empty((int(lib.scf_pack_conv_weight_wino_size(cout, cin)),), dtype=torch.float32)
    _lib.check(lib.scf_pack_conv_weight_wino(host_w.data_ptr(), cout, cin, out.data_ptr()), 'scf_pack_conv_weight_wino')
    return out.to(weight.device)


def pack_conv_weight_wino1d(weight: Tensor) -> Tensor:
    """(Cout, Cin, 1, 5) or (Cout, Cin, 5, 1) -> U = G g in conv_wino1d.hip's layout, on the weight's device
    (``scf_pack_conv_weight_wino1d``: G = the 6 x 5 matrix of the points 0, 1, -1, 2, -2, infinity; computed
    in double, rounded once)."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((1, 5), (5, 1)):
        raise ValueError('F(2, 5) packing: 1x5 / 5x1 kernels')
    lib = _lib.load()
    host_w = weight.detach().to('cpu', torch.float32).reshape(cout, cin, 5).contiguous()
    out = torch.empty((int(lib.scf_pack_conv_weight_wino1d_size(cout, cin)),), dtype=torch.float32)
    _lib.check(lib.scf_pack_conv_weight_wino1d(host_w.data_ptr(), cout, cin, out.data_ptr()), 'scf_pack_conv_weight_wino1d')
    return out.to(weight.device)


def pack_conv_weight_wino1d4(weight: Tensor) -> Tensor:
    """(Cout, Cin, 1, 5) or (Cout, Cin, 5, 1) -> U = G g in conv_wino1d4.hip's layout, on the weight's device
    (``scf_pack_conv_weight_wino1d4``: G = the 8 x 5 matrix of the points 0, 1, -1, 2, -2, 1/2, -1/2, infinity;
    computed in double, rounded once)."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) not in ((1, 5), (5, 1)):
        raise ValueError('F(4, 5) packing: 1x5 / 5x1 kernels')
    lib = _lib.load()
    host_w = weight.detach().to('cpu', torch.float32).reshape(cout, cin, 5).contiguous()
    out = torch.empty((int(lib.scf_pack_conv_weight_wino1d4_size(cout, cin)),), dtype=torch.float32)
    _lib.check(lib.scf_pack_conv_weight_wino1d4(host_w.data_ptr(), cout, cin, out.data_ptr()), 'scf_pack_conv_weight_wino1d4')
    return out.to(weight.device)


_CONV_PRECISION = 'f32'
_CONV_WINOGRAD = True


def set_conv_winograd(on: bool) -> bool:
    """3x3 / stride-1 / pad-1 layers with plain or affine epilogues through the Winograd F(2x2, 3x3)
    kernel (conv_wino.hip: fp32 throughout, 2.25x fewer matrix-core flops, sums re-associated: error vs
    fp64 1.5-1.8x the direct kernels' ~1e-6, end-to-end flow EPE unchanged at 6e-5 px) instead of the direct
    kernels, on grids of >= CUs / 2 blocks (small grids stay direct); the 1x5 / 5x1 GRU gates likewise through
    the F(2, 5) kernel (conv_wino1d.hip).  On by default.  Under conv precision 'f16x3' the layers that carry
    a split-fp16 packing take that kernel instead; layers without one (Cin < 16) still take Winograd.
    ``False`` = the direct kernels everywhere (bit-for-bit fp32 fma chains in the reference's summation
    order).  NB the kernel choice depends on the grid size (batch x map size) and the device's CU count:
    the same pair can differ by ~1e-6 between batch 1 and batch 32; ``set_conv_winograd(False)`` removes
    the largest part of that dependence (the direct kernels still pick K-split tiles on small grids).
    Returns the previous setting."""
    global _CONV_WINOGRAD
    prev, _CONV_WINOGRAD = _CONV_WINOGRAD, bool(on)
    return prev


def get_conv_winograd() -> bool:
    return _CONV_WINOGRAD


def set_conv_precision(mode: str) -> str:
    """'f32'   : v_mfma_f32_32x32x2_f32 everywhere (bit-for-bit fp32 fma chains);
    'f16x3' : spatial convolutions with >= 16 input channels run on the fp16 matrix cores as
              three MFMAs over an exact hi/lo split of both operands (fp32 accumulate, ~22
              mantissa bits; flow EPE ~1e-4 px vs fp32 on this path), everything else fp32.
    Returns the previous mode."""
    global _CONV_PRECISION
    if mode not in ('f32', 'f16x3'):
        raise ValueError(mode)
    prev, _CONV_PRECISION = _CONV_PRECISION, mode
    return prev


def get_conv_precision() -> str:
    return _CONV_PRECISION


@dataclass
class PackedConv:
    """a convolution's parameters in kernel layout (device resident)."""
    wp: Tensor
    bias: Optional[Tensor]
    scale: Optional[Tensor]
    shift: Optional[Tensor]
    cin: int
    cout: int
    kh: int
    kw: int
    stride: int
    pad_h: int
    pad_w: int
    kc: int
    mld: int
    wp_alt: Optional[Tensor] = None   # KC=32 packing of the same weights (short-chunk regime)
    plans: Optional[dict] = None      # (N, H, W, C0, C1) -> use the KC=32 packing?
    wp16: Optional[Tensor] = None     # split-fp16 packing (spatial kernels, Cin >= 16)
    wp4: Optional[Tensor] = None      # LDS-DMA packing (stride 1, Cin >= 8)
    g4: int = 0
    wthin: Optional[Tensor] = None    # [Cin][T][CO] packing (Cout <= 4)
    wp4s: Optional[Tensor] = None     # small-grid LDS-DMA packing (bigger chunks)
    g4s: int = 0
    wtaps: Optional[Tensor] = None    # [Cin*T][Mld] packing (Cin <= 4: contraction over taps)
    desc: Optional[ConvDesc] = None   # scf_conv_desc with this layer's own fields filled (built lazily)
    wp4t: Optional[Tensor] = None     # tiny-grid LDS-DMA packing of 3x3 layers (32-channel chunks)
    g4t: int = 0
    wwino: Optional[Tensor] = None    # G g G^T packing (3x3, stride 1, pad 1, Cin >= 8)
    wwino1d: Optional[Tensor] = None  # G g packing (1x5 / 5x1, stride 1, 'same', Cin >= 16, Cout % 64 == 0)
    wwino1d4: Optional[Tensor] = None  # the F(4, 5) packing of the same layers

    @staticmethod
    def from_weight(weight: Tensor, bias: Optional[Tensor], stride: int = 1,
                    padding=0, bn: Optional[Sequence[Tensor]] = None,
                    eps: float = 1e-5, dma_packing: bool = True) -> 'PackedConv':
        """``bn`` = (gamma, beta, running_mean, running_var): eval-mode BatchNorm
        folded into a per-channel scale/shift applied after the bias.  ``dma_packing=False``
        leaves out the LDS-DMA kernel's packing (the layer then runs on the register-staged
        kernel; used by the A/B parity tests)."""
        cout, cin, kh, kw = weight.shape
        kc = choose_kc(cin, kh, kw, stride)
        wp, mld = pack_conv_weight(weight, kc)
        wp_alt = pack_conv_weight(weight, 32)[0] if (kc == 8 and cin >= 32) else None
        wp16 = pack_conv_weight_f16x3(weight) if (cin >= 16 and kh * kw > 1) else None
        ph, pw = (padding, padding) if isinstance(padding, int) else padding
        scale = shift = None
        if bn is not None:
            gamma, beta, mean, var = [b.float() for b in bn]
            scale = (gamma / torch.sqrt(var + eps)).contiguous()
            shift = (beta - mean * scale).contiguous()
        g4 = choose_a4_groups(cin, kh, kw, stride) if dma_packing else 0
        wp4 = pack_conv_weight_a4(weight, g4)[0] if g4 else None
        g4s = choose_a4s_groups(cin, kh, kw, stride) if dma_packing else 0
        wp4s = pack_conv_weight_a4(weight, g4s)[0] if g4s else None
        return PackedConv(wp, None if bias is None else bias.float().contiguous(), scale, shift,
                          cin, cout, kh, kw, stride, ph, pw, kc, mld, wp_alt, {}, wp16, wp4, g4,
                          pack_conv_weight_thin(weight) if (cout <= 4 and stride == 1 and cin >= 32) else None,
                          wp4s, g4s, pack_conv_weight_taps(weight) if (cin <= 4 and dma_packing) else None,
                          None, *PackedConv._tiny(weight, kh, kw, stride, dma_packing),
                          pack_conv_weight_wino(weight) if (dma_packing and (kh, kw, stride, ph, pw) == (3, 3, 1, 1, 1)
                                                            and cin >= 8) else None,
                          pack_conv_weight_wino1d(weight) if (dma_packing and stride == 1 and cin >= 16 and cout % 64 == 0
                                                              and (kh, kw, ph, pw) in ((1, 5, 0, 2), (5, 1, 2, 0))) else None,
                          pack_conv_weight_wino1d4(weight) if (dma_packing and stride == 1 and cin >= 16 and cout % 64 == 0
                                                               and (kh, kw, ph, pw) in ((1, 5, 0, 2), (5, 1, 2, 0))) else None)

    @staticmethod
    def _tiny(weight, kh, kw, stride, dma_packing):
        g = choose_a4t_groups(weight.shape[1], kh, kw, stride) if dma_packing else 0
        return (pack_conv_weight_a4(weight, g)[0] if g else None), g

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        return ((h + 2 * self.pad_h - self.kh) // self.stride + 1,
                (w + 2 * self.pad_w - self.kw) // self.stride + 1)


def _desc_template(pc: 'PackedConv') -> ConvDesc:
    d = ConvDesc()
    d.wp, d.w_nstride, d.Mld, d.Cout = pc.wp.data_ptr(), 0, pc.mld, pc.cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.KC = pc.kh, pc.kw, pc.stride, pc.pad_h, pc.pad_w, pc.kc
    d.bias = None if pc.bias is None else pc.bias.data_ptr()
    d.scale = None if pc.scale is None else pc.scale.data_ptr()
    d.shift = None if pc.shift is None else pc.shift.data_ptr()
    d.out_div = 1.0
    if pc.wthin is not None:
        d.wp_thin = pc.wthin.data_ptr()
    if pc.wtaps is not None:
        d.wp_taps = pc.wtaps.data_ptr()
    if pc.wp4 is not None:
        d.wp_a4, d.a4_groups, d.a4_mld = pc.wp4.data_ptr(), pc.g4, pc.mld
    if pc.wp4s is not None:
        d.wp_a4s, d.a4s_groups, d.a4_mld = pc.wp4s.data_ptr(), pc.g4s, pc.mld
    if pc.wp4t is not None:
        d.wp_a4t, d.a4t_groups, d.a4_mld = pc.wp4t.data_ptr(), pc.g4t, pc.mld
    return d


def conv_desc(pc: PackedConv, x0: Tensor, x1: Optional[Tensor] = None, out: Optional[Tensor] = None,
              **kw) -> Tuple[ConvDesc, Tensor]:
    """the filled ``scf_conv_desc`` of ``conv2d(pc, x0, x1, out, **kw)`` WITHOUT launching it, and the
    output tensor: for entry points that take whole descriptors (``scf_scflow_iteration``)."""
    return conv2d(pc, x0, x1, out, _launch=False, **kw)


def conv2d(pc: PackedConv, x0: Tensor, x1: Optional[Tensor] = None, out: Optional[Tensor] = None,
           *, res: Optional[Tensor] = None, act: int = ACT_NONE, act2: int = ACT_NONE,
           act_split: int = 0, mode: int = CONV_PLAIN, gru_h: Optional[Tensor] = None,
           gru_aux: Optional[Tensor] = None, gru_z: Optional[Tensor] = None, kslices: int = 1,
           _launch: bool = True):
    """implicit-GEMM MFMA convolution with fused epilogue (scf_conv2d).
    input = channel concat of x0 and x1; ``out`` may be a channel slice.
    ``kslices`` = S > 1: the contraction over the input channels is split across S groups of blocks
    (``scf_conv_desc.k_slices``); returns the (S, N, Cout, Ho, Wo) RAW partial sums -- the consumer adds them in slice
    order (``group_norm_relu`` / ``fc_splitk`` take such a tensor).  Plain epilogue only (the layer has no bias)."""
    p0, n, c0, h, w, s0 = _nchw(x0, 'x0')
    p1, c1, s1 = None, 0, 0
    if x1 is not None:
        p1, n1, c1, h1, w1, s1 = _nchw(x1, 'x1')
        if (n1, h1, w1) != (n, h, w):
            raise _lib.ScflowHipError('x0/x1 shape mismatch')
    if c0 + c1 != pc.cin:
        raise _lib.ScflowHipError(f'conv expects {pc.cin} input channels, got {c0}+{c1}')
    ho, wo = pc.out_hw(h, w)
    sliced = None
    if kslices > 1:
        if pc.bias is not None or pc.scale is not None or res is not None or act != ACT_NONE or mode != CONV_PLAIN:
            raise _lib.ScflowHipError('conv2d(kslices > 1): partial sums take a plain epilogue (no bias / BN / res / act)')
        sliced = out if out is not None else torch.empty((kslices, n, pc.cout, ho, wo), dtype=torch.float32, device=x0.device)
        if tuple(sliced.shape) != (kslices, n, pc.cout, ho, wo) or not sliced.is_contiguous():
            raise _lib.ScflowHipError(f'conv2d(kslices={kslices}): out must be a contiguous {(kslices, n, pc.cout, ho, wo)} tensor')
        out = sliced[0]
    if out is None:
        out = torch.empty((n, pc.cout // 2 if mode == CONV_GRU_ZR else pc.cout, ho, wo),
                          dtype=torch.float32, device=x0.device)
    po, no, co, oh, ow, so = _nchw(out, 'out')
    want_c = pc.cout // 2 if mode == CONV_GRU_ZR else pc.cout   # ZR: only the z half lands in out
    if (no, co, oh, ow) != (n, want_c, ho, wo):
        raise _lib.ScflowHipError(f'out has shape {tuple(out.shape)}, expected '
                                  f'{(n, want_c, ho, wo)}')
    # the layer's own fields (weights in every packing, geometry, bias / BN) are filled once per
    # PackedConv and copied; a call sets the tensors and the epilogue
    tmpl = pc.desc
    if tmpl is None:
        tmpl = pc.desc = _desc_template(pc)
    d = ConvDesc.from_buffer_copy(tmpl)
    d.in0, d.in1, d.C0, d.C1 = p0, p1, c0, c1
    d.in0_nstride, d.in1_nstride = s0, s1
    d.N, d.H, d.W = n, h, w
    d.out, d.out_nstride = po, so
    if sliced is not None:
        d.k_slices, d.out_slice_stride = kslices, sliced.stride(0)
        out = sliced
    if res is not None:
        pr, nr, cr, hr, wr, sr = _nchw(res, 'res')
        if (nr, cr, hr, wr) != (n, pc.cout, ho, wo):
            raise _lib.ScflowHipError('res shape mismatch')
        d.res, d.res_nstride = pr, sr
    d.act, d.act2, d.act_split, d.mode = act, act2, act_split, mode
    if gru_h is not None:
        ph_, _, _, _, _, sh_ = _nchw(gru_h, 'gru_h')
        d.gru_h, d.gru_h_nstride = ph_, sh_
    if gru_aux is not None:
        pa_, _, _, _, _, sa_ = _nchw(gru_aux, 'gru_aux')
        d.gru_aux, d.gru_aux_nstride = pa_, sa_
    if gru_z is not None:
        pz_, _, _, _, _, sz_ = _nchw(gru_z, 'gru_z')
        d.gru_z, d.gru_z_nstride = pz_, sz_
    # Short-chunk regime (smallest tile, WM = WN = 1): with KC=8 the MFMA phase of a chunk
    # (T*4*WM*WN MFMAs of 64 cycles) is shorter than the L2 round trip its prefetch has to
    # hide.  Stage 32 channels per chunk instead when that packing fits (decided once per shape).
    if _CONV_PRECISION == 'f16x3' and pc.wp16 is not None:
        d.wp_f16 = pc.wp16.data_ptr()
    elif _CONV_WINOGRAD and pc.wwino is not None and mode == CONV_PLAIN:
        d.wp_wino = pc.wwino.data_ptr()
    elif _CONV_WINOGRAD and pc.wwino1d is not None:
        d.wp_wino1d = pc.wwino1d.data_ptr()
        d.wp_wino1d4 = pc.wwino1d4.data_ptr() if pc.wwino1d4 is not None else None
    if x1 is not None:
        d.wp_taps = None                # the thin-input kernel takes one input segment
    if pc.wp_alt is not None and (c1 == 0 or c0 % 32 == 0):
        key = (n, h, w, c0, c1, d.wp_f16 is not None)
        use_alt = pc.plans.get(key)
        if use_alt is None:
            lib = _lib.load()
            info = (C.c_int32 * 4)()
            use_alt = False
            # the KC decision describes the DIRECT register-staged kernel: asked with the Winograd packings
            # cleared, so that the cached plan does not depend on whether Winograd was on at the first call
            ww, ww1, ww4, d.wp_wino, d.wp_wino1d, d.wp_wino1d4 = d.wp_wino, d.wp_wino1d, d.wp_wino1d4, None, None, None
            if (lib.scf_conv2d_query(C.byref(d), info) == 0 and info[0] * info[1] == 1
                    and 0 <= info[3] * 64 < 6000):
                d.wp, d.KC = pc.wp_alt.data_ptr(), 32
                use_alt = lib.scf_conv2d_query(C.byref(d), info) == 0
                d.wp, d.KC = pc.wp.data_ptr(), pc.kc
            d.wp_wino, d.wp_wino1d, d.wp_wino1d4 = ww, ww1, ww4
            pc.plans[key] = use_alt
        if use_alt:
            d.wp, d.KC = pc.wp_alt.data_ptr(), 32
    if not _launch:
        return d, out
    if _CONV_EVENTS is not None:        # bench.py: a timer bound to this launch + algorithmic flops
        lib = _lib.load()
        tm = C.c_void_p()
        _lib.check(lib.scf_timer_create(C.byref(tm)), 'scf_timer_create')
        lib.scf_timer_arm(tm)
        try:
            _lib.check(lib.scf_conv2d(C.byref(d), _stream()), 'scf_conv2d')
        finally:
            lib.scf_timer_arm(None)
        tag = ''
        if d.wp_wino or d.wp_wino1d:    # does a Winograd kernel take this launch (grid-size policy)?
            info = (C.c_int32 * 4)()
            if lib.scf_conv2d_query(C.byref(d), info) == 0 and info[3] < 0 and info[0] in (16, 6, 8):
                tag = {16: ' [winograd]', 6: ' [winograd F(2,5)]', 8: ' [winograd F(4,5)]'}[info[0]]
        _CONV_EVENTS.append((tm, 2.0 * pc.cin * pc.kh * pc.kw * pc.cout * ho * wo * n,
                             f'{pc.cin}->{pc.cout} {pc.kh}x{pc.kw}/s{pc.stride} @{ho}x{wo} N{n}' + tag))
        return out
    rc = _lib.load().scf_conv2d(C.byref(d), _stream())
    if rc == -2 and x1 is not None and sliced is None:
        # SCF_EUNSUPPORTED with two input segments: every kernel needs the segment boundary on a channel-chunk boundary
        # (C0 % 8 / 16 / 32 == 0 by packing) -- a layer of the path never violates it, an arbitrary caller may (found by
        # tools/lab/conv_fuzz.py: 1x1 layers with C0 = 8 / 16).  Materialise the concatenation once and run the same
        # layer on one segment: still the HIP kernels, one extra copy.
        return conv2d(pc, torch.cat([x0, x1], 1), None, out, res=res, act=act, act2=act2, act_split=act_split, mode=mode,
                      gru_h=gru_h, gru_aux=gru_aux, gru_z=gru_z)
    _lib.check(rc, 'scf_conv2d')
    return out


def conv2d_pair(a, b):
    """two INDEPENDENT convolutions, ``a`` and ``b`` = ``(PackedConv, x0[, kwargs of conv2d])``, through ``scf_conv2d_pair``: ONE
    launch where both fall to the same small-grid kernel instantiation and fit the chip together, else one after the other.
    Returns the two outputs (bit-identical to two ``conv2d`` calls either way)."""
    da, oa = conv2d(a[0], a[1], _launch=False, **(a[2] if len(a) > 2 else {}))
    db, ob = conv2d(b[0], b[1], _launch=False, **(b[2] if len(b) > 2 else {}))
    _lib.check(_lib.load().scf_conv2d_pair(C.byref(da), C.byref(db), _stream()), 'scf_conv2d_pair')
    return oa, ob


_CONV_EVENTS = None

# ---- read-only constant tensors (the all-zero initial flow, the all-ones first-iteration mask): filled ONCE per
#      (device, shape, value) and handed out again -- a steady-state step then launches no fill kernel.  Kernels only
#      READ them (init_flow feeds the first 1/8 down-sampling, the ones map the first mask multiply); callers that
#      want to write must not ask here.
#      Lifetime rules: entries are NEVER evicted -- a captured hipGraph holds only the raw pointer of the
#      tensor it read, so dropping one would hand its memory to the allocator while replays still read it
#      (``clear_constants`` is explicit and for processes that hold no graph); a constant is never CREATED while a
#      stream is capturing (its fill would be a node of that graph's private pool and eager callers would be handed an
#      unfilled tensor from the cache): create it in a warm-up pass; and the creating stream is synchronised once after
#      the fill, so a consumer on any other stream (side branches, a later capture stream) reads a finished tensor. ----
_CONSTANTS = {}


def constant(shape, value: float, device) -> Tensor:
    key = (str(torch.device(device)), tuple(int(d) for d in shape), float(value))
    t = _CONSTANTS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.ScflowHipError(
                f'ops.constant{key[1]}: first use inside a stream capture -- run one eager warm-up pass of the same '
                f'shapes before capturing (scflow_amd.GraphedRefiner does, warmup >= 1)')
        t = torch.full(key[1], float(value), dtype=torch.float32, device=device)
        torch.cuda.current_stream(t.device).synchronize()
        _CONSTANTS[key] = t
    return t


def clear_constants() -> None:
    """drop the cached constant tensors.  ONLY when no captured graph that read one of them will be replayed again."""
    _CONSTANTS.clear()


# ---- K split across blocks for bias-free conv + GroupNorm blocks on small grids (the pose head's stride-2 layers) ----
_CONV_KSLICES = True


def set_conv_kslices(on: bool) -> bool:
    """False: ``conv_kslices`` answers 1 (every convolution contracts its whole K inside one block: the r4 arithmetic
    order); returns the previous setting.  Independent of the library's own automatic slicing of small grids
    (``tune('conv_autoslice', ..)``, which needs a registered workspace and is off unless a caller registers one)."""
    global _CONV_KSLICES
    prev, _CONV_KSLICES = _CONV_KSLICES, bool(on)
    return prev


def conv_kslices(pc: 'PackedConv', n: int, h: int, w: int) -> int:
    """WANTED slice count for ``conv2d(pc, x, kslices=...)`` of a bias-free layer whose consumer adds partial tensors
    (``conv_kslices_for`` = this, checked against the library for the actual tensors).
    A launch with few blocks is a serial chain of one memory round trip per staged channel chunk; S slices are S times
    the blocks and 1 / S of that chain.  Rule from ``tools/lab/kslice_sweep.py`` on the MI355X (kb = blocks of the
    32-channel x 32-pixel tile): kb < 128 -> 4 (11.8 -> 6.1 us, 18.0 -> 8.3), 128 <= kb < 512 -> 2 (4 can tip the
    launch onto the pixel-split tile: 11.9 -> 16.4 us), kb >= 512 -> 4 (64.2 -> 44.6, 31.8 -> 28.4); never more
    slices than 32-channel chunks.  NB the count depends on the batch size, so a sample's bits may differ between
    batch sizes (the partial sums re-associate); ``set_conv_kslices(False)`` is the batch-invariant setting."""
    if not _CONV_KSLICES or pc.bias is not None or pc.scale is not None or (pc.wp4 is None and pc.wp4s is None):
        return 1
    if _CONV_PRECISION == 'f16x3' and pc.wp16 is not None:
        return 1
    ho, wo = pc.out_hw(h, w)
    kb = n * ((ho * wo + 31) // 32) * ((pc.cout + 31) // 32)
    s = 2 if 128 <= kb < 512 else 4
    return max(1, min(s, pc.cin // 32))


def conv_kslices_for(pc: 'PackedConv', x0: Tensor, x1: Optional[Tensor] = None) -> int:
    """``conv_kslices`` validated by the library: an explicit ``k_slices`` launch only exists on the LDS-DMA
    dispatch, which refuses some layers outright (other strides than 1 / 2, a channel concat that splits a chunk, an
    ``S`` that leaves a slice without chunks, ...) -- the same layer unsliced would simply have run on another kernel.
    ``scf_conv2d_query`` is asked with the real descriptor, S halves until it says yes (1 = unsliced always works);
    the answer is cached per (shape, channel split, S)."""
    n, _, h, w = x0.shape
    s = conv_kslices(pc, n, h, w)
    if s <= 1:
        return 1
    c1 = 0 if x1 is None else x1.shape[1]
    key = ('ks', n, h, w, x0.shape[1], c1, s, _CONV_PRECISION, _CONV_WINOGRAD)
    if pc.plans is None:
        pc.plans = {}
    ok = pc.plans.get(key)
    if ok is None:
        lib = _lib.load()
        info = (C.c_int32 * 4)()
        ok = s
        while ok > 1:
            d, _ = conv2d(pc, x0, x1, kslices=ok, _launch=False)
            if lib.scf_conv2d_query(C.byref(d), info) == 0:
                break
            ok //= 2
        pc.plans[key] = ok = max(ok, 1)
    return ok


def _gru_passes(packs):
    arr = (_lib.GruPass * len(packs))()
    f16 = _CONV_PRECISION == 'f16x3'
    for g, (pzr, pq) in zip(arr, packs):
        g.KH, g.KW, g.pad_h, g.pad_w = pzr.kh, pzr.kw, pzr.pad_h, pzr.pad_w
        g.wp_zr, g.bias_zr = pzr.wp.data_ptr(), (None if pzr.bias is None else pzr.bias.data_ptr())
        g.wp_q, g.bias_q = pq.wp.data_ptr(), (None if pq.bias is None else pq.bias.data_ptr())
        if pzr.wp4 is not None and pq.wp4 is not None and pzr.g4 == pq.g4:
            g.wp_zr_a4, g.wp_q_a4, g.a4_groups = pzr.wp4.data_ptr(), pq.wp4.data_ptr(), pzr.g4
        if f16 and pzr.wp16 is not None and pq.wp16 is not None:
            g.wp_zr_f16, g.wp_q_f16 = pzr.wp16.data_ptr(), pq.wp16.data_ptr()
        if pzr.wp_alt is not None and pq.wp_alt is not None:
            g.wp_zr_k32, g.wp_q_k32 = pzr.wp_alt.data_ptr(), pq.wp_alt.data_ptr()
        if pzr.wp4s is not None and pq.wp4s is not None and pzr.g4s == pq.g4s:
            g.wp_zr_a4s, g.wp_q_a4s, g.a4s_groups = pzr.wp4s.data_ptr(), pq.wp4s.data_ptr(), pzr.g4s
        if pzr.wp4t is not None and pq.wp4t is not None and pzr.g4t == pq.g4t:
            g.wp_zr_a4t, g.wp_q_a4t, g.a4t_groups = pzr.wp4t.data_ptr(), pq.wp4t.data_ptr(), pzr.g4t
        if _CONV_WINOGRAD and not f16 and pzr.wwino1d is not None and pq.wwino1d is not None:
            g.wp_zr_wino1d, g.wp_q_wino1d = pzr.wwino1d.data_ptr(), pq.wwino1d.data_ptr()
            if pzr.wwino1d4 is not None and pq.wwino1d4 is not None:
                g.wp_zr_wino1d4, g.wp_q_wino1d4 = pzr.wwino1d4.data_ptr(), pq.wwino1d4.data_ptr()
    return arr


gru_passes = _gru_passes


def side_stream_handle() -> int:
    """raw handle of the side stream that belongs to (current device, current stream) -- the stream
    ``side_stream`` blocks run on; created on first use."""
    main = torch.cuda.current_stream()
    key = (torch.cuda.current_device(), main.cuda_stream)
    side = _SIDE.get(key)
    if side is None:
        side = _SIDE[key] = torch.cuda.Stream(device=key[0])
    return side.cuda_stream


def scflow_iteration(it: '_lib.ScflowIter') -> None:
    """one refinement iteration through ``scf_scflow_iteration`` (the struct is filled by
    ``SCFlowDecoder``).  With lookup timing enabled (bench.py) the lookup launch carries a timer."""
    lib = _lib.load()
    it.lookup_timer = _LOOKUP_TIMERS.take() if _LOOKUP_TIMERS is not None else None
    _lib.check(lib.scf_scflow_iteration(C.byref(it), _stream()), 'scf_scflow_iteration')


def sepconv_gru(packs, hx: Tensor, h_channels: int, z: Tensor, rh: Tensor,
                ctx: Optional[Sequence[Tensor]] = None, ctx_channels: int = 0) -> None:
    """ConvGRU.forward (raft_decoder.py:235-253) on hx = [h | x], h updated in place:
    ``scf_sepconv_gru``.  ``packs``: [(PackedConv of cat(conv_z, conv_r), PackedConv of conv_q)]
    per pass.  With ``ctx`` (one (N, 3 h_channels, H, W) tensor per pass) hx = [h | c | x'] and
    the packs cover [h | x'] only: ``scf_sepconv_gru_ctx`` (the c part of the convolutions, computed
    once per pair, enters before the gates' activations).  With convolution timers armed (bench.py)
    the same launches are issued one by one through ``conv2d`` so that each carries its own timer."""
    hc = h_channels
    if ctx is not None and (ctx_channels <= 0 or len(ctx) != len(packs)):
        raise _lib.ScflowHipError('sepconv_gru: one context term per pass, ctx_channels > 0')
    if _CONV_EVENTS is not None:
        hv, xv = hx[:, :hc], hx[:, hc + ctx_channels:]
        for i, (pzr, pq) in enumerate(packs):
            if ctx is None:
                conv2d(pzr, hx, out=z, mode=CONV_GRU_ZR, gru_h=hv, gru_aux=rh)
                conv2d(pq, rh, xv, out=hv, mode=CONV_GRU_Q, gru_h=hv, gru_z=z)
            else:
                conv2d(pzr, hv, xv, out=z, mode=CONV_GRU_ZR, gru_h=hv, gru_aux=rh, res=ctx[i][:, :2 * hc])
                conv2d(pq, rh, xv, out=hv, mode=CONV_GRU_Q, gru_h=hv, gru_z=z, res=ctx[i][:, 2 * hc:])
        return
    p, n, c, h, w, sn = _nchw(hx, 'hx')
    arr = _gru_passes(packs)
    if ctx is None:
        _lib.check(_lib.load().scf_sepconv_gru(p, sn, n, hc, c - hc, h, w, arr, len(packs),
                                               _dense(z, 'z'), _dense(rh, 'rh'), _stream()),
                   'scf_sepconv_gru')
        return
    ptrs, cs = (C.c_void_p * len(ctx))(), None
    for i, t in enumerate(ctx):
        pc_, nc, cc_, hh, ww, s_ = _nchw(t, 'ctx')
        if (nc, cc_, hh, ww) != (n, 3 * hc, h, w) or (cs is not None and s_ != cs):
            raise _lib.ScflowHipError('sepconv_gru: ctx must be (N, 3*h_channels, H, W), equal strides')
        ptrs[i], cs = pc_, s_
    _lib.check(_lib.load().scf_sepconv_gru_ctx(p, sn, n, hc, ctx_channels, c - hc - ctx_channels, h, w,
                                               arr, len(packs), ptrs, cs, _dense(z, 'z'),
                                               _dense(rh, 'rh'), _stream()),
               'scf_sepconv_gru_ctx')


def _read_timers(timers):
    lib = _lib.load()
    out = []
    for tm in timers:
        us = C.c_float()
        _lib.check(lib.scf_timer_elapsed_us(tm, C.byref(us)), 'scf_timer_elapsed_us')
        lib.scf_timer_destroy(tm)
        out.append(float(us.value))
    return out


def conv_timing(enable: bool):
    """like ``lookup_timing`` for the convolution launches: enable=False returns a list of
    (microseconds, algorithmic flops = 2*Cin*KH*KW*Cout*Ho*Wo*N, shape tag) per launch; the tag ends in
    ' [winograd]' when the F(2x2, 3x3) kernel ran the launch (it executes 1 / 2.25 of those flops), in
    ' [winograd F(2,5)]' / ' [winograd F(4,5)]' for the 1x5 / 5x1 kernels (1 / 1.667, 1 / 2.5)."""
    global _CONV_EVENTS
    if enable:
        _CONV_EVENTS = []
        return None
    evs, _CONV_EVENTS = _CONV_EVENTS or [], None
    torch.cuda.synchronize()
    return list(zip(_read_timers([e[0] for e in evs]), [e[1] for e in evs], [e[2] for e in evs]))


TUNE_KEYS = {'wino_variant': 1, 'dma_force_ksplit': 2, 'dma_ksplit_groups': 3, 'wino1d4': 4, 'lookup_pipe': 5, 'lookup_store': 6, 'conv_autoslice': 7, 'iter_merge': 8, 'wino1d4_half': 9, 'conv_pair': 10}


def tune(key: str, value: int) -> int:
    """measurement knob of the library (``scf_tune``, scflow_hip_prof.h): returns the previous value.
    ``'wino_variant'``: 0 = the dispatch's choice, 1 = pair kernel, 2 = quarter-domain kernel (3 / 4 / 5 exist in lab builds
    of the library only; the product library refuses them);
    ``'wino1d4'``: 1 = F(4, 5) where the dispatch prefers it (default), 0 = never, 2 = on every grid it supports;
    ``'lookup_pipe'``: 0 = the dispatch's choice, 1 = one group per block (r3 kernel), 2 / 3 = pipelined kernel with
    that many groups per block, 4 / 5 / 6 = two / four / three groups per block (same waves, fewer workgroups).  Unknown keys raise ``ValueError``; a value the library refuses raises
    ``RuntimeError`` (the return value is always a previous setting, never an error code)."""
    if key not in TUNE_KEYS:
        raise ValueError(f'unknown tune key {key!r}; known: {sorted(TUNE_KEYS)}')
    prev = int(_lib.load().scf_tune(TUNE_KEYS[key], int(value)))
    if prev < 0:
        _lib.check(prev, f'scf_tune({key}, {value})')
    return prev


class record_conv_kernels:
    """``with ops.record_conv_kernels() as ran: ...`` -- afterwards ``ran`` is a list of
    ``(layer tag, kernel family)`` for every convolution launch the library made inside the block, in
    launch order, INCLUDING those issued inside ``scf_sepconv_gru*`` / ``scf_scflow_iteration``
    (``scf_conv_log_*`` of scflow_hip_prof.h).  Tag = ``'<Cin>-><Cout> <KH>x<KW>/s<stride> @<Ho>x<Wo> N<N>'``
    (the tag ``conv_timing`` uses), family = one of ``_lib.KERNEL_NAMES``' values.  Parity tests assert with
    it that the kernel they name really ran (the choice depends on grid size and device)."""

    _active = False      # the library keeps ONE process-wide log: a nested recorder would clear the outer one's records

    def __init__(self, capacity: int = 1 << 16) -> None:
        self.capacity, self.ran = capacity, []

    def __enter__(self):
        if record_conv_kernels._active:
            raise RuntimeError('record_conv_kernels does not nest (one process-wide dispatch log)')
        _lib.check(_lib.load().scf_conv_log_enable(self.capacity), 'scf_conv_log_enable')
        record_conv_kernels._active = True
        return self.ran

    def __exit__(self, *exc):
        record_conv_kernels._active = False
        lib = _lib.load()
        n = lib.scf_conv_log_read(None, 0)
        buf = (_lib.ConvLogEntry * max(n, 1))()
        n = min(lib.scf_conv_log_read(buf, n), n)
        lib.scf_conv_log_enable(0)
        for e in buf[:n]:
            self.ran.append((f'{e.Cin}->{e.Cout} {e.KH}x{e.KW}/s{e.stride} @{e.Ho}x{e.Wo} N{e.N}',
                             _lib.KERNEL_NAMES.get(e.kernel, str(e.kernel))))
        return False


def time_first_kernel(fn) -> float:
    """run ``fn()`` with a launch-bound timer armed: microseconds of the FIRST kernel it launches
    through the library (e.g. the contraction of ``corr_build``, not its pooling cascade)."""
    lib = _lib.load()
    tm = C.c_void_p()
    _lib.check(lib.scf_timer_create(C.byref(tm)), 'scf_timer_create')
    lib.scf_timer_arm(tm)
    try:
        fn()
    finally:
        lib.scf_timer_arm(None)
    torch.cuda.synchronize()
    return _read_timers([tm])[0]


# ----------------------------------------------------- correlation volume
def pyramid_layout(h: int, w: int, radius: int = 4, num_levels: int = 4) -> int:
    """the pyramid layout the lookup kernel is fastest with: bit l set = level l stored in 8x4-float
    tiles of 128 B (``scf_corr_preferred_layout``: every level whose rows are >= 24 floats and that
    does not fit the lookup window whole).  0 = the reference's row-major pyramid."""
    return int(_lib.load().scf_corr_preferred_layout(h, w, radius, num_levels))


def level_storage_shape(h: int, w: int, level: int, tiled: bool) -> Tuple[int, int]:
    """(rows, cols) a query's level-``level`` map occupies: the map itself, or -- tiled -- the map
    padded to 4 rows x 8 columns."""
    lh, lw = h >> level, w >> level
    return ((lh + 3) // 4 * 4, (lw + 7) // 8 * 8) if tiled else (lh, lw)


def untile_level(level: Tensor, lh: int, lw: int) -> Tensor:
    """torch view-shuffle of a tiled pyramid level (q, 1, PH, PW) back to the reference's row-major
    (q, 1, lh, lw) (tests / debugging only; the hot path never untiles)."""
    q, _, ph, pw = level.shape
    return (level.reshape(q, ph // 4, pw // 8, 4, 8).permute(0, 1, 3, 2, 4).reshape(q, 1, ph, pw)
            [:, :, :lh, :lw].contiguous())


def untile_level0(level0: Tensor) -> Tensor:
    return untile_level(level0, level0.shape[-2], level0.shape[-1])


def _pyramid_shapes(n: int, h: int, w: int, num_levels: int, tiled_levels: int):
    return [(n * h * w, 1, *level_storage_shape(h, w, l, bool((tiled_levels >> l) & 1)))
            for l in range(num_levels)]


def corr_build(feat1: Tensor, feat2: Tensor, num_levels: int = 4,
               out: Optional[List[Tensor]] = None, tiled_levels: int = 0) -> List[Tensor]:
    """CorrelationPyramid.forward (raft_decoder.py:35-58) -> list of (N*h*w, 1, h>>l, w>>l).
    ``tiled_levels`` (bit mask, see ``pyramid_layout``): those levels are stored in 128-byte 8x4
    tiles for the lookup -- same values, permuted inside each query's map, the map padded to 4 rows x
    8 columns (``level_storage_shape``); pass the same mask to ``corr_lookup``."""
    p1 = _dense(feat1, 'feat1')
    p2 = _dense(feat2, 'feat2')
    if feat1.shape != feat2.shape or feat1.dim() != 4:
        raise _lib.ScflowHipError('feat1/feat2 must be equal-shape NCHW')
    n, c, h, w = feat1.shape
    shapes = _pyramid_shapes(n, h, w, num_levels, tiled_levels)
    if out is None:
        out = [torch.empty(sh, dtype=torch.float32, device=feat1.device) for sh in shapes]
    for l, (t, sh) in enumerate(zip(out, shapes)):
        if tuple(t.shape) != sh:
            raise _lib.ScflowHipError(f'pyramid level {l} has shape {tuple(t.shape)}, expected {sh}')
    arr = (C.c_void_p * num_levels)(*[_dense(t, 'level') for t in out])
    lib = _lib.load()
    timed = _CORR_TIMERS is not None      # bench.py: a timer bound to the FIRST launch = the contraction
    if timed:
        lib.scf_timer_arm(_CORR_TIMERS.take())
    try:
        _lib.check(lib.scf_corr_build_ex(p1, p2, arr, n, c, h, w, num_levels, int(tiled_levels),
                                         _stream()), 'scf_corr_build')
    finally:
        if timed:
            lib.scf_timer_arm(None)
    return out


def corr_lookup(pyramid: Sequence[Tensor], flow: Tensor, radius: int = 4,
                out: Optional[Tensor] = None, tiled_levels: int = 0) -> Tensor:
    """CorrLookup.forward (corr_lookup.py:102-136) -> (N, L*(2r+1)^2, h, w)."""
    pf = _dense(flow, 'flow')
    n, two, h, w = flow.shape
    if two != 2:
        raise _lib.ScflowHipError('flow must be (N,2,h,w)')
    L = len(pyramid)
    for l, (lv, sh) in enumerate(zip(pyramid, _pyramid_shapes(n, h, w, L, tiled_levels))):
        if tuple(lv.shape) != sh:
            raise _lib.ScflowHipError(f'pyramid level {l} has shape {tuple(lv.shape)}, expected {sh}')
    k = L * (2 * radius + 1) ** 2
    if out is None:
        out = torch.empty((n, k, h, w), dtype=torch.float32, device=flow.device)
    arr = (C.c_void_p * L)(*[_dense(t, 'level') for t in pyramid])
    lib = _lib.load()
    timed = _LOOKUP_TIMERS is not None      # bench.py: a HIP start/stop event pair bound to this launch
    if timed:
        lib.scf_timer_arm(_LOOKUP_TIMERS.take())
    try:
        _lib.check(lib.scf_corr_lookup_ex(arr, pf, _dense(out, 'out'), n, h, w, radius, L,
                                          int(tiled_levels), _stream()), 'scf_corr_lookup')
    finally:
        if timed:
            lib.scf_timer_arm(None)
    return out


class _TimerPool:
    """launch-bound timers (scf_timer_*), created up front so that a timed loop only takes one
    from the pool; ``read()`` returns the durations of the timers used since the last reset."""

    def __init__(self, reserve: int) -> None:
        self.timers, self.used = [], 0
        self._grow(max(int(reserve), 1))

    def _grow(self, n: int) -> None:
        lib = _lib.load()
        for _ in range(n):
            tm = C.c_void_p()
            _lib.check(lib.scf_timer_create(C.byref(tm)), 'scf_timer_create')
            self.timers.append(tm)

    def take(self):
        if self.used == len(self.timers):     # pool exhausted: grow (outside the fast path by design)
            self._grow(len(self.timers))
        tm = self.timers[self.used]
        self.used += 1
        return tm

    def read(self):
        lib = _lib.load()
        out = []
        for tm in self.timers[:self.used]:
            us = C.c_float()
            _lib.check(lib.scf_timer_elapsed_us(tm, C.byref(us)), 'scf_timer_elapsed_us')
            out.append(float(us.value))
        return out

    def reset(self) -> None:
        self.used = 0

    def destroy(self) -> None:
        lib = _lib.load()
        for tm in self.timers:
            lib.scf_timer_destroy(tm)
        self.timers, self.used = [], 0


_LOOKUP_TIMERS: Optional[_TimerPool] = None
_CORR_TIMERS: Optional[_TimerPool] = None


def corr_build_timing(enable: bool, reserve: int = 16):
    """like ``lookup_timing`` for the correlation build: every ``corr_build`` call from now on carries a
    timer bound to its first kernel launch -- the contraction (with the layout the decoders use: the
    GEMM with the fused first pool, ``corr_gemm_kernel<true, true>``), not the pooling cascade behind
    it.  enable=False: stop, synchronise, return the durations in microseconds."""
    global _CORR_TIMERS
    if enable:
        if _CORR_TIMERS is not None:
            _CORR_TIMERS.destroy()
        _CORR_TIMERS = _TimerPool(reserve)
        return None
    pool, _CORR_TIMERS = _CORR_TIMERS, None
    if pool is None:
        return []
    torch.cuda.synchronize()
    out = pool.read()
    pool.destroy()
    return out


def corr_build_timing_reset() -> None:
    if _CORR_TIMERS is not None:
        torch.cuda.synchronize()
        _CORR_TIMERS.reset()


def corr_build_timing_read():
    if _CORR_TIMERS is None:
        return []
    torch.cuda.synchronize()
    return _CORR_TIMERS.read()


def lookup_timing(enable: bool, reserve: int = 64):
    """enable=True: every corr-lookup launch from now on carries a timer (HIP start / stop events
    bound to the launch on its stream: the kernel's own duration); ``reserve`` timers are created
    now, before any timed loop.  enable=False: stop, synchronise and return the per-launch
    durations in microseconds since the last ``lookup_timing_reset``."""
    global _LOOKUP_TIMERS
    if enable:
        if _LOOKUP_TIMERS is not None:
            _LOOKUP_TIMERS.destroy()
        _LOOKUP_TIMERS = _TimerPool(reserve)
        return None
    pool, _LOOKUP_TIMERS = _LOOKUP_TIMERS, None
    if pool is None:
        return []
    torch.cuda.synchronize()
    out = pool.read()
    pool.destroy()
    return out


def lookup_timing_reset() -> None:
    """forget the launches recorded so far (the timers are reused)."""
    if _LOOKUP_TIMERS is not None:
        torch.cuda.synchronize()
        _LOOKUP_TIMERS.reset()


def lookup_timing_read():
    """synchronise and return the durations (us) recorded since the last reset."""
    if _LOOKUP_TIMERS is None:
        return []
    torch.cuda.synchronize()
    return _LOOKUP_TIMERS.read()


# ------------------------------------------------------------- norms etc.
def instance_norm(x: Tensor, res: Optional[Tensor] = None, relu: bool = False,
                  out: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
    px = _dense(x, 'x')
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().scf_instance_norm(px, _opt(res, 'res'), _dense(out, 'out'), n * c,
                                             h * w, eps, int(relu), _stream()),
               'scf_instance_norm')
    return out


def group_norm_relu(x: Tensor, gamma: Tensor, beta: Tensor, groups: int, eps: float = 1e-5,
                    out: Optional[Tensor] = None) -> Tensor:
    """GroupNorm(groups, eps, affine) + ReLU of an (N, C, H, W) map, or of the (S, N, C, H, W) partial sums of
    ``conv2d(..., kslices=S)`` (added in slice order inside the kernel) -> (N, C, H, W)."""
    px = _dense(x, 'x')
    parts, pstride = 1, 0
    if x.dim() == 5:
        parts, pstride = x.shape[0], x.stride(0)
        x = x[0]
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().scf_group_norm_relu_parts(px, parts, pstride, _dense(gamma, 'gamma'), _dense(beta, 'beta'),
                                                     _dense(out, 'out'), n, c, h * w, groups, eps,
                                                     _stream()), 'scf_group_norm_relu')
    return out


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: int = ACT_NONE,
           out: Optional[Tensor] = None) -> Tensor:
    px = _dense(x, 'x')
    n, k = x.shape
    o = weight.shape[0]
    if weight.shape[1] != k:
        raise _lib.ScflowHipError('linear: weight/in_features mismatch')
    if out is None:
        out = torch.empty((n, o), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().scf_linear(px, _dense(weight, 'weight'), _opt(bias, 'bias'),
                                      _dense(out, 'out'), n, k, o, act, _stream()), 'scf_linear')
    return out


def fc_slices(k: int) -> int:
    """K-slices ``fc_splitk`` is given for an input of ``k`` features: 256 features per slice (one block tile's
    worth of LDS) where K allows it, else 0 = the shape does not fit (callers use ``linear``)."""
    if k % 8 == 0 and k <= 256:
        return 1
    return k // 256 if k % 256 == 0 else 0


def fc_splitk(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, x_bias: Optional[Tensor] = None,
              x_relu: bool = False, gn=None, weight2: Optional[Tensor] = None, bias2: Optional[Tensor] = None,
              act: int = ACT_NONE, slices: int = 1):
    """nn.Linear as a split-K MFMA GEMM with fused neighbours (``scf_fc_splitk``).  ``x``: (N, K), or
    (parts, N, K) partial sums of a previous ``fc_splitk(..., slices=parts)`` whose bias / ReLU this call applies
    (``x_bias``, ``x_relu``); ``gn`` = (groups, hw, gamma, beta, eps): GroupNorm + affine + ReLU over the features
    first (the flattened (C, h, w) map of the pose head, hw = h * w).  ``slices`` > 1: returns (slices, N, O)
    partial sums (no bias / act); 1: the finished (N, O) [and (N, O2) for ``weight2``]."""
    _dev(x, 'x')
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if x.dim() != 3 or not x.is_contiguous():
        raise _lib.ScflowHipError('fc_splitk: x must be a contiguous (N, K) or (parts, N, K) tensor')
    parts, n, k = x.shape
    o = weight.shape[0]
    if weight.shape[1] != k or (weight2 is not None and weight2.shape[1] != k):
        raise _lib.ScflowHipError('fc_splitk: weight / in_features mismatch')
    d = _lib.FcDesc()
    d.x, d.x_parts, d.x_part_stride = x.data_ptr(), parts, n * k
    d.x_bias, d.x_relu = _opt(x_bias, 'x_bias'), int(bool(x_relu))
    if gn is not None:
        groups, hw, gamma, beta, eps = gn
        d.gn_groups, d.gn_hw, d.gn_gamma, d.gn_beta, d.gn_eps = groups, hw, _dense(gamma, 'gamma'), _dense(beta, 'beta'), eps
    d.N, d.K = n, k
    y = torch.empty((slices, n, o) if slices > 1 else (n, o), dtype=torch.float32, device=x.device)
    d.W, d.bias, d.y, d.O = _dense(weight, 'weight'), _opt(bias, 'bias'), y.data_ptr(), o
    y2 = None
    if weight2 is not None:
        y2 = torch.empty((n, weight2.shape[0]), dtype=torch.float32, device=x.device)
        d.W2, d.bias2, d.y2, d.O2 = _dense(weight2, 'weight2'), _opt(bias2, 'bias2'), y2.data_ptr(), weight2.shape[0]
    d.act, d.slices = act, slices
    _lib.check(_lib.load().scf_fc_splitk(C.byref(d), _stream()), 'scf_fc_splitk')
    return y if y2 is None else (y, y2)


def linear_pair(x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor],
                act: int = ACT_NONE) -> Tuple[Tensor, Tensor]:
    """(linear(x, w1, b1), linear(x, w2, b2)) in one launch (scf_linear_pair)."""
    px = _dense(x, 'x')
    n, k = x.shape
    if w1.shape[1] != k or w2.shape[1] != k:
        raise _lib.ScflowHipError('linear_pair: weight/in_features mismatch')
    y1 = torch.empty((n, w1.shape[0]), dtype=torch.float32, device=x.device)
    y2 = torch.empty((n, w2.shape[0]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().scf_linear_pair(px, _dense(w1, 'w1'), _opt(b1, 'b1'), y1.data_ptr(),
                                           w1.shape[0], _dense(w2, 'w2'), _opt(b2, 'b2'),
                                           y2.data_ptr(), w2.shape[0], n, k, act, _stream()),
               'scf_linear_pair')
    return y1, y2


def pose_update(rot_all: Tensor, trans_all: Tensor, label: Tensor, num_class: int,
                rot: Tensor, trans: Tensor, label_mode: int = 0):
    """-> (d_rot (N,6), d_trans (N,3), R' (N,3,3), t' (N,3))."""
    n = rot_all.shape[0]
    if not label.is_cuda or label.dtype != torch.int64 or not label.is_contiguous():
        raise _lib.ScflowHipError('label must be a contiguous int64 GPU tensor')
    dev = rot_all.device
    d_rot = torch.empty((n, 6), dtype=torch.float32, device=dev)
    d_trans = torch.empty((n, 3), dtype=torch.float32, device=dev)
    r_out = torch.empty((n, 3, 3), dtype=torch.float32, device=dev)
    t_out = torch.empty((n, 3), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().scf_pose_update(
        _dense(rot_all, 'rot_all'), _dense(trans_all, 'trans_all'), label.data_ptr(), num_class,
        label_mode, _dense(rot, 'rot'), _dense(trans, 'trans'), d_rot.data_ptr(),
        d_trans.data_ptr(), r_out.data_ptr(), t_out.data_ptr(), n, _stream()), 'scf_pose_update')
    return d_rot, d_trans, r_out, t_out


def reproject_flow(depth: Tensor, k: Tensor, rot0: Tensor, trans0: Tensor, rot: Tensor,
                   trans: Tensor, invalid_num: float = 0., out: Optional[Tensor] = None) -> Tensor:
    n, h, w = depth.shape
    if out is None:
        out = torch.empty((n, 2, h, w), dtype=torch.float32, device=depth.device)
    _lib.check(_lib.load().scf_reproject_flow(
        _dense(depth, 'depth'), _dense(k, 'k'), _dense(rot0, 'rot0'), _dense(trans0, 'trans0'),
        _dense(rot, 'rot'), _dense(trans, 'trans'), _dense(out, 'out'), n, h, w,
        float(invalid_num), _stream()), 'scf_reproject_flow')
    return out


def filter_flow_by_mask_(flow: Tensor, mask: Tensor, invalid_num: float = 400.,
                         align_corners: bool = False) -> Tensor:
    """in-place ``filter_flow_by_mask`` (utils/flow.py:6-26); flow (N,2,H,W), mask (N,H,W)."""
    n, two, h, w = flow.shape
    if two != 2 or tuple(mask.shape) != (n, h, w):
        raise _lib.ScflowHipError('filter_flow_by_mask: flow (N,2,H,W), mask (N,H,W)')
    _lib.check(_lib.load().scf_filter_flow_by_mask(_dense(flow, 'flow'), _dense(mask, 'mask'), n, h, w,
                                                   float(invalid_num), int(align_corners), _stream()),
               'scf_filter_flow_by_mask')
    return flow


def unproject_depth(depth: Tensor, k: Tensor, rot0: Tensor, trans0: Tensor) -> Tensor:
    n, h, w = depth.shape
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=depth.device)
    _lib.check(_lib.load().scf_unproject_depth(
        _dense(depth, 'depth'), _dense(k, 'k'), _dense(rot0, 'rot0'), _dense(trans0, 'trans0'),
        out.data_ptr(), n, h, w, _stream()), 'scf_unproject_depth')
    return out


def resize_bilinear(a: Tensor, out_hw: Tuple[int, int], mul: float = 1.0,
                    b: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """mul * F.interpolate(a + b, size=out_hw, mode='bilinear', align_corners=True)."""
    pa = _dense(a, 'a')
    n, c, h, w = a.shape
    ho, wo = out_hw
    if out is None:
        out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().scf_resize_bilinear(pa, _opt(b, 'b'), _dense(out, 'out'), n * c, h, w,
                                               ho, wo, float(mul), _stream()),
               'scf_resize_bilinear')
    return out


def convex_upsample(x: Tensor, mask: Tensor, scale: int = 8, x_mul: float = 1.0,
                    mask_mul: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """RAFT convex up-sampling (raft_decoder.py:381-416): x (N,C,h,w), mask (N,9*scale^2,h,w)
    -> (N,C,scale*h,scale*w)."""
    px = _dense(x, 'x')
    n, c, h, w = x.shape
    if tuple(mask.shape) != (n, 9 * scale * scale, h, w):
        raise _lib.ScflowHipError(f'mask has shape {tuple(mask.shape)}')
    if out is None:
        out = torch.empty((n, c, scale * h, scale * w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().scf_convex_upsample(px, _dense(mask, 'mask'), _dense(out, 'out'), n, c,
                                               h, w, scale, float(x_mul), float(mask_mul),
                                               _stream()), 'scf_convex_upsample')
    return out


def avgpool2x2(x: Tensor) -> Tensor:
    px = _dense(x, 'x')
    n, c, h, w = x.shape
    out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().scf_avgpool2x2(px, out.data_ptr(), n * c, h, w, _stream()),
               'scf_avgpool2x2')
    return out


def mul_mask(x: Tensor, mask: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """x * mask with mask (N, 1, H, W) broadcast over the channels (scflow_decoder.py:199-205)."""
    px, n, c, h, w, sx = _nchw(x, 'x')
    if tuple(mask.shape) != (n, 1, h, w):
        raise _lib.ScflowHipError(f'mask has shape {tuple(mask.shape)}, expected {(n, 1, h, w)}')
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    po, n2, c2, h2, w2, so = _nchw(out, 'out')
    if (n2, c2, h2, w2) != (n, c, h, w):
        raise _lib.ScflowHipError('mul_mask: out shape mismatch')
    _lib.check(_lib.load().scf_mul_mask(px, sx, _dense(mask, 'mask'), po, so, n, c, h * w, _stream()),
               'scf_mul_mask')
    return out


def copy_channels(src: Tensor, dst: Tensor) -> Tensor:
    """copy a sample-strided NCHW tensor into another (same N, C, H, W)."""
    ps, n, c, h, w, ss = _nchw(src, 'src')
    pd, n2, c2, h2, w2, sd = _nchw(dst, 'dst')
    if (n, c, h, w) != (n2, c2, h2, w2):
        raise _lib.ScflowHipError('copy_channels shape mismatch')
    _lib.check(_lib.load().scf_copy_strided(ps, ss, pd, sd, n, c * h * w, _stream()),
               'scf_copy_strided')
    return dst
