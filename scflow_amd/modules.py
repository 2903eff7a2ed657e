"""Host-side mirror of the reference's encoder / decoder / head classes.

Each class keeps the reference's constructor signature, registry name and
``state_dict`` key layout (tests/golden/state_dict_keys.json),
so the reference ``model=`` config dict and reference checkpoints apply
unchanged -- but ``forward`` only sequences hand-written gfx950 kernels through
``scflow_amd.ops`` (C ABI, include/scflow_hip.h).  The ``torch.nn`` leaf
modules below are parameter containers: their own ``forward`` is never called,
and there is no CPU fallback (ops reject non-GPU tensors).

Kernel-layout copies of the parameters (``PackedConv``) are built lazily on the
parameters' device and dropped whenever the module is moved or re-loaded.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .ops import (ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, CONV_GRU_Q, CONV_GRU_ZR, PackedConv)
from .registry import DECODERS, ENCODERS, HEAD, build_head

Tensor = torch.Tensor

_ACTS = {None: ACT_NONE, 'ReLU': ACT_RELU, 'Sigmoid': ACT_SIGMOID, 'Tanh': ACT_TANH}


def _act_code(act_cfg: Optional[dict]) -> int:
    if act_cfg is None:
        return ACT_NONE
    kind = act_cfg.get('type')
    if kind not in _ACTS:
        raise NotImplementedError(f'activation {kind} has no HIP epilogue')
    return _ACTS[kind]


class HipModule(nn.Module):
    """nn.Module with a kernel-layout copy of its parameters (``packed``).

    The copy is keyed on the identity AND version of every tensor it was built from
    (``(data_ptr, _version)`` of ``_pack_sources()``), so it follows the ways weights normally
    change: ``.to()``, ``load_state_dict`` on this module or on any wrapper / parent (mmcv's
    ``load_checkpoint`` recurses through ``_load_from_state_dict`` and never calls this
    class's ``load_state_dict``), in-place ops on the parameter under ``torch.no_grad()``
    (``weight.mul_``, ``weight.copy_``) and ``nn.init`` -- all of them bump ``_version`` or replace
    the storage.  NOT covered: edits through ``param.data`` (``param.data.copy_(...)``,
    ``param.data.mul_(...)``): ``.data`` is a detached alias with its own version counter, so the
    parameter's ``_version`` stays put.  After such an edit call ``invalidate_packed()``."""

    def invalidate_packed(self) -> None:
        """drop every kernel-layout weight copy held by this module and its children; they are
        rebuilt from the current parameters at the next forward.  Needed only after weight edits
        the version key cannot see (``param.data.copy_`` and friends)."""
        self._drop_packed()
        for m in self.modules():
            m.__dict__.pop('_ctx_cache', None)

    def _drop_packed(self) -> None:
        for m in self.modules():
            if isinstance(m, HipModule):
                m.__dict__['_packed'] = None
                m.__dict__.pop('_pack_refs', None)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._drop_packed()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._drop_packed()
        return out

    def _pack(self):
        raise NotImplementedError

    def _pack_modules(self):
        """the modules whose own parameters / buffers ``_pack`` reads (default: this module and its
        non-HipModule descendants, e.g. nn.Conv2d / norm leaves)."""
        out, stack = [], [self]
        while stack:
            m = stack.pop()
            out.append(m)
            for c in m._modules.values():
                if c is not None and not isinstance(c, HipModule):
                    stack.append(c)
        return out

    def _pack_sources(self):
        """the tensors ``_pack`` reads."""
        return [t for m in self._pack_modules() for d in (m._parameters, m._buffers)
                for t in d.values() if t is not None]

    def _pack_key(self):
        # (registry dict, name) pairs are resolved once per module (the module TREE is fixed after
        # construction); the tensors are looked up through them on every call, so a replaced
        # Parameter object (load_state_dict(assign=True), setattr) is seen like an in-place edit
        refs = self.__dict__.get('_pack_refs')
        if refs is None:
            if type(self)._pack_sources is not HipModule._pack_sources:       # composite modules name their sources
                return tuple((t.data_ptr(), t._version) for t in self._pack_sources())
            refs = self.__dict__['_pack_refs'] = [(d, n) for m in self._pack_modules()
                                                  for d in (m._parameters, m._buffers) for n in d]
        key = []
        for d, n in refs:
            t = d.get(n)
            if t is not None:
                key.append((t.data_ptr(), t._version))
        return tuple(key)

    @property
    def packed(self):
        d = self.__dict__
        key = self._pack_key()
        if d.get('_packed') is None or d.get('_packed_key') != key:
            with torch.no_grad():
                d['_packed'] = self._pack()
            d['_packed_key'] = key
        return d['_packed']


class ConvBlock(HipModule):
    """parameter layout of mmcv ``ConvModule``: ``.conv`` (+ ``.gn``); conv -> norm -> act,
    bias iff no norm."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, act_cfg=dict(type='ReLU'),
                 norm_cfg: Optional[dict] = None):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding,
                              bias=norm_cfg is None)
        self.act = _act_code(act_cfg)
        self.groups = None
        if norm_cfg is not None:
            if norm_cfg.get('type') != 'GN':
                raise NotImplementedError('ConvBlock supports GN only (pose head)')
            self.groups = norm_cfg['num_groups']
            self.gn = nn.GroupNorm(self.groups, cout, eps=norm_cfg.get('eps', 1e-5))

    def _pack(self) -> PackedConv:
        c = self.conv
        return PackedConv.from_weight(c.weight, c.bias, stride=c.stride[0], padding=c.padding)

    def forward(self, x0: Tensor, x1: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
        if self.groups is None:
            return ops.conv2d(self.packed, x0, x1, out=out, act=self.act)
        if self.act != ACT_RELU:
            raise NotImplementedError('GN is fused with ReLU only')
        # small grids: K split across blocks, the normalisation adds the partial tensors (ops.conv_kslices)
        ks = ops.conv_kslices_for(self.packed, x0, x1)
        y = ops.conv2d(self.packed, x0, x1, kslices=ks)
        return ops.group_norm_relu(y, self.gn.weight, self.gn.bias, self.groups, self.gn.eps, out=out)


# =============================================================== encoder
class _BasicBlock(HipModule):
    """backbone/resnet.py:14-94 parameters: conv1, {in,bn}1, conv2, {in,bn}2, downsample."""

    def __init__(self, inplanes, planes, stride, kind, downsample: bool):
        super().__init__()
        tag = 'in' if kind == 'IN' else 'bn'
        mk = (lambda c: nn.InstanceNorm2d(c)) if kind == 'IN' else (lambda c: nn.BatchNorm2d(c))
        self.kind, self.stride, self.tag = kind, stride, tag
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=True)
        setattr(self, tag + '1', mk(planes))
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=True)
        setattr(self, tag + '2', mk(planes))
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=True),
                                            mk(planes))

    def _bn(self, m):
        return (m.weight, m.bias, m.running_mean, m.running_var) if self.kind == 'BN' else None

    def _pack(self):
        n1, n2 = getattr(self, self.tag + '1'), getattr(self, self.tag + '2')
        eps = n1.eps
        p = dict(c1=PackedConv.from_weight(self.conv1.weight, self.conv1.bias, self.stride, 1,
                                           bn=self._bn(n1), eps=eps),
                 c2=PackedConv.from_weight(self.conv2.weight, self.conv2.bias, 1, 1,
                                           bn=self._bn(n2), eps=eps))
        if self.downsample is not None:
            ds = self.downsample[0]
            p['ds'] = PackedConv.from_weight(ds.weight, ds.bias, self.stride, 0,
                                             bn=self._bn(self.downsample[1]), eps=eps)
        return p

    def forward(self, x: Tensor) -> Tensor:
        """resnet.py:67-94.  BN (eval) is folded into the conv epilogue; IN needs the
        whole plane, so it is its own kernel (residual add + ReLU fused there)."""
        p = self.packed
        if self.kind == 'BN':
            y = ops.conv2d(p['c1'], x, act=ACT_RELU)
            idt = ops.conv2d(p['ds'], x) if 'ds' in p else x
            return ops.conv2d(p['c2'], y, res=idt, act=ACT_RELU)
        y = ops.conv2d(p['c1'], x)
        ops.instance_norm(y, relu=True, out=y)
        y2 = ops.conv2d(p['c2'], y)
        if 'ds' in p:
            idt = ops.conv2d(p['ds'], x)
            ops.instance_norm(idt, out=idt)
        else:
            idt = x
        return ops.instance_norm(y2, res=idt, relu=True, out=y2)


@ENCODERS.register_module()
class RAFTEncoder(HipModule):
    """encoder/raft_encoder.py:13-314, net_type 'Basic' (the SCFlow config):
    7x7/s2 stem, three stages of two BasicBlocks (64, 96, 128; strides 1, 2, 2), 1x1 head.
    ``norm_cfg`` IN -> feature encoder, BN -> context encoder (eval statistics)."""

    def __init__(self, in_channels: int, out_channels: int, scale: float = 1 / 8,
                 net_type: str = 'Basic', norm_cfg: dict = dict(type='BN', requires_grad=True),
                 init_cfg=None, **unsupported) -> None:
        super().__init__()
        if net_type != 'Basic':
            raise NotImplementedError("only net_type='Basic' is built (SCFlow config)")
        bad = {k: v for k, v in unsupported.items()
               if v not in (None, False, -1) and k not in ('conv_cfg',)}
        if bad:
            raise NotImplementedError(f'RAFTEncoder options not built: {sorted(bad)}')
        kind = norm_cfg['type']
        if kind not in ('IN', 'BN', 'SyncBN'):
            raise NotImplementedError(f'norm {kind}')
        kind = 'IN' if kind == 'IN' else 'BN'
        self.kind, self.tag = kind, ('in' if kind == 'IN' else 'bn')
        self.in_channels, self.out_channels, self.scale = in_channels, out_channels, scale
        mk = (lambda c: nn.InstanceNorm2d(c)) if kind == 'IN' else (lambda c: nn.BatchNorm2d(c))
        self.stem_stride = 1 if scale == 1 / 4 else 2
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=self.stem_stride, padding=3, bias=True)
        setattr(self, self.tag + '1', mk(64))
        inplanes = 64
        self.res_layers = []
        for i, (planes, stride) in enumerate(zip((64, 96, 128), (1, 2, 2)), start=1):
            layer = nn.Sequential(
                _BasicBlock(inplanes, planes, stride, kind, stride != 1 or inplanes != planes),
                _BasicBlock(planes, planes, 1, kind, False))
            self.add_module(f'res_layer{i}', layer)
            self.res_layers.append(f'res_layer{i}')
            inplanes = planes
        self.conv2 = nn.Conv2d(128, out_channels, 1)

    def _pack(self):
        n1 = getattr(self, self.tag + '1')
        bn = (n1.weight, n1.bias, n1.running_mean, n1.running_var) if self.kind == 'BN' else None
        return dict(stem=PackedConv.from_weight(self.conv1.weight, self.conv1.bias,
                                                self.stem_stride, 3, bn=bn, eps=n1.eps),
                    head=PackedConv.from_weight(self.conv2.weight, self.conv2.bias, 1, 0))

    def forward(self, x: Tensor, out: Optional[Tensor] = None, head_act: int = ACT_NONE,
                head_act2: int = ACT_NONE, head_split: int = 0) -> Tensor:
        """raft_encoder.py:286-314.  ``out`` / ``head_*`` let the caller have the 1x1 head
        write (with a split tanh/relu epilogue) straight into a slice of a larger buffer."""
        p = self.packed
        if self.kind == 'BN':
            x = ops.conv2d(p['stem'], x, act=ACT_RELU)
        else:
            x = ops.conv2d(p['stem'], x)
            ops.instance_norm(x, relu=True, out=x)
        for name in self.res_layers:
            for blk in getattr(self, name):
                x = blk(x)
        return ops.conv2d(p['head'], x, out=out, act=head_act, act2=head_act2,
                          act_split=head_split)


def raft_encoder_pair(fe: 'RAFTEncoder', xf: Tensor, ce: 'RAFTEncoder', xc: Tensor, out_c: Optional[Tensor] = None,
                      head_act: int = ACT_NONE, head_act2: int = ACT_NONE, head_split: int = 0) -> Tuple[Tensor, Tensor]:
    """the feature encoder (InstanceNorm) on ``xf`` and the context encoder (BatchNorm, folded) on ``xc`` -- two independent
    passes over layers of identical geometry -- walked TOGETHER, every pair of convolutions through ``ops.conv2d_pair``: at
    batch 1-4 the context encoder's launches ride in the feature encoder's wherever both grids fit the chip (r6; before: one
    after the other, or side by side on a second stream).  Same kernels per layer as the two separate passes, same bits."""
    if fe.kind != 'IN' or ce.kind != 'BN':
        raise ValueError('raft_encoder_pair: (InstanceNorm feature encoder, BatchNorm context encoder)')
    pf, pc = fe.packed, ce.packed
    yf, yc = ops.conv2d_pair((pf['stem'], xf), (pc['stem'], xc, dict(act=ACT_RELU)))
    ops.instance_norm(yf, relu=True, out=yf)
    for name in fe.res_layers:
        for bf, bc in zip(getattr(fe, name), getattr(ce, name)):
            qf, qc = bf.packed, bc.packed
            tf_, tc_ = ops.conv2d_pair((qf['c1'], yf), (qc['c1'], yc, dict(act=ACT_RELU)))
            ops.instance_norm(tf_, relu=True, out=tf_)
            if 'ds' in qf:
                if_, ic_ = ops.conv2d_pair((qf['ds'], yf), (qc['ds'], yc))
                ops.instance_norm(if_, out=if_)
            else:
                if_, ic_ = yf, yc
            uf_, yc = ops.conv2d_pair((qf['c2'], tf_), (qc['c2'], tc_, dict(res=ic_, act=ACT_RELU)))
            yf = ops.instance_norm(uf_, res=if_, relu=True, out=uf_)
    return ops.conv2d_pair((pf['head'], yf), (pc['head'], yc, dict(out=out_c, act=head_act, act2=head_act2,
                                                                  act_split=head_split)))


# =============================================================== decoder
class CorrelationPyramid(HipModule):
    """decoder/raft_decoder.py:19-58."""

    def __init__(self, num_levels: int = 4) -> None:
        super().__init__()
        self.num_levels = num_levels

    def forward(self, feat1: Tensor, feat2: Tensor, tiled_levels: int = 0) -> List[Tensor]:
        """``tiled_levels`` (decoder-internal, ``ops.pyramid_layout``): those levels in the lookup's
        8x4-tile layout; the default is the reference's row-major pyramid."""
        return ops.corr_build(feat1, feat2, self.num_levels, tiled_levels=tiled_levels)


class CorrLookup(HipModule):
    """utils/corr_lookup.py:71-136 (bilinear, zeros padding, align_corners=True only)."""

    def __init__(self, radius: int = 4, mode: str = 'bilinear', padding_mode: str = 'zeros',
                 align_corners: bool = True) -> None:
        super().__init__()
        if mode != 'bilinear' or padding_mode != 'zeros' or not align_corners:
            raise NotImplementedError('HIP CorrLookup: bilinear / zeros / align_corners=True')
        self.r = radius

    def forward(self, corr_pyramid: Sequence[Tensor], flow: Tensor,
                tiled_levels: int = 0) -> Tensor:
        return ops.corr_lookup(corr_pyramid, flow, self.r, tiled_levels=tiled_levels)


def _pyramid_layout(feat: Tensor, radius: int, num_levels: int, allow: bool = True) -> int:
    """the decoders keep the pyramid to themselves, so they are free to store it in the layout the
    lookup likes best (``decoder.tiled_pyramid = False`` forces the reference layout: A/B
    measurements in tools/)."""
    return ops.pyramid_layout(feat.shape[-2], feat.shape[-1], radius, num_levels) if allow else 0


class MotionEncoder(HipModule):
    """decoder/raft_decoder.py:61-166, 'Basic' channel plan."""

    def __init__(self, num_levels: int = 4, radius: int = 4, net_type: str = 'Basic',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU')) -> None:
        super().__init__()
        if net_type not in ('Basic', 'Large') or norm_cfg is not None:
            raise NotImplementedError('MotionEncoder: Basic, no norm')
        cin = num_levels * (2 * radius + 1) ** 2
        self.corr_net = nn.Sequential(ConvBlock(cin, 256, 1, padding=0, act_cfg=act_cfg),
                                      ConvBlock(256, 192, 3, padding=1, act_cfg=act_cfg))
        self.flow_net = nn.Sequential(ConvBlock(2, 128, 7, padding=3, act_cfg=act_cfg),
                                      ConvBlock(128, 64, 3, padding=1, act_cfg=act_cfg))
        self.out_net = nn.Sequential(ConvBlock(256, 126, 3, padding=1, act_cfg=act_cfg))
        self.out_channels = [126]

    def forward(self, corr: Tensor, flow: Tensor, out: Optional[Tensor] = None,
                overlap: bool = False, cf: Optional[Tensor] = None, fork=None) -> Tensor:
        """raft_decoder.py:152-166.  -> (N, 128, h, w) = [out_net(126) | flow(2)], optionally
        written into ``out`` (a channel slice of the GRU input buffer).  ``overlap`` (small
        batches): the flow branch runs on the side stream, from the caller's ``fork`` event
        (``ops.fork_point()``, taken once ``flow`` is enqueued); the caller then also passes ``cf``
        (N, 256, h, w), allocated BEFORE that fork point (the side branch writes it: see
        SCFlowRefiner.extract_feat for the allocator rule)."""
        n, _, h, w = flow.shape
        dev = flow.device
        if out is None:
            out = torch.empty((n, 128, h, w), dtype=torch.float32, device=dev)
        if cf is None:
            if overlap:
                raise ValueError('overlap=True needs a cf buffer allocated before the fork point')
            cf = torch.empty((n, 256, h, w), dtype=torch.float32, device=dev)
        if not overlap and 'flow' in ops.PAIR_BRANCHES and ops._CONV_EVENTS is None:
            # r6: the flow branch rides in the correlation branch's launches, layer by layer (ops.conv2d_pair: one launch where
            # the two grids need fewer rounds of resident blocks together, else one after the other; same bits)
            cn, fn = self.corr_net, self.flow_net
            c1, f1 = ops.conv2d_pair((cn[0].packed, corr, dict(act=cn[0].act)), (fn[0].packed, flow, dict(act=fn[0].act)))
            ops.conv2d_pair((cn[1].packed, c1, dict(out=cf[:, :192], act=cn[1].act)),
                            (fn[1].packed, f1, dict(out=cf[:, 192:], act=fn[1].act)))
            del c1, f1
        else:
            br = ops.side_stream(overlap, after=fork)
            with br:
                f1 = self.flow_net[0](flow)
                self.flow_net[1](f1, out=cf[:, 192:])
                del f1
            c1 = self.corr_net[0](corr)
            self.corr_net[1](c1, out=cf[:, :192])
            br.join()
        self.out_net[0](cf, out=out[:, :126])
        ops.copy_channels(flow, out[:, 126:128])
        return out


class ConvGRU(HipModule):
    """decoder/raft_decoder.py:168-253.  z and r share their input, so their weights are
    stacked into one 256-row convolution whose epilogue emits z and r*h; the q convolution's
    epilogue applies tanh and the state update (1-z)*h + z*q in place."""
    _kernel = {'Conv': [3], 'SeqConv': [(1, 5), (5, 1)]}
    _padding = {'Conv': [1], 'SeqConv': [(0, 2), (2, 0)]}

    def __init__(self, h_channels: int, x_channels: int, net_type: str = 'SeqConv') -> None:
        super().__init__()
        self.h_channels, self.x_channels = h_channels, x_channels
        mk = lambda act: nn.ModuleList([
            ConvBlock(h_channels + x_channels, h_channels, k, padding=p, act_cfg=dict(type=act))
            for k, p in zip(self._kernel[net_type], self._padding[net_type])])
        self.conv_z, self.conv_r, self.conv_q = mk('Sigmoid'), mk('Sigmoid'), mk('Tanh')

    def _pack_sources(self):          # the packing reads the ConvBlock children's tensors
        return [t for m in (*self.conv_z, *self.conv_r, *self.conv_q) for t in (m.conv.weight, m.conv.bias)]

    def _pack(self):
        packs = []
        for z, r, q in zip(self.conv_z, self.conv_r, self.conv_q):
            w = torch.cat([z.conv.weight, r.conv.weight], 0)
            b = torch.cat([z.conv.bias, r.conv.bias], 0)
            packs.append((PackedConv.from_weight(w, b, 1, z.conv.padding),
                          PackedConv.from_weight(q.conv.weight, q.conv.bias, 1, q.conv.padding)))
        return packs

    # ---- iteration-invariant context (DESIGN.md "GRU context hoisting") ----
    # x = [c | x'] where c (the context features) is the same in every refinement iteration:
    # conv([h | c | x']) = conv([h | x']) + conv_c(c).  ``context_terms`` evaluates conv_c(c) + bias
    # for z | r | q once per pair; ``forward_inplace`` with those terms convolves [h | x'] only.
    def _ctx_packs(self, cc: int):
        """per pass: (PackedConv c -> 3 h_channels rows [z | r | q] with the biases,
        PackedConv [h | x'] -> z | r without bias, PackedConv [h | x'] -> q without bias)"""
        hc = self.h_channels
        key = ('ctx', cc, self._pack_key())
        if self.__dict__.get('_ctx_cache', (None,))[0] != key:
            packs = []
            for z, r, q in zip(self.conv_z, self.conv_r, self.conv_q):
                wz, wr, wq = z.conv.weight.detach(), r.conv.weight.detach(), q.conv.weight.detach()
                sel = lambda w: torch.cat([w[:, :hc], w[:, hc + cc:]], 1)
                w_c = torch.cat([wz[:, hc:hc + cc], wr[:, hc:hc + cc], wq[:, hc:hc + cc]], 0)
                b_c = torch.cat([z.conv.bias, r.conv.bias, q.conv.bias], 0).detach()
                packs.append((PackedConv.from_weight(w_c, b_c, 1, z.conv.padding),
                              PackedConv.from_weight(torch.cat([sel(wz), sel(wr)], 0), None, 1, z.conv.padding),
                              PackedConv.from_weight(sel(wq), None, 1, q.conv.padding)))
            self.__dict__['_ctx_cache'] = (key, packs)
        return self.__dict__['_ctx_cache'][1]

    def context_terms(self, c: Tensor) -> List[Tensor]:
        """one (N, 3 h_channels, H, W) tensor per pass: the c part of conv_z | conv_r | conv_q + bias"""
        return [ops.conv2d(pk[0], c) for pk in self._ctx_packs(c.shape[1])]

    def forward_inplace(self, hx: Tensor, ctx: Optional[Sequence[Tensor]] = None,
                        ctx_channels: int = 0, scratch: Optional[Tensor] = None) -> Tensor:
        """hx: (N, h_ch + x_ch, h, w) = [h | x]; h is updated in place.  One C-ABI call
        (``scf_sepconv_gru``: the launch sequence lives in the library).  ``ctx`` =
        ``context_terms(hx[:, h_ch:h_ch + ctx_channels])`` of this pair: those channels are then
        not convolved again (``scf_sepconv_gru_ctx``)."""
        hc = self.h_channels
        n, _, h, w = hx.shape
        if scratch is None:          # (2, N, Ch, H, W): the z and r*h buffers (a caller's loop passes its own)
            scratch = torch.empty((2, n, hc, h, w), dtype=torch.float32, device=hx.device)
        z, rh = scratch[0], scratch[1]
        if ctx is None:
            ops.sepconv_gru(self.packed, hx, hc, z, rh)
        else:
            packs = [(pk[1], pk[2]) for pk in self._ctx_packs(ctx_channels)]
            ops.sepconv_gru(packs, hx, hc, z, rh, ctx=ctx, ctx_channels=ctx_channels)
        return hx[:, :hc]

    def forward(self, h: Tensor, x: Tensor) -> Tensor:
        """raft_decoder.py:235-253 signature (copies h, x into one buffer)."""
        n, _, hh, ww = h.shape
        hx = torch.empty((n, self.h_channels + self.x_channels, hh, ww), dtype=torch.float32,
                         device=h.device)
        ops.copy_channels(h, hx[:, :self.h_channels])
        ops.copy_channels(x, hx[:, self.h_channels:])
        return self.forward_inplace(hx)


class XHead(HipModule):
    """decoder/raft_decoder.py:256-294."""

    def __init__(self, in_channels: int, feat_channels: Sequence[int], x_channels: int, x: str):
        super().__init__()
        if len(feat_channels) != 1:
            raise NotImplementedError('XHead: one hidden layer')
        self.layers = nn.Sequential(ConvBlock(in_channels, feat_channels[0], 3, padding=1))
        k = 1 if x == 'mask' else 3
        self.predict_layer = nn.Conv2d(feat_channels[0], x_channels, k, padding=k // 2)

    def _pack(self) -> PackedConv:
        c = self.predict_layer
        return PackedConv.from_weight(c.weight, c.bias, 1, c.padding)

    def predict(self, feat: Tensor, act: int = ACT_NONE) -> Tensor:
        return ops.conv2d(self.packed, feat, act=act)

    def forward(self, x: Tensor) -> Tensor:
        return self.predict(self.layers[0](x))


@HEAD.register_module()
class MultiClassPoseHead(HipModule):
    """head/pose_head.py:110-211."""

    def __init__(self, num_class: int, in_channels: int, net_type: str, norm_cfg: dict,
                 act_cfg: dict, feat_size: Optional[tuple] = None,
                 rotation_mode: str = 'quaternion', init_cfg=None):
        super().__init__()
        if rotation_mode != 'ortho6d':
            raise NotImplementedError('quaternion branch needs kornia in the reference and is '
                                      'not used by the SCFlow config')
        feat_size = feat_size or {'Basic': (32, 32), 'Large': (64, 64)}[net_type]
        self.num_class = num_class
        self.rotation_out_channels = 6
        convs, cin, size = [], in_channels, feat_size[0] * feat_size[1]
        for _ in range(3):
            convs.append(ConvBlock(cin, 128, 3, stride=2, padding=1, act_cfg=act_cfg,
                                   norm_cfg=norm_cfg))
            cin, size = 128, int(size / 4)
        self.conv_layers = nn.Sequential(*convs)
        self.fc_layers = nn.Sequential(nn.Sequential(nn.Linear(128 * size, 1024), nn.ReLU()),
                                       nn.Sequential(nn.Linear(1024, 256), nn.ReLU()))
        self.rotation_pred = nn.Linear(256, 6 * num_class)
        self.translation_pred = nn.Linear(256, 3 * num_class)
        # reference label selection uses label[0] for the whole batch (pose_head.py:209-210,
        # SURVEY.md 8 a8); label_mode=1 selects per sample instead (what index_select was meant to do;
        # oracle.multiclass_pose_head(label_mode=1), tests/test_gpu_refiner.py::test_label_mode_per_sample).
        self.label_mode = 0
        # fc1 / fc2 / heads as split-K MFMA GEMMs with the last GroupNorm folded in (scf_fc_splitk); False = one
        # scf_linear launch per layer after a separate GroupNorm (A/B measurements, parity tests)
        self.fused_fc = True
        self.init_weights()

    def init_weights(self):
        """pose_head.py:187-198: zero heads, identity ortho6d bias."""
        nn.init.zeros_(self.translation_pred.weight)
        nn.init.zeros_(self.translation_pred.bias)
        nn.init.zeros_(self.rotation_pred.weight)
        with torch.no_grad():
            self.rotation_pred.bias.copy_(torch.tensor([1., 0., 0., 0., 1., 0.] * self.num_class))

    def fc_plan(self) -> Tuple[int, int]:
        """(K-slices of fc1, of fc2) when the fully connected tail fits ``scf_fc_splitk`` (three launches: the last
        GroupNorm + ReLU folded into fc1's load, every weight read once per batch), else (0, 0): plain
        ``scf_linear`` launches."""
        fc1, fc2, last = self.fc_layers[0][0], self.fc_layers[1][0], self.conv_layers[2]
        s1, s2 = ops.fc_slices(fc1.in_features), ops.fc_slices(fc2.in_features)
        ok = (s1 > 0 and s2 > 0 and ops.fc_slices(fc2.out_features) == 1 and last.groups is not None
              and last.act == ACT_RELU and self.fused_fc)
        if ok:      # a K-slice of fc1 must hold whole normalisation groups
            gsz = fc1.in_features // last.groups
            ok = fc1.in_features % last.groups == 0 and gsz % 2 == 0 and (fc1.in_features // s1) % gsz == 0
        return (s1, s2) if ok else (0, 0)

    def features(self, x0: Tensor, x1: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        x = self.conv_layers[0](x0, x1)
        x = self.conv_layers[1](x)
        s1, s2 = self.fc_plan()
        fc1, fc2 = self.fc_layers[0][0], self.fc_layers[1][0]
        if s1:
            last = self.conv_layers[2]
            # GroupNorm + ReLU: applied by fc1's operand load, which also adds the partial tensors of a K-sliced launch
            ks = ops.conv_kslices_for(last.packed, x)
            y = ops.conv2d(last.packed, x, kslices=ks)
            hw = y.shape[-2] * y.shape[-1]
            feat = y.shape[-3] * hw
            if feat != fc1.in_features:
                raise _lib_error(f'pose head expects {fc1.in_features} features, the maps give {feat}')
            yv = y.view(ks, x.shape[0], feat) if ks > 1 else y.view(x.shape[0], feat)
            p1 = ops.fc_splitk(yv, fc1.weight, gn=(last.groups, hw, last.gn.weight, last.gn.bias, last.gn.eps),
                               slices=s1)
            p2 = ops.fc_splitk(p1, fc2.weight, x_bias=fc1.bias, x_relu=True, slices=s2)
            return ops.fc_splitk(p2, self.rotation_pred.weight, self.rotation_pred.bias, x_bias=fc2.bias, x_relu=True,
                                 weight2=self.translation_pred.weight, bias2=self.translation_pred.bias)
        x = self.conv_layers[2](x)
        x = x.view(x.shape[0], -1)
        for fc in self.fc_layers:
            x = ops.linear(x, fc[0].weight, fc[0].bias, ACT_RELU)
        rot_all, trans_all = ops.linear_pair(x, self.rotation_pred.weight, self.rotation_pred.bias,
                                             self.translation_pred.weight, self.translation_pred.bias)
        return rot_all, trans_all

    def forward(self, x: Tensor, label: Tensor) -> Tuple[Tensor, Tensor]:
        """pose_head.py:201-211 -> (delta rotation (N,6), delta translation (N,3))."""
        rot_all, trans_all = self.features(x)
        n = x.shape[0]
        eye = torch.eye(3, device=x.device).repeat(n, 1, 1)
        one = torch.ones((n, 3), device=x.device)
        d_rot, d_trans, _, _ = ops.pose_update(rot_all, trans_all, label, self.num_class, eye, one,
                                               self.label_mode)
        return d_rot, d_trans


@DECODERS.register_module()
class SCFlowDecoder(HipModule):
    """decoder/scflow_decoder.py:18-251."""
    _h_channels = {'Basic': 128, 'Small': 96}
    _cxt_channels = {'Basic': 128, 'Small': 64}

    def __init__(self, net_type: str, num_levels: int, radius: int, iters: int, detach_flow: bool,
                 detach_mask: bool, detach_pose: bool, mask_flow: bool, mask_corr: bool,
                 pose_head_cfg: dict, depth_transform: str = 'exp',
                 detach_depth_for_xy: bool = False,
                 corr_lookup_cfg: dict = dict(align_corners=True), gru_type: str = 'SeqConv',
                 feat_channels: Union[int, Sequence[int]] = 256, conv_cfg=None, norm_cfg=None,
                 act_cfg=None) -> None:
        super().__init__()
        if net_type != 'Basic':
            raise NotImplementedError("SCFlowDecoder: net_type='Basic'")
        self.mask_flow, self.mask_corr = bool(mask_flow), bool(mask_corr)      # scflow_decoder.py:199-205
        # pose.py:137-141: 'exp' -> t_z / exp(d_z); ANY other value takes the reference's else branch t_z * (d_z + 1)
        self.depth_transform = depth_transform
        self.detach_depth_for_xy = detach_depth_for_xy      # autograd only (pose.py:142-147): no effect at inference
        self.net_type, self.num_levels, self.radius, self.iters = net_type, num_levels, radius, iters
        self.h_channels = self._h_channels[net_type]
        self.cxt_channels = self._cxt_channels[net_type]
        self.corr_block = CorrelationPyramid(num_levels)
        cl = dict(corr_lookup_cfg)
        cl.pop('type', None)
        cl['radius'] = radius
        self.corr_lookup = CorrLookup(**cl)
        self.encoder = MotionEncoder(num_levels, radius, net_type, conv_cfg, norm_cfg, act_cfg)
        self.gru = ConvGRU(self.h_channels, 126 + 2 + self.cxt_channels, gru_type)
        self.pose_pred = build_head(pose_head_cfg)
        fc = [256]    # scflow_decoder.py:73-74: the tuple check always yields [feat_channels]
        self.flow_pred = XHead(self.h_channels, fc, 2, x='flow')
        self.mask_pred = XHead(self.h_channels, fc, 1, x='mask')
        self.delta_flow_encoder = nn.Sequential(ConvBlock(2, 128, 7, padding=3, act_cfg=act_cfg),
                                                ConvBlock(128, 64, 3, padding=1, act_cfg=act_cfg))
        self.mask_encoder = nn.Sequential(ConvBlock(1, 64, 3, padding=1, act_cfg=act_cfg),
                                          ConvBlock(64, 32, 3, padding=1, act_cfg=act_cfg))
        self.tiled_pyramid = True     # decoder-internal pyramid layout (see _pyramid_layout)
        self.hoist_context = True     # GRU: convolve the (iteration-invariant) context channels once per pair
        # the launch sequence of an iteration issued by ONE C call (scf_scflow_iteration) instead of
        # ~33 Python-sequenced ones: same kernels, same order, same bits; False = sequence from here
        self.c_iteration = True

    def pose_flags(self) -> int:
        """the ``label_mode`` bit set of ``scf_pose_update`` (scflow_hip.h: SCF_POSE_*)."""
        return (self.pose_pred.label_mode & 1) | (0 if self.depth_transform == 'exp' else 2)

    def _pack_sources(self):
        a, b = self.flow_pred.layers[0].conv, self.mask_pred.layers[0].conv
        return [a.weight, a.bias, b.weight, b.bias]

    def _pack(self):
        # the two XHead hidden layers read the same h: one 512-row convolution
        a, b = self.flow_pred.layers[0].conv, self.mask_pred.layers[0].conv
        return PackedConv.from_weight(torch.cat([a.weight, b.weight], 0),
                                      torch.cat([a.bias, b.bias], 0), 1, 1)

    def forward(self, feat_render: Tensor, feat_real: Tensor, h_feat: Tensor, cxt_feat: Tensor,
                ref_rotation: Tensor, ref_translation: Tensor, depth: Tensor, internel_k: Tensor,
                label: Tensor, init_flow: Tensor, invalid_flow_num: float,
                _consume_state: bool = False):
        """scflow_decoder.py:150-251 (inference).  Returns the reference's 7-tuple of
        per-iteration lists."""
        hc, cc = self.h_channels, self.cxt_channels
        n, H, W = depth.shape
        scale = 2 ** (self.num_levels - 1)
        h, w = H // scale, W // scale
        dev = depth.device
        f32 = dict(dtype=torch.float32, device=dev)

        tiled = _pyramid_layout(feat_render, self.radius, self.num_levels, self.tiled_pyramid)
        pyramid = self.corr_block(feat_render, feat_real, tiled_levels=tiled)      # :172
        # GRU buffer [h | cxt | motion(126) | flow(2)]: the caller's own buffer only when the
        # refiner says it may be consumed (_consume_state); the public forward never mutates
        # its inputs (the reference decoder does not either)
        hx = _as_gru_buffer(h_feat, cxt_feat, hc + cc + 128, _consume_state)
        rot, trans = ref_rotation.contiguous(), ref_translation.contiguous()
        rot0, trans0 = rot, trans
        flow = init_flow
        outs = ([], [], [], [], [], [], [])
        dm = torch.empty((n, 96, h, w), **f32)
        heads = torch.empty((n, 512, h, w), **f32)
        # the context channels of hx never change: their part of the GRU convolutions, once
        ctx = self.gru.context_terms(hx[:, hc:hc + cc]) if self.hoist_context else None
        # small batches: independent branches side by side
        ov_flow, ov_mask, ov_up = (ops.small_work(n, H, W, b) for b in ('flow', 'mask', 'upsample'))
        if self.c_iteration and ops._CONV_EVENTS is None:
            return self._forward_c(pyramid, tiled, hx, ctx, rot0, trans0, depth, internel_k, label, init_flow,
                                   invalid_flow_num, tuple(ops.branch_mode(n, H, W, b) for b in ('flow', 'mask', 'upsample')))
        # occlusion mask of the previous iteration (ones before the first: the 1/8 bilinear
        # down-sampling of a ones map, :188-190), used only with mask_flow / mask_corr
        mask = ops.constant((n, 1, h, w), 1.0, dev) if (self.mask_flow or self.mask_corr) else None
        # scratch reused by every iteration (allocated once, before any fork point: the side branches
        # write into cf; z / rh are the GRU's gate buffers)
        cf = torch.empty((n, 256, h, w), **f32)
        zbuf = torch.empty((2, n, hc, h, w), **f32)
        for _ in range(self.iters):
            flow_lr = ops.resize_bilinear(flow, (h, w), mul=1.0 / scale)           # :196-197
            flow_in = ops.mul_mask(flow_lr, mask) if self.mask_flow else flow_lr   # :203-204
            fork = ops.fork_point() if ov_flow else None    # the motion encoder's flow branch starts here
            corr = self.corr_lookup(pyramid, flow_lr, tiled_levels=tiled)          # :198
            if self.mask_corr:
                ops.mul_mask(corr, mask, out=corr)                                 # :200-201
            self.encoder(corr, flow_in, out=hx[:, hc + cc:], overlap=ov_flow, cf=cf, fork=fork)   # :206
            hv = self.gru.forward_inplace(hx, ctx, cc, scratch=zbuf)               # :207-208
            ops.conv2d(self.packed, hv, out=heads, act=ACT_RELU)
            d_flow = self.flow_pred.predict(heads[:, :256])                        # :210
            mask = self.mask_pred.predict(heads[:, 256:], act=ACT_SIGMOID)         # :212-213
            br = ops.side_stream(ov_mask)
            with br:
                m1 = self.mask_encoder[0](mask)                                    # :217
                self.mask_encoder[1](m1, out=dm[:, 64:])
                del m1
            d1 = self.delta_flow_encoder[0](d_flow)                                # :216
            self.delta_flow_encoder[1](d1, out=dm[:, :64])
            br.join()
            # the two full-resolution outputs do not feed the pose head: side branch (outputs are
            # allocated here, on the main stream, because they escape the branch)
            flow_pred = torch.empty((n, 2, H, W), **f32)
            up_mask = torch.empty((n, 1, H, W), **f32)
            br = ops.side_stream(ov_up)
            with br:
                ops.resize_bilinear(flow_lr, (H, W), mul=float(scale), b=d_flow, out=flow_pred)  # :222-224
                ops.resize_bilinear(mask, (H, W), out=up_mask)                     # :226-227
            rot_all, trans_all = self.pose_pred.features(hv, dm)                   # :218-219
            d_rot, d_trans, rot, trans = ops.pose_update(                          # :230-236
                rot_all, trans_all, label, self.pose_pred.num_class, rot, trans,
                self.pose_flags())
            flow = ops.reproject_flow(depth, internel_k, rot0, trans0, rot, trans,  # :239-243
                                      invalid_flow_num)
            br.join()
            for lst, v in zip(outs, (flow, flow_pred, rot, trans, up_mask, d_rot, d_trans)):
                lst.append(v)
        return outs


def _scflow_forward_c(self, pyramid, tiled, hx, ctx, rot0, trans0, depth, internel_k, label, init_flow,
                      invalid_flow_num, overlap):
    """the refinement loop through ``scf_scflow_iteration``: every scratch buffer and convolution
    descriptor of an iteration is set up ONCE per pass; an iteration is then a handful of pointer
    updates and one C call.  Mirrors ``SCFlowDecoder.forward``'s Python-sequenced loop launch for
    launch (``tests/test_gpu_refiner.py::test_c_iteration_is_bit_identical``)."""
    from ._lib import ScflowIter
    import ctypes as C
    hc, cc = self.h_channels, self.cxt_channels
    n, H, W = depth.shape
    scale = 2 ** (self.num_levels - 1)
    h, w = H // scale, W // scale
    f32 = dict(dtype=torch.float32, device=depth.device)
    iters = self.iters
    E = lambda *shape: torch.empty(shape, **f32)
    enc, ph = self.encoder, self.pose_pred
    kch = self.num_levels * (2 * self.radius + 1) ** 2
    # ---- scratch of one iteration (reused by all of them) ----
    flow_lr, flow_m, corr = E(n, 2, h, w), (E(n, 2, h, w) if self.mask_flow else None), E(n, kch, h, w)
    f1, c1, cf = E(n, 128, h, w), E(n, 256, h, w), E(n, 256, h, w)
    zbuf, heads, dm = E(2, n, hc, h, w), E(n, 512, h, w), E(n, 96, h, w)
    d_flow, mask, m1, d1 = E(n, 2, h, w), E(n, 1, h, w), E(n, 64, h, w), E(n, 128, h, w)
    ones = ops.constant((n, 1, h, w), 1.0, depth.device) if (self.mask_flow or self.mask_corr) else None
    hv, xm = hx[:, :hc], hx[:, hc + cc:]
    # ---- outputs of all iterations: one buffer per kind, a view per iteration ----
    flows, fpreds, masks = E(iters, n, 2, H, W), E(iters, n, 2, H, W), E(iters, n, 1, H, W)
    rots, transs, drots, dtranss = E(iters, n, 3, 3), E(iters, n, 3), E(iters, n, 6), E(iters, n, 3)
    it = ScflowIter()
    it.struct_size = C.sizeof(ScflowIter)
    it.N, it.H, it.W, it.h, it.w = n, H, W, h, w
    it.L, it.radius, it.tiled_levels, it.corr_channels = len(pyramid), self.radius, int(tiled), kch
    for l, lv in enumerate(pyramid):
        it.levels[l] = lv.data_ptr()
    it.flow_lr, it.corr = flow_lr.data_ptr(), corr.data_ptr()
    it.mask_flow, it.mask_corr = int(self.mask_flow), int(self.mask_corr)
    it.flow_masked = None if flow_m is None else flow_m.data_ptr()
    keep = [pyramid, flow_lr, flow_m, corr, f1, c1, cf, zbuf, heads, dm, d_flow, mask, m1, d1, ones, hx, ctx]
    cd = lambda blk, *a, **k: ops.conv_desc(blk.packed, *a, act=blk.act, **k)[0]
    it.flow0 = cd(enc.flow_net[0], flow_lr, out=f1)
    it.flow1 = cd(enc.flow_net[1], f1, out=cf[:, 192:])
    it.corr0 = cd(enc.corr_net[0], corr, out=c1)
    it.corr1 = cd(enc.corr_net[1], c1, out=cf[:, :192])
    it.outn = cd(enc.out_net[0], cf, out=xm[:, :126])
    it.flow_copy_dst = xm[:, 126:128].data_ptr()
    it.hx, it.hx_nstride = hx.data_ptr(), hx.stride(0)
    it.Ch, it.Cc, it.Cx = hc, cc, hx.shape[1] - hc - cc
    if ctx is not None:
        packs = [(pk[1], pk[2]) for pk in self.gru._ctx_packs(cc)]
        for i, t in enumerate(ctx):
            it.ctx[i] = t.data_ptr()
        it.ctx_nstride = ctx[0].stride(0)
    else:
        packs = self.gru.packed
    arr = ops.gru_passes(packs)
    it.npass = len(packs)
    for i in range(len(packs)):
        it.gru[i] = arr[i]
    it.z, it.rh = zbuf[0].data_ptr(), zbuf[1].data_ptr()
    it.heads = ops.conv_desc(self.packed, hv, out=heads, act=ACT_RELU)[0]
    it.fpred = ops.conv_desc(self.flow_pred.packed, heads[:, :256], out=d_flow)[0]
    it.mpred = ops.conv_desc(self.mask_pred.packed, heads[:, 256:], out=mask, act=ACT_SIGMOID)[0]
    it.menc0 = cd(self.mask_encoder[0], mask, out=m1)
    it.menc1 = cd(self.mask_encoder[1], m1, out=dm[:, 64:])
    it.denc0 = cd(self.delta_flow_encoder[0], d_flow, out=d1)
    it.denc1 = cd(self.delta_flow_encoder[1], d1, out=dm[:, :64])
    # ---- pose head ----
    x0, x1 = hv, dm
    for i, blk in enumerate(ph.conv_layers):
        if blk.groups is None or blk.act != ACT_RELU:
            raise NotImplementedError('pose head: conv + GroupNorm + ReLU blocks')
        ks = ops.conv_kslices_for(blk.packed, x0, x1)
        d_, y = ops.conv_desc(blk.packed, x0, x1, kslices=ks)       # ks > 1: (ks, N, C, h, w) partial tensors
        g = torch.empty_like(y[0] if ks > 1 else y)
        it.pose[i] = d_
        it.gn[i].gamma, it.gn[i].beta, it.gn[i].out = blk.gn.weight.data_ptr(), blk.gn.bias.data_ptr(), g.data_ptr()
        it.gn[i].C, it.gn[i].HW, it.gn[i].G, it.gn[i].eps = y.shape[-3], y.shape[-2] * y.shape[-1], blk.groups, blk.gn.eps
        keep += [y, g]
        x0, x1 = g, None
    fc1, fc2 = ph.fc_layers[0][0], ph.fc_layers[1][0]
    if fc1.in_features != x0[0].numel():
        raise _lib_error(f'pose head expects {fc1.in_features} features, the maps give {x0[0].numel()}')
    s1, s2 = ph.fc_plan()
    it.fc_fused, it.fc1_slices, it.fc2_slices = int(s1 > 0), max(s1, 1), max(s2, 1)
    y1, y2 = E(max(s1, 1), n, fc1.out_features), E(max(s2, 1), n, fc2.out_features)
    ra, ta = E(n, ph.rotation_pred.out_features), E(n, ph.translation_pred.out_features)
    keep += [y1, y2, ra, ta]
    P = lambda t: None if t is None else t.data_ptr()
    it.fc1_w, it.fc1_b, it.fc1_out, it.fc1_K, it.fc1_O = P(fc1.weight), P(fc1.bias), y1.data_ptr(), fc1.in_features, fc1.out_features
    it.fc2_w, it.fc2_b, it.fc2_out, it.fc2_O = P(fc2.weight), P(fc2.bias), y2.data_ptr(), fc2.out_features
    it.rot_w, it.rot_b, it.rot_all, it.rot_O = P(ph.rotation_pred.weight), P(ph.rotation_pred.bias), ra.data_ptr(), ra.shape[1]
    it.trans_w, it.trans_b, it.trans_all, it.trans_O = (P(ph.translation_pred.weight), P(ph.translation_pred.bias),
                                                        ta.data_ptr(), ta.shape[1])
    if not label.is_cuda or label.dtype != torch.int64 or not label.is_contiguous():
        raise _lib_error('label must be a contiguous int64 GPU tensor')
    it.label, it.num_class, it.label_mode = label.data_ptr(), ph.num_class, self.pose_flags()
    for name, t in (('depth', depth), ('internel_k', internel_k), ('ref_rotation', rot0), ('ref_translation', trans0),
                    ('init_flow', init_flow)):
        ops._dense(t, name)
    it.depth, it.K, it.R0, it.t0 = depth.data_ptr(), internel_k.data_ptr(), rot0.data_ptr(), trans0.data_ptr()
    it.invalid_flow_num = float(invalid_flow_num)
    it.overlap_flow, it.overlap_mask, it.overlap_up = (int(b) for b in overlap)      # 0 in order / 1 side stream / 2 merged launches
    it.side_stream = ops.side_stream_handle() if any(int(b) == 1 for b in overlap) else None
    outs = ([], [], [], [], [], [], [])
    flow, rot, trans = init_flow, rot0, trans0
    for i in range(iters):
        it.flow_in, it.R_in, it.t_in = flow.data_ptr(), rot.data_ptr(), trans.data_ptr()
        it.mask_prev = None if ones is None else (ones if i == 0 else mask).data_ptr()
        flow, rot, trans = flows[i], rots[i], transs[i]
        it.flow_out, it.flow_pred, it.mask_up = flow.data_ptr(), fpreds[i].data_ptr(), masks[i].data_ptr()
        it.R_out, it.t_out, it.d_rot, it.d_trans = rot.data_ptr(), trans.data_ptr(), drots[i].data_ptr(), dtranss[i].data_ptr()
        ops.scflow_iteration(it)
        for lst, v in zip(outs, (flow, fpreds[i], rot, trans, masks[i], drots[i], dtranss[i])):
            lst.append(v)
    if any(overlap):          # the scratch is freed on return: the side stream's last reads are joined already
        pass
    del keep
    return outs


def _lib_error(msg):
    from ._lib import ScflowHipError
    return ScflowHipError(msg)


SCFlowDecoder._forward_c = _scflow_forward_c


class _RAFTDecoderBase(HipModule):
    """shared part of RAFTDecoder / RAFTDecoderMask ('Basic'): correlation pyramid, lookup,
    motion encoder, SepConvGRU, flow head, 576-channel convex-up-sampling mask head."""
    _h_channels = {'Basic': 128, 'Small': 96}
    _cxt_channels = {'Basic': 128, 'Small': 64}

    def __init__(self, net_type: str, num_levels: int, radius: int, iters: int,
                 corr_lookup_cfg: dict = dict(type='CorrLookup', align_corners=True),
                 gru_type: str = 'SeqConv', feat_channels: Union[int, Sequence[int]] = 256,
                 mask_channels: int = 64, convex_unsample_flow: bool = True, conv_cfg=None,
                 norm_cfg=None, act_cfg=None) -> None:
        super().__init__()
        if net_type != 'Basic':
            raise NotImplementedError("RAFT decoders: net_type='Basic'")
        self.net_type, self.num_levels, self.radius, self.iters = net_type, num_levels, radius, iters
        self.h_channels = self._h_channels[net_type]
        self.cxt_channels = self._cxt_channels[net_type]
        self.mask_channels = mask_channels * (2 * radius + 1)      # raft_decoder.py:355
        self.corr_block = CorrelationPyramid(num_levels)
        cl = dict(corr_lookup_cfg)
        cl.pop('type', None)
        cl['radius'] = radius
        self.corr_lookup = CorrLookup(**cl)
        self.encoder = MotionEncoder(num_levels, radius, net_type, conv_cfg, norm_cfg, act_cfg)
        self.gru = ConvGRU(self.h_channels, 126 + 2 + self.cxt_channels, gru_type)
        self.flow_pred = XHead(self.h_channels, [256], 2, x='flow')
        self.mask_pred = XHead(self.h_channels, [256], self.mask_channels, x='mask')
        self.convex_upsample_flow = convex_unsample_flow
        self.tiled_pyramid = True     # decoder-internal pyramid layout (see _pyramid_layout)
        self.hoist_context = True     # GRU: convolve the (iteration-invariant) context channels once per pair
        if self.mask_channels != 9 * (2 ** (num_levels - 1)) ** 2:
            raise NotImplementedError('convex up-sampling kernel: 9 x 8 x 8 mask (radius 4, 4 levels)')

    def _context(self, hx):
        hc, cc = self.h_channels, self.cxt_channels
        return self.gru.context_terms(hx[:, hc:hc + cc]) if self.hoist_context else None

    def _step(self, pyramid, flow, hx, tiled=0, ctx=None):
        """one update: lookup, motion encoder, GRU (in place in hx), flow += delta."""
        hc, cc = self.h_channels, self.cxt_channels
        corr = self.corr_lookup(pyramid, flow, tiled_levels=tiled)
        self.encoder(corr, flow, out=hx[:, hc + cc:])
        hv = self.gru.forward_inplace(hx, ctx, cc)
        d_flow = self.flow_pred(hv)
        n, _, h, w = flow.shape
        return hv, ops.resize_bilinear(flow, (h, w), b=d_flow)       # flow + delta (same size)

    def _upsample(self, x: Tensor, mask: Optional[Tensor], mul: float) -> Tensor:
        scale = 2 ** (self.num_levels - 1)
        n, c, h, w = x.shape
        if mask is None:                         # raft_decoder.py:397-400
            return ops.resize_bilinear(x, (scale * h, scale * w), mul=mul)
        return ops.convex_upsample(x, mask, scale, x_mul=mul, mask_mul=0.25)


@DECODERS.register_module()
class RAFTDecoder(_RAFTDecoderBase):
    """decoder/raft_decoder.py:299-457."""

    def forward(self, feat1: Tensor, feat2: Tensor, flow: Tensor, h_feat: Tensor,
                cxt_feat: Tensor, _consume_state: bool = False) -> List[Tensor]:
        tiled = _pyramid_layout(feat1, self.radius, self.num_levels, self.tiled_pyramid)
        pyramid = self.corr_block(feat1, feat2, tiled_levels=tiled)
        hx = _as_gru_buffer(h_feat, cxt_feat, self.h_channels + self.cxt_channels + 128,
                            _consume_state)
        scale = float(2 ** (self.num_levels - 1))
        flow = flow.contiguous()
        outs = []
        ctx = self._context(hx)
        for _ in range(self.iters):
            hv, flow = self._step(pyramid, flow, hx, tiled, ctx)
            mask = self.mask_pred(hv) if self.convex_upsample_flow else None   # 0.25 folded in
            outs.append(self._upsample(flow, mask, scale))
        return outs


@DECODERS.register_module()
class RAFTDecoderMask(_RAFTDecoderBase):
    """decoder/raft_decoder_mask.py:21-208: RAFTDecoder + occlusion head."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.occlusion_pred = XHead(self.h_channels, [256], 1, x='mask')

    def forward(self, feat1: Tensor, feat2: Tensor, flow: Tensor, h_feat: Tensor,
                cxt_feat: Tensor, _consume_state: bool = False):
        tiled = _pyramid_layout(feat1, self.radius, self.num_levels, self.tiled_pyramid)
        pyramid = self.corr_block(feat1, feat2, tiled_levels=tiled)
        hx = _as_gru_buffer(h_feat, cxt_feat, self.h_channels + self.cxt_channels + 128,
                            _consume_state)
        scale = float(2 ** (self.num_levels - 1))
        flow = flow.contiguous()
        flows, occs = [], []
        ctx = self._context(hx)
        for _ in range(self.iters):
            hv, flow = self._step(pyramid, flow, hx, tiled, ctx)
            occ = self.occlusion_pred.predict(self.occlusion_pred.layers[0](hv), act=ACT_SIGMOID)
            mask = self.mask_pred(hv) if self.convex_upsample_flow else None
            flows.append(self._upsample(flow, mask, scale))
            occs.append(self._upsample(occ, mask, 1.0))
        return flows, occs


def _as_gru_buffer(h_feat: Tensor, cxt_feat: Tensor, total: int, consume: bool = False) -> Tensor:
    """[h | cxt | ...] buffer of ``total`` channels.  The GRU updates h in place and the motion
    features land next to cxt, so the buffer is private to one decoder run: a fresh copy of the
    inputs, unless the caller hands its own buffer over (``consume``: the refiner's get_pose /
    get_flow, whose extract_feat output is used exactly once) and h_feat / cxt_feat already are
    adjacent channel slices of such a buffer (zero-copy)."""
    n, hc, h, w = h_feat.shape
    cc = cxt_feat.shape[1]
    base = h_feat._base
    if (consume and base is not None and base is cxt_feat._base and base.dim() == 4 and base.is_contiguous()
            and tuple(base.shape) == (n, total, h, w)
            and h_feat.data_ptr() == base.data_ptr()
            and cxt_feat.data_ptr() == base.data_ptr() + hc * h * w * 4
            and h_feat.stride() == base.stride() and cxt_feat.stride() == base.stride()):
        return base
    hx = torch.empty((n, total, h, w), dtype=torch.float32, device=h_feat.device)
    ops.copy_channels(h_feat, hx[:, :hc])
    ops.copy_channels(cxt_feat, hx[:, hc:hc + cc])
    return hx
