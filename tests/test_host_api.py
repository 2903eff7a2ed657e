"""CPU: the host-side mirror of the reference interface -- registries, config
compatibility, state_dict key layout, cache invalidation, weight packing."""
import json
import os
import runpy

import pytest
import torch

import scflow_amd
from scflow_amd import ops
from scflow_amd.registry import DECODERS, ENCODERS, HEAD, REFINERS, build_from_cfg


def test_registries_hold_the_reference_class_names():
    assert 'SCFlowRefiner' in REFINERS
    assert 'RAFTEncoder' in ENCODERS
    assert 'SCFlowDecoder' in DECODERS
    assert 'MultiClassPoseHead' in HEAD
    with pytest.raises(KeyError):
        build_from_cfg(dict(type='Nope'), REFINERS)
    with pytest.raises(KeyError):
        build_from_cfg(dict(foo=1), REFINERS)


def test_state_dict_layout_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, 'state_dict_keys.json')))
    model = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
    sd = model.state_dict()
    assert len(sd) == ref['num_tensors'] == 226
    assert {k: list(v.shape) for k, v in sd.items()} == ref['shapes']
    uniq = sum(v.numel() for k, v in sd.items() if not k.startswith('real_encoder.'))
    assert uniq == ref['num_params_unique']
    # shared encoder: one module under two names (base_refiner.py:36-39)
    assert model.real_encoder is model.render_encoder
    sd2 = scflow_amd.fill_state_dict({k: v.shape for k, v in sd.items()}, seed=0)
    model.load_state_dict(sd2, strict=True)


def test_reference_config_file_builds_unchanged():
    path = '/root/reference/configs/refine_models/scflow.py'
    if not os.path.exists(path):
        pytest.skip('reference checkout not present (GPU box)')
    cfg = runpy.run_path(path)['model']
    model = scflow_amd.build_refiner(cfg)
    assert model.decoder.iters == 8 and model.test_iter_num == 8
    assert model.decoder.corr_lookup.r == 4 and model.decoder.num_levels == 4
    ours = scflow_amd.scflow_model_cfg()
    for key in ('cxt_channels', 'h_channels', 'seperate_encoder', 'max_flow'):
        assert ours[key] == cfg[key]
    for key in ('net_type', 'num_levels', 'radius', 'iters', 'mask_flow', 'mask_corr',
                'gru_type', 'pose_head_cfg'):
        assert ours['decoder'][key] == cfg['decoder'][key], key


def test_pose_head_zero_init_like_reference():
    head = build_from_cfg(scflow_amd.scflow_model_cfg()['decoder']['pose_head_cfg'], HEAD)
    assert float(head.translation_pred.weight.abs().max()) == 0.0
    assert float(head.rotation_pred.weight.abs().max()) == 0.0
    assert head.rotation_pred.bias[:6].tolist() == [1., 0., 0., 0., 1., 0.]


def test_pack_conv_weight_layout():
    w = torch.arange(5 * 3 * 2 * 2, dtype=torch.float32).reshape(5, 3, 2, 2)
    wp, mld = ops.pack_conv_weight(w, kc=2)
    assert mld == 32 and wp.shape == (2 * 4 * 2, 32)         # 2 chunks x 4 taps x KC=2
    for co in range(5):
        for ci in range(3):
            for t in range(4):
                row = ((ci // 2) * 4 + t) * 2 + ci % 2
                assert wp[row, co] == w[co, ci, t // 2, t % 2]
    assert float(wp[:, 5:].abs().max()) == 0.0
    pad_rows = [((1 * 4 + t) * 2 + 1) for t in range(4)]      # channel 3 does not exist
    assert float(wp[pad_rows].abs().max()) == 0.0
    assert ops.choose_kc(3, 7, 7) == 2 and ops.choose_kc(64, 3, 3) == 8 and ops.choose_kc(2, 7, 7) == 2
    assert ops.choose_kc(324, 1, 1) == 32 and ops.choose_kc(64, 1, 1, 2) == 8


def test_packed_cache_invalidation():
    enc = build_from_cfg(scflow_amd.scflow_model_cfg()['encoder'], ENCODERS)
    p1 = enc.packed
    assert enc.packed is p1
    enc.load_state_dict(enc.state_dict())
    assert enc.__dict__['_packed'] is None
    blk = enc.res_layer1[0]
    _ = blk.packed
    enc.float()                                  # any _apply drops caches of all submodules
    assert blk.__dict__['_packed'] is None


def test_unsupported_options_fail_loudly():
    cfg = scflow_amd.scflow_model_cfg()
    cfg['decoder']['net_type'] = 'Small'
    with pytest.raises(NotImplementedError):
        scflow_amd.build_refiner(cfg)
    cfg = scflow_amd.scflow_model_cfg()
    cfg['decoder']['mask_corr'] = cfg['decoder']['mask_flow'] = True       # built since round 3
    assert scflow_amd.build_refiner(cfg).decoder.mask_corr
    model = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(scflow_amd._lib.ScflowHipError):
        model.extract_feat(x, x)               # CPU tensors: no fallback
    # r6: options that are built now construct (pose.py:137-141, base_refiner.py:33-35, pose_head.py:121-162,
    # raft_decoder.py:178-181); what still needs the renderer / kornia refuses with the reason
    cfg = scflow_amd.scflow_model_cfg()
    cfg['seperate_encoder'] = True
    cfg['decoder'].update(depth_transform='linear', gru_type='Conv', radius=3)
    cfg['decoder']['pose_head_cfg']['feat_size'] = (64, 80)
    m = scflow_amd.build_refiner(cfg)
    assert m.render_encoder is not m.real_encoder and m.decoder.pose_flags() == 2
    assert m.decoder.pose_pred.fc_layers[0][0].in_features == 128 * 80
    m.decoder.pose_pred.label_mode = 1
    assert m.decoder.pose_flags() == 3
    cfg = scflow_amd.scflow_model_cfg()
    cfg['test_cfg'] = dict(iters=8, cycles=2)                  # re-rendering between cycles: base_refiner.py:250-258
    with pytest.raises(NotImplementedError, match='cycles'):
        scflow_amd.build_refiner(cfg).forward(dict(), None)
    cfg = scflow_amd.scflow_model_cfg()
    cfg['decoder']['pose_head_cfg']['rotation_mode'] = 'quaternion'       # kornia branch, pose.py:132-133
    with pytest.raises(NotImplementedError):
        scflow_amd.build_refiner(cfg)


def test_checkpoint_ingestion(tmp_path, golden_dir):
    """mmflow -> SCFlow key mapping (tools/mmflow_ckpt_converter.py:30-35) and mmcv-style files."""
    from scflow_amd.checkpoint import convert_mmflow_state_dict, load_checkpoint
    model = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
    sd = scflow_amd.fill_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed=2)
    # an "mmflow" checkpoint: one encoder.* copy, DDP prefix, mmcv file layout
    mm = {('module.' + k.replace('render_encoder', 'encoder')): v for k, v in sd.items()
          if not k.startswith('real_encoder.')}
    conv = convert_mmflow_state_dict({'encoder.conv1.weight': 1, 'decoder.gru.x': 2, 'context.a': 3})
    assert set(conv) == {'real_encoder.conv1.weight', 'render_encoder.conv1.weight',
                         'decoder.gru.x', 'context.a'}
    path = os.path.join(tmp_path, 'ckpt.pth')
    torch.save({'state_dict': mm, 'meta': {}}, path)
    _ = model.render_encoder.packed
    missing, unexpected, mismatched = load_checkpoint(model, path, strict=True, from_mmflow=True)
    assert not missing and not unexpected and not mismatched
    assert model.render_encoder.__dict__['_packed'] is None       # kernel-layout cache dropped
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_checkpoint_ingestion_real_raft_layout():
    """the checkpoint the reference config's init_cfg points at is an mmflow RAFT one:
    ONE encoder, a 576-channel convex-up-sampling ``mask_pred`` head under the key SCFlow uses
    for its 1-channel mask head, no pose head / delta-flow / mask encoders.  mmcv's loader (the
    reference's train.py / test.py) warns and skips; so does load_checkpoint(from_mmflow=True)."""
    from scflow_amd.checkpoint import load_checkpoint
    raft = scflow_amd.build_refiner(dict(
        type='RAFTRefinerFlow', cxt_channels=128, h_channels=128, seperate_encoder=False,
        encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                     norm_cfg=dict(type='IN')),
        cxt_encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                         norm_cfg=dict(type='BN')),
        decoder=dict(type='RAFTDecoder', net_type='Basic', num_levels=4, radius=4, iters=12,
                     corr_lookup_cfg=dict(align_corners=True), gru_type='SeqConv',
                     act_cfg=dict(type='ReLU'))))
    rsd = scflow_amd.fill_state_dict({k: v.shape for k, v in raft.state_dict().items()}, seed=4)
    mm = {k.replace('render_encoder', 'encoder'): v for k, v in rsd.items()
          if not k.startswith('real_encoder.')}
    assert mm['decoder.mask_pred.predict_layer.weight'].shape[0] == 576
    model = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    with pytest.raises(RuntimeError):                      # torch raises on the size mismatch ...
        load_checkpoint(model, mm, strict=True, from_mmflow=True)
    missing, unexpected, mismatched = load_checkpoint(model, mm, from_mmflow=True)   # ... mmcv-like: skip
    assert sorted(k for k, _, _ in mismatched) == ['decoder.mask_pred.predict_layer.bias',
                                                   'decoder.mask_pred.predict_layer.weight']
    assert not unexpected
    assert all(k.startswith(('decoder.pose_pred.', 'decoder.delta_flow_encoder.', 'decoder.mask_encoder.',
                             'decoder.mask_pred.predict_layer.')) for k in missing)
    after = model.state_dict()
    for k, v in after.items():
        if k in missing:
            assert torch.equal(v, before[k]), k             # untouched
        else:
            src = k.replace('real_encoder', 'render_encoder')
            assert torch.equal(v, rsd[src]), k


# ------------------------------------------------------------ weight packings (host logic)
def test_pack_conv_weight_a4_layout():
    """[chunk][tap][g][h][Mld][4]: channel chunk*8G + 8g + 2s + h at float s (conv_dma.hip)."""
    from scflow_amd import ops
    g = torch.Generator().manual_seed(3)
    for cout, cin, kh, kw, G in ((40, 20, 3, 3, 1), (64, 48, 1, 5, 2), (33, 70, 3, 3, 1)):
        w = torch.randn((cout, cin, kh, kw), generator=g)
        flat, mld = ops.pack_conv_weight_a4(w, G)
        t, kc = kh * kw, 8 * G
        nchunk = (cin + kc - 1) // kc
        assert mld == (cout + 31) // 32 * 32
        p = flat.reshape(nchunk, t, G, 2, mld, 4)
        for (m, c, tap) in ((0, 0, 0), (cout - 1, cin - 1, t - 1), (7, 13, t // 2), (cout // 2, 9, 1 % t)):
            chunk, r = divmod(c, kc)
            gg, r = divmod(r, 8)
            s, h = divmod(r, 2)
            assert p[chunk, tap, gg, h, m, s] == w[m, c, tap // kw, tap % kw]
        # padded channels / couts are zero
        assert float(p[:, :, :, :, cout:, :].abs().sum()) == 0.0
        total = float(p.abs().sum())
        assert abs(total - float(w.abs().sum())) <= 1e-3 * total


def test_pack_conv_weight_thin_layout_and_eligibility():
    from scflow_amd import ops
    w = torch.randn((2, 40, 3, 3), generator=torch.Generator().manual_seed(4))
    flat = ops.pack_conv_weight_thin(w)
    assert flat.numel() % 4 == 0
    p = flat[:40 * 9 * 2].reshape(40, 9, 2)
    assert p[17, 5, 1] == w[1, 17, 1, 2] and p[0, 0, 0] == w[0, 0, 0, 0]
    w3 = torch.randn((3, 32, 1, 1))
    assert ops.pack_conv_weight_thin(w3)[:32 * 4].reshape(32, 1, 4)[:, 0, 3].abs().sum() == 0   # CO = 4
    # which layers get the LDS-DMA packing
    assert ops.choose_a4_groups(64, 3, 3, 1) == 1 and ops.choose_a4_groups(384, 1, 5, 1) == 2
    assert ops.choose_a4_groups(224, 3, 3, 2) == 1
    assert ops.choose_a4_groups(324, 1, 1, 1) == 4 and ops.choose_a4_groups(3, 7, 7, 2) == 0
    assert ops.choose_a4_groups(64, 1, 1, 2) == 4 and ops.choose_a4_groups(32, 1, 1, 1) == 0    # 1x1: >= 64 channels (stride 2: dilated gather)
    # ... and the small-grid variant with bigger chunks (dense 1x1 layers join in)
    assert ops.choose_a4s_groups(324, 1, 1, 1) == 4 and ops.choose_a4s_groups(384, 5, 1, 1) == 4
    assert ops.choose_a4s_groups(128, 3, 3, 2) == 2 and ops.choose_a4s_groups(8, 3, 3, 1) == 0
    assert ops.choose_a4s_groups(3, 7, 7, 2) == 0
    assert ops.choose_a4_groups(2, 3, 3, 1) == 0


# ------------------------------------------------------------ property tests (hypothesis)
def test_shard_range_partitions_any_job():
    from hypothesis import given, settings, strategies as st
    from scflow_amd.dist import shard_range

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 5000), st.integers(1, 64))
    def check(total, world):
        edges = [shard_range(total, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        for (a, b), (c, d) in zip(edges, edges[1:]):
            assert b == c and a <= b
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1

    check()


def test_weight_packings_are_permutations_of_the_weights():
    """every packing holds each weight exactly once (plus zero padding): same multiset of values."""
    from hypothesis import given, settings, strategies as st
    from scflow_amd import ops

    @settings(max_examples=25, deadline=None)
    @given(st.integers(1, 70), st.integers(8, 70), st.sampled_from([(3, 3), (1, 5), (5, 1), (1, 1)]),
           st.integers(0, 2 ** 31 - 1))
    def check(cout, cin, k, seed):
        w = torch.randn((cout, cin, *k), generator=torch.Generator().manual_seed(seed))
        ref = torch.sort(w.reshape(-1)).values
        for kc in (2, 8, 32):
            p, _ = ops.pack_conv_weight(w, kc)
            nz = p.reshape(-1)
            assert torch.equal(torch.sort(nz[nz != 0]).values, ref[ref != 0])
        for g in (1, 2, 4):
            p, _ = ops.pack_conv_weight_a4(w, g)
            assert torch.equal(torch.sort(p[p != 0]).values, ref[ref != 0])
        h16 = ops.pack_conv_weight_f16x3(w).float()
        hi, lo = h16[:, 0], h16[:, 1]
        # hi + lo * 2^-11 reproduces every weight to ~22 bits
        tot = float((hi + lo / 2048.0).abs().sum())
        assert abs(tot - float(w.abs().sum())) <= 2e-6 * float(w.abs().sum()) + 1e-6

    check()


def test_packed_weights_follow_every_kind_of_weight_change():
    """the kernel-layout copy is keyed on (data_ptr, _version) of its source tensors,
    so in-place edits and a PARENT module's load_state_dict (mmcv's load_checkpoint route, which
    never calls HipModule.load_state_dict) both invalidate it.  CPU: packing is plain torch."""
    from scflow_amd.modules import ConvBlock
    blk = ConvBlock(16, 8, 3, padding=1)
    p0 = blk.packed
    assert blk.packed is p0                                  # cached while nothing changes
    with torch.no_grad():
        blk.conv.weight.mul_(2.0)                            # in place: _version bumps
    p1 = blk.packed
    assert p1 is not p0 and torch.allclose(p1.wp, 2.0 * p0.wp)
    parent = torch.nn.Sequential(blk)                        # a wrapper's loader recurses past the override
    parent.load_state_dict({'0.conv.weight': torch.ones(8, 16, 3, 3), '0.conv.bias': torch.zeros(8)})
    p2 = blk.packed
    assert p2 is not p1 and float(p2.wp.max()) == 1.0 and float(p2.bias.abs().max()) == 0.0
    torch.nn.init.constant_(blk.conv.bias, 0.25)
    assert float(blk.packed.bias.min()) == 0.25
    # an edit through .data does NOT bump the parameter's version (the key cannot see
    # it): that route is documented as needing invalidate_packed()
    p3 = blk.packed
    v = blk.conv.weight._version
    blk.conv.weight.data.copy_(torch.full((8, 16, 3, 3), 2.0))
    assert blk.conv.weight._version == v and blk.packed is p3          # stale, as documented
    blk.invalidate_packed()
    assert blk.packed is not p3 and float(blk.packed.wp.max()) == 2.0


def test_composite_modules_repack_when_a_child_block_changes():
    """the GRU (stacked z|r weights) and the decoder (stacked flow|mask head) pack tensors that live
    in ConvBlock children: their cache key must cover those tensors too."""
    from scflow_amd.modules import ConvGRU
    gru = ConvGRU(16, 32, 'SeqConv')
    p0 = gru.packed
    assert gru.packed is p0
    with torch.no_grad():
        gru.conv_r[1].conv.weight.zero_()
    p1 = gru.packed
    assert p1 is not p0 and float(p1[1][0].wp.abs().sum()) < float(p0[1][0].wp.abs().sum())
    dec = scflow_amd.build_decoder(scflow_amd.scflow_model_cfg()['decoder'])
    h0 = dec.packed
    with torch.no_grad():
        dec.mask_pred.layers[0].conv.bias.fill_(3.0)
    h1 = dec.packed
    assert h1 is not h0 and float(h1.bias[256:].min()) == 3.0 and float(h1.bias[:256].max()) < 3.0


def test_conv_tile_selection_dry_run():
    """scf_conv2d_query (no launch, runs without a GPU): the tile the LDS-DMA dispatcher picks for
    the layers of the two benchmark configurations -- by estimated slot utilisation x tile rate
    (conv_dma.hip): (2,2) where its block count is a whole number of 512-slot rounds, (2,1) / (1,1)
    where Cout or a partial last round argue for it; 16-column fragments on an 80-wide map."""
    import ctypes as C
    from scflow_amd import ops, _lib
    lib = _lib.load()

    def q(n, cin, cout, k, pad, H, W, stride=1, c0=None):
        w = torch.randn(cout, cin, *k) * 0.05
        pc = ops.PackedConv.from_weight(w, None, stride, pad)
        d = _lib.ConvDesc()
        d.N, d.H, d.W = n, H, W
        d.C0 = cin if c0 is None else c0
        d.C1 = 0 if c0 is None else cin - c0
        d.in0 = 0x1000
        d.in1 = 0x2000 if c0 is not None else None
        d.in0_nstride = d.in1_nstride = cin * H * W
        d.wp, d.Mld, d.Cout = pc.wp.data_ptr(), pc.mld, cout
        d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.KC = pc.kh, pc.kw, stride, pc.pad_h, pc.pad_w, pc.kc
        d.out, d.out_nstride, d.out_div = 0x3000, cout * H * W, 1.0
        if pc.wp4 is not None:
            d.wp_a4, d.a4_groups, d.a4_mld = pc.wp4.data_ptr(), pc.g4, pc.mld
        if pc.wp4s is not None:
            d.wp_a4s, d.a4s_groups, d.a4_mld = pc.wp4s.data_ptr(), pc.g4s, pc.mld
        info = (C.c_int32 * 4)()
        assert lib.scf_conv2d_query(C.byref(d), info) == 0
        assert info[3] < 0, 'expected the LDS-DMA kernel'
        return info[0], info[1], info[2]

    # batch 32, 256x256 crops: 32x32 maps in the loop, whole rounds of 512 blocks
    assert q(32, 128, 512, (3, 3), 1, 32, 32) == (2, 2, 1024)
    assert q(32, 256, 256, (1, 5), (0, 2), 32, 32, c0=128) == (2, 2, 512)
    assert q(32, 256, 128, (1, 5), (0, 2), 32, 32, c0=128) == (2, 1, 512)      # Cout = 128: (2,2) would give 256 blocks
    assert q(32, 256, 126, (3, 3), 1, 32, 32) == (2, 1, 512)
    assert q(32, 128, 64, (3, 3), 1, 32, 32) == (1, 1, 512)
    assert q(64, 64, 64, (3, 3), 1, 128, 128) == (2, 2, 4096)
    assert q(32, 324, 256, (1, 1), 0, 32, 32) == (2, 2, 512)                   # dense 1x1, stride 1
    # 8 x 480x640 crops: 60x80 maps; 5 x 16-column fragments per row; (2,2) would be 640 blocks = 1.25 rounds
    assert q(8, 128, 256, (3, 3), 1, 60, 80) == (2, 1, 8 * 8 * 5 * 4)
