"""The mini-mmcv used to generate the conv-stack fixtures is 'mmcv 1.3.16 restated'.  Its
assumptions are written down in tests/golden/MMCV_SHIM_AUDIT.md; this test keeps that table honest:
every call the reference makes into the mmcv factories on the hot-path files (found by AST), and
every keyword it passes, must appear in the audit.  Runs only where /root/reference exists."""
import ast
import os
import re

import pytest

REF = '/root/reference'
FILES = ['models/decoder/raft_decoder.py', 'models/decoder/scflow_decoder.py',
         'models/decoder/raft_decoder_mask.py', 'models/head/pose_head.py',
         'models/encoder/raft_encoder.py', 'models/backbone/resnet.py']
APIS = {'ConvModule', 'build_norm_layer', 'build_conv_layer', 'build_activation_layer',
        'build_plugin_layer'}
AUDIT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'MMCV_SHIM_AUDIT.md')

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present')


def _call_sites():
    out = []
    for f in FILES:
        tree = ast.parse(open(os.path.join(REF, f)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call):
                fn = node.func
                name = fn.id if isinstance(fn, ast.Name) else getattr(fn, 'attr', None)
                if name in APIS:
                    out.append((f, node.lineno, name, sorted(k.arg for k in node.keywords if k.arg)))
    return out


def _audited():
    """(file -> set of audited line numbers, api -> set of audited kwargs) from the markdown table."""
    text = open(AUDIT).read()
    lines, kwargs = {}, {a: set() for a in APIS}
    for row in text.splitlines():
        m = re.match(r'\| `(models/[\w/]+\.py):([\d,\-]+)`[^|]*\| ([^|]+)\| ([^|]*)\|', row)
        if not m:
            continue
        f, spec, apis, kws = m.group(1), m.group(2), m.group(3), m.group(4)
        s = lines.setdefault(f, set())
        for part in spec.split(','):
            if '-' in part:
                lo, hi = part.split('-')
                s.update(range(int(lo), int(hi) + 1))
            else:
                s.add(int(part))
        names = set(re.findall(r'[a-z_]+', kws)) - {'positional'}
        for a in APIS:
            if a in apis:
                kwargs[a] |= names
    return lines, kwargs


def test_every_mmcv_call_site_and_kwarg_is_audited():
    sites = _call_sites()
    assert len(sites) >= 30
    lines, kwargs = _audited()
    for f, ln, api, kws in sites:
        assert ln in lines.get(f, ()), f'{f}:{ln} ({api}) is not in MMCV_SHIM_AUDIT.md'
        extra = set(kws) - kwargs[api]
        assert not extra, f'{f}:{ln} passes {sorted(extra)} to {api}: not audited'


def test_shim_matches_the_audited_semantics():
    """the behaviours of table rows A1-A4, checked on the shim itself."""
    import importlib.util
    import torch.nn as nn
    spec = importlib.util.spec_from_file_location(
        '_refshim', os.path.join(os.path.dirname(AUDIT), '_refshim.py'))
    shim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shim)
    m = shim.ConvModule(8, 16, 3, padding=1)                              # A1: bias iff no norm, ReLU
    assert m.conv.bias is not None and isinstance(m.activate, nn.ReLU) and m.activate.inplace
    g = shim.ConvModule(8, 32, 3, stride=2, padding=1, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                        act_cfg=dict(type='ReLU'))
    assert g.conv.bias is None and isinstance(g.gn, nn.GroupNorm) and g.gn.num_groups == 32 and g.gn.eps == 1e-5
    assert sorted(k for k, _ in g.named_parameters()) == ['conv.weight', 'gn.bias', 'gn.weight']
    s = shim.ConvModule(8, 8, (1, 5), padding=(0, 2), act_cfg=dict(type='Sigmoid'))
    assert isinstance(s.activate, nn.Sigmoid)
    name, layer = shim.build_norm_layer(dict(type='IN'), 64, postfix=1)     # A3
    assert name == 'in1' and isinstance(layer, nn.InstanceNorm2d) and not layer.affine \
        and not layer.track_running_stats and not list(layer.parameters()) and not list(layer.buffers())
    name, layer = shim.build_norm_layer(dict(type='BN', requires_grad=True), 64, postfix=2)
    assert name == 'bn2' and isinstance(layer, nn.BatchNorm2d) and layer.eps == 1e-5 and layer.track_running_stats
    name, layer = shim.build_norm_layer(dict(type='SyncBN'), 8)
    assert name == 'bn' and isinstance(layer, nn.BatchNorm2d)
    c = shim.build_conv_layer(None, 3, 64, kernel_size=7, stride=2, padding=3, bias=True)   # A2
    assert isinstance(c, nn.Conv2d) and c.bias is not None and c.stride == (2, 2)
