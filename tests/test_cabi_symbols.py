"""CPU: libscflow_hip.so loads and exports every symbol include/scflow_hip.h (operator ABI) and
include/scflow_hip_prof.h (measurement aids) declare (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

from scflow_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(headers=('scflow_hip.h', 'scflow_hip_prof.h')):
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, 'include', h)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        out |= set(re.findall(r'\b(scf_[a-z0-9_]+)\s*\(', text))
    return sorted(out)


def test_library_built():
    assert os.path.exists(_lib.LIB_PATH), 'run python scflow_amd/csrc/build.py'


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in scflow_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature in scflow_amd/_lib.py'
    for name in _lib.SIGNATURES:
        assert name in declared, f'{name} bound in _lib.py but not declared in the header'


def test_operator_header_holds_no_measurement_entry_points():
    ops_abi = _declared_symbols(('scflow_hip.h',))
    assert not [n for n in ops_abi if n.startswith('scf_timer') or n.endswith(('_timed', '_query')) or n.startswith('scf_conv_log')]
    prof = set(_declared_symbols(('scflow_hip_prof.h',))) - set(ops_abi)
    assert prof == {'scf_timer_create', 'scf_timer_destroy', 'scf_timer_arm', 'scf_timer_elapsed_us',
                    'scf_conv2d_query', 'scf_conv_log_enable', 'scf_conv_log_read', 'scf_tune'}


def test_version_and_error_strings():
    lib = _lib.load()
    text = open(os.path.join(ROOT, 'include', 'scflow_hip.h')).read()
    major = int(re.search(r'#define\s+SCF_ABI_MAJOR\s+(\d+)', text).group(1))
    assert lib.scf_version() // 100 == major == _lib.ABI_MAJOR      # header, library and binding agree
    assert int(re.search(r'#define\s+SCF_MAX_LEVELS\s+(\d+)', text).group(1)) == _lib.MAX_LEVELS
    assert lib.scf_error_string(0) == b'ok'
    assert b'invalid' in lib.scf_error_string(-1)


def test_tune_knobs_validate_their_values():
    """scf_tune (scflow_hip_prof.h): returns the previous value, rejects unknown keys and out-of-range values, and
    leaves the defaults in place (no GPU involved: the knobs are host-side dispatch state)."""
    lib = _lib.load()
    from scflow_amd import ops
    assert lib.scf_tune(999, 0) < 0
    assert lib.scf_tune(ops.TUNE_KEYS['wino1d4'], 3) < 0 and lib.scf_tune(ops.TUNE_KEYS['wino1d4'], -1) < 0
    try:
        assert ops.tune('wino1d4', 0) == 1            # default: F(4, 5) where the dispatch prefers it
        assert ops.tune('wino1d4', 2) == 0
        assert ops.tune('wino1d4', 1) == 2
        assert ops.tune('wino_variant', 0) == 0
        # r6: the lab-only variants are refused by the product library instead of silently running variant 2;
        # lookup_pipe = 6 (three groups per block) is documented and accepted
        assert lib.scf_tune(ops.TUNE_KEYS['wino_variant'], 3) < 0 and lib.scf_tune(ops.TUNE_KEYS['wino_variant'], 4) < 0
        assert ops.tune('wino_variant', 2) == 0 and ops.tune('wino_variant', 0) == 2
        assert ops.tune('lookup_pipe', 6) == 0 and ops.tune('lookup_pipe', 0) == 6
        # r6: merged launches of scf_conv2d_pair (default on), 0 / 1 only
        assert ops.tune('conv_pair', 1) == 0 and ops.tune('conv_pair', 0) == 1
        assert lib.scf_tune(ops.TUNE_KEYS['conv_pair'], 2) < 0
        # the lookup knobs (r5): pipelined variant 0..3, store policy 0..5
        assert lib.scf_tune(ops.TUNE_KEYS['lookup_pipe'], 7) < 0 and lib.scf_tune(ops.TUNE_KEYS['lookup_store'], 6) < 0
        assert ops.tune('lookup_pipe', 2) == 0 and ops.tune('lookup_pipe', 0) == 2
        assert ops.tune('lookup_store', 4) == 0 and ops.tune('lookup_store', 0) == 4
        # r5: merged launches of the iteration (default on), half-domain F(4, 5) kernel (default off): 0 / 1 only
        assert ops.tune('iter_merge', 0) == 1 and ops.tune('iter_merge', 1) == 0
        assert ops.tune('wino1d4_half', 1) == 0 and ops.tune('wino1d4_half', 0) == 1
        assert lib.scf_tune(ops.TUNE_KEYS['iter_merge'], 2) < 0 and lib.scf_tune(ops.TUNE_KEYS['wino1d4_half'], -1) < 0
        # ops.tune never hands an error code back as a "previous value"
        with pytest.raises(ValueError):
            ops.tune('no_such_knob', 1)
        with pytest.raises(_lib.ScflowHipError):
            ops.tune('wino1d4', 7)
    finally:                                          # a failed assertion must not leave a knob off its default
        for key, default in (('wino1d4', 1), ('wino_variant', 0), ('lookup_pipe', 0), ('lookup_store', 0), ('iter_merge', 1),
                             ('wino1d4_half', 0), ('conv_pair', 0)):
            lib.scf_tune(ops.TUNE_KEYS[key], default)


def test_conv_desc_layout_matches_c():
    """sizeof(scf_conv_desc) from a C compile must equal the ctypes mirror."""
    import ctypes, subprocess, tempfile
    src = ('#include "scflow_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu\\n", '
           'sizeof(scf_conv_desc), sizeof(scf_gru_pass));printf("%zu %zu\\n", sizeof(scf_scflow_iter), sizeof(scf_iter_gn));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(d, 't')
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        size, gsize, isize, nsize = map(int, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert size == ctypes.sizeof(_lib.ConvDesc)
    assert gsize == ctypes.sizeof(_lib.GruPass)
    assert isize == ctypes.sizeof(_lib.ScflowIter) and nsize == ctypes.sizeof(_lib.IterGN)


def test_c_weight_packers_match_host_packers():
    """scf_pack_conv_weight / _a4 (what a C or ctypes caller uses) == the torch packers the
    Python modules use, bit for bit; bad arguments are refused."""
    import ctypes as C
    import torch
    from scflow_amd import ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    for cout, cin, kh, kw in ((40, 20, 3, 3), (256, 384, 1, 5), (33, 70, 5, 1), (2, 3, 7, 7), (64, 64, 1, 1)):
        w = torch.randn((cout, cin, kh, kw), generator=g).contiguous()
        for kc in (2, 8, 32):
            want, mld = ops.pack_conv_weight(w, kc)
            n = lib.scf_pack_conv_weight_size(cout, cin, kh, kw, kc)
            assert n == want.numel() and mld == (cout + 31) // 32 * 32
            out = torch.full((n,), float('nan'))
            assert lib.scf_pack_conv_weight(w.data_ptr(), cout, cin, kh, kw, kc, out.data_ptr()) == 0
            assert torch.equal(out, want.reshape(-1))
        for grp in (1, 2, 4):
            want, _ = ops.pack_conv_weight_a4(w, grp)
            n = lib.scf_pack_conv_weight_a4_size(cout, cin, kh, kw, grp)
            assert n == want.numel()
            out = torch.full((n,), float('nan'))
            assert lib.scf_pack_conv_weight_a4(w.data_ptr(), cout, cin, kh, kw, grp, out.data_ptr()) == 0
            assert torch.equal(out, want.reshape(-1))
    for cout, cin, kh, kw in ((64, 3, 7, 7), (128, 2, 7, 7), (64, 1, 3, 3), (40, 4, 3, 3)):
        w = torch.randn((cout, cin, kh, kw), generator=g).contiguous()
        want = ops.pack_conv_weight_taps(w)
        n = lib.scf_pack_conv_weight_taps_size(cout, cin, kh, kw)
        assert n == want.numel() and want.shape[0] % 8 == 0
        out = torch.full((n,), float('nan'))
        assert lib.scf_pack_conv_weight_taps(w.data_ptr(), cout, cin, kh, kw, out.data_ptr()) == 0
        assert torch.equal(out, want.reshape(-1))
    assert lib.scf_pack_conv_weight_taps_size(8, 5, 3, 3) < 0
    assert lib.scf_pack_conv_weight_size(4, 4, 3, 3, 5) < 0
    assert lib.scf_pack_conv_weight_a4_size(4, 4, 3, 3, 3) < 0
    assert lib.scf_pack_conv_weight(None, 4, 4, 3, 3, 8, None) < 0


def test_pyramid_layout_helpers_run_on_the_host():
    """scf_corr_level_floats / scf_corr_preferred_layout are plain host functions."""
    lib = _lib.load()
    assert lib.scf_corr_level_floats(60, 80, 0, 1) == 60 * 80
    assert lib.scf_corr_level_floats(60, 80, 1, 1) == 32 * 40 and lib.scf_corr_level_floats(60, 80, 1, 0) == 30 * 40
    assert lib.scf_corr_level_floats(60, 80, 3, 1) == 8 * 16 and lib.scf_corr_level_floats(60, 80, 3, 0) == 7 * 10
    assert lib.scf_corr_level_floats(8, 8, 4, 0) < 0          # an empty level
    assert lib.scf_corr_preferred_layout(32, 32, 4, 4) == 0b0001
    assert lib.scf_corr_preferred_layout(60, 80, 4, 4) == 0b0011
    assert lib.scf_corr_preferred_layout(128, 160, 4, 4) == 0b0111
    assert lib.scf_corr_preferred_layout(30, 44, 4, 4) == 0b0000      # level 0 is never padded; level 1 is 22 wide


def test_ops_reject_cpu_tensors():
    import torch
    from scflow_amd import ops
    with pytest.raises(_lib.ScflowHipError):
        ops.corr_lookup([torch.zeros(4, 1, 2, 2)], torch.zeros(1, 2, 2, 2))
