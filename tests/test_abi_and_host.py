"""CPU: the C-ABI library loads and exports every symbol include/sda_hip.h declares (no compute calls), the Python
mirror of `struct sda_conv_desc` matches the C layout, the product has no CPU fallback, and the reference-compatible
module tree produces the reference's state_dict."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sda_hip.h')


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sda_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from sda_amd import _lib, build
    build.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in sda_hip.h but not exported'
    assert set(declared) == set(_lib.SIGNATURES), (set(declared) ^ set(_lib.SIGNATURES))
    assert lib.sda_abi_version() == 13


@pytest.mark.parametrize('mirror,ctype', [('ConvDesc', 'sda_conv_desc'), ('Block1dDesc', 'sda_block1d_desc'),
                                          ('Net1dDesc', 'sda_net1d_desc'), ('Net1dFuse', 'sda_net1d_fuse'), ('MlpDesc', 'sda_mlp_desc'), ('MlpWin', 'sda_mlp_win'), ('Conv3dDesc', 'sda_conv3d_desc')])
def test_desc_layouts_match_c(tmp_path, mirror, ctype):
    """sizeof/offsetof of the ctypes mirrors == what gcc sees in the header."""
    from sda_amd import _lib
    Desc = getattr(_lib, mirror)
    fields = [f[0] for f in Desc._fields_]
    src = tmp_path / 'layout.c'
    prints = '\n'.join(f'printf("{f} %zu\\n", offsetof({ctype}, {f}));' for f in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{HEADER}"\nint main(){{printf("size %zu\\n", '
                   f'sizeof({ctype}));\n{prints}\nreturn 0;}}')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', str(src), '-o', str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    assert int(out['size']) == ctypes.sizeof(Desc)
    for f in fields:
        assert int(out[f]) == getattr(Desc, f).offset, f


def test_bad_arguments_are_rejected_without_a_gpu():
    from sda_amd import _lib
    from sda_amd.ops import make_conv_desc
    lib = _lib.load()
    d = make_conv_desc(x_ptr=None, n=1, cx=1, hs=1, ws=1, x_sc=1, x_sy=1, x_sx=1, x_sn_outer=1, w_ptr=None, cin_pad=8,
                       cout_pad=32, cout=1, kh=3, kw=3, out_ptr=None, ho=1, wo=1, mt=1)
    assert lib.sda_conv_igemm(ctypes.byref(d), None) == -1
    assert lib.sda_conv_igemm_lds_bytes(ctypes.byref(d)) == -1
    assert lib.sda_ln_stats(None, 1, 2, 3, None, 0, 1e-5, 1, None, None, None) == -1
    assert lib.sda_fold(None, 1, 1, 1, 1, 1, None, None) == -1


def test_lds_plan_for_reference_shapes():
    """K64 layer shapes fit the 160 KiB LDS with room for >= 2 workgroups per CU."""
    from sda_amd import _lib
    from sda_amd.ops import make_conv_desc, pick_mt, round_up
    lib = _lib.load()
    for cin, cout, h, stride in ((11, 96, 64, 1), (96, 96, 64, 1), (96, 192, 64, 2), (192, 192, 32, 1),
                                 (192, 384, 32, 2), (384, 384, 16, 1), (96, 10, 64, 1), (96, 96, 256, 1)):
        mt = pick_mt(cout)
        d = make_conv_desc(x_ptr=8, n=4, cx=cin, hs=h, ws=h, x_sc=h * h, x_sy=h, x_sx=1, x_sn_outer=cin * h * h, w_ptr=8,
                           cin_pad=round_up(cin, 8), cout_pad=round_up(cout, 32 * mt), cout=cout, kh=3, kw=3, out_ptr=8,
                           ho=h // stride, wo=h // stride, mt=mt, stride_h=stride, stride_w=stride, circular=True)
        nbytes = lib.sda_conv_igemm_lds_bytes(ctypes.byref(d))
        assert 0 < nbytes <= 80 * 1024, (cin, cout, h, nbytes)


def test_no_cpu_fallback():
    from sda_amd._lib import SdaHipError
    from sda_amd.experiments.lorenz import make_global_score
    from sda_amd.score import VPSDE
    net = make_global_score()
    with pytest.raises(SdaHipError):
        net(torch.zeros(1, 8, 3), torch.tensor(0.5))
    with pytest.raises(SdaHipError):
        VPSDE(net, shape=(8, 3)).sample((1,), steps=2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'sda_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.hpp'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
                assert '/root/reference' not in text, f


def _ref_shapes(name):
    data = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    return {k.split('/', 1)[1]: tuple(int(i) for i in data[k]) for k in data.files if k.startswith('shapes/')}


def test_state_dict_matches_reference_inventory():
    from sda_amd.experiments.kolmogorov import make_score
    from sda_amd.experiments.lorenz import make_global_score
    from sda_amd.score import GaussianScore, VPSDE
    k64 = make_score(window=5, embedding=64, hidden_channels=(96, 192, 384), hidden_blocks=(3, 3, 3))
    assert {k: tuple(v.shape) for k, v in k64.state_dict().items()} == _ref_shapes('k64_keys')
    assert sum(p.numel() for p in k64.parameters()) == 22874922
    lor = make_global_score()
    assert {k: tuple(v.shape) for k, v in lor.state_dict().items()} == _ref_shapes('lorenz_global_keys')
    outer = VPSDE(GaussianScore(torch.zeros(3), A=lambda x: x, std=0.1, sde=VPSDE(lor, shape=())), shape=(65, 3))
    keys = set(outer.state_dict())
    assert {'device', 'eps.y', 'eps.std', 'eps.gamma', 'eps.sde.device'} <= keys      # SURVEY 8b device quirk
    assert outer.dims == (-2, -1) and outer.eps.sde.shape == ()
    with pytest.raises(ValueError):
        VPSDE(lor, shape=(), alpha='nope')


def test_schedule_matches_golden():
    from sda_amd.score import SubSubVPSDE, SubVPSDE, VPSDE
    data = np.load(os.path.join(ROOT, 'tests', 'golden', 'schedule.npz'))
    t = torch.from_numpy(data['t'])
    ident = torch.nn.Identity()
    sde = VPSDE(ident, shape=())
    assert torch.allclose(sde.mu(t), torch.from_numpy(data['mu_cos']), rtol=1e-6, atol=1e-7)
    assert torch.allclose(sde.sigma(t), torch.from_numpy(data['sigma_cos']), rtol=1e-6, atol=1e-7)
    assert torch.allclose(SubVPSDE(ident, shape=()).sigma(t), torch.from_numpy(data['sigma_sub']), rtol=1e-6, atol=1e-7)
    assert torch.allclose(SubSubVPSDE(ident, shape=()).sigma(t), torch.from_numpy(data['sigma_subsub']), rtol=1e-6,
                          atol=1e-7)
    assert torch.allclose(VPSDE(ident, shape=(), alpha='lin').mu(t), torch.from_numpy(data['mu_lin']), rtol=1e-6)
    assert torch.allclose(VPSDE(ident, shape=(), alpha='exp').mu(t), torch.from_numpy(data['mu_exp']), rtol=1e-6)


def test_load_score_roundtrip(tmp_path):
    """Checkpoint compatibility: state.pth + config.json as the reference's train scripts write them."""
    import json
    from sda_amd.experiments import kolmogorov as K, lorenz as Lz
    cfg = dict(window=3, embedding=16, hidden_channels=(4, 8), hidden_blocks=(1, 1), kernel_size=3, activation='SiLU',
               epochs=1, batch_size=2)
    net = K.make_score(**cfg)
    (tmp_path / 'k').mkdir()
    torch.save(net.state_dict(), tmp_path / 'k' / 'state.pth')
    json.dump(cfg, open(tmp_path / 'k' / 'config.json', 'w'))
    net2 = K.load_score(tmp_path / 'k' / 'state.pth')
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    for local in (False, True):
        cfg = dict(embedding=8, hidden_channels=(8,), hidden_blocks=(1,), window=5, width=16, depth=2, activation='SiLU')
        net = Lz.make_local_score(**cfg) if local else Lz.make_global_score(**cfg)
        d = tmp_path / f'l{int(local)}'
        d.mkdir()
        torch.save(net.state_dict(), d / 'state.pth')
        json.dump(cfg, open(d / 'config.json', 'w'))
        net2 = Lz.load_score(d / 'state.pth', local=local)
        assert set(net.state_dict()) == set(net2.state_dict())
