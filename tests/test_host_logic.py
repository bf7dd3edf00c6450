"""CPU: host-side mirror of the reference's plugin surface (no device calls)."""
import os
import re
import subprocess
import sys

import pytest

import nlt_amd
from nlt_amd.models import get_model_class
from nlt_amd.networks import convnet, elements
from nlt_amd.util.net import gen_feat_n
from oracle import nlt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gen_feat_n_matches_oracle():
    for a, b, c in [(16, 256, 3), (16, 1024, 3), (8, 64, 3), (16, 16, 3), (12, 100, 3), (32, 64, 1)]:
        assert gen_feat_n(a, b, c) == O.gen_feat_n(a, b, c)
    with pytest.raises(AssertionError):
        gen_feat_n(64, 16)


def test_convnet_structure():
    net = convnet.Network(16, 256, 2, 2, norm_type='None', act_type='leakyrelu', pool_type='None')
    assert len(net.layers) == 14
    assert net.is_contracting == [True] * 7 + [False] * 7
    assert net.spatsize_changes == [1] + [0.5] * 6 + [2] * 6 + [1]
    (c1, a1), (c2, a2) = net.layers[1].convs()
    assert (c1.mode, c2.mode) == (nlt_amd.capi.CONV_K2S2, nlt_amd.capi.CONV_K2S1) and a1.alpha == 0.3
    (d1, _), (d2, _) = net.layers[7].convs()
    assert (d1.mode, d2.mode) == (nlt_amd.capi.DECONV_K2S2, nlt_amd.capi.DECONV_K2S1) and d1.n_ch_out == 128
    assert convnet.Network(16, 1024, 2, 2, act_type='relu').layers[1].convs()[0][1].alpha == 0.0
    with pytest.raises(AssertionError):
        convnet.Network(8, 64, 2, 2)                       # depth0 must be 16
    for bad in (dict(norm_type='instance'), dict(act_type='gelu'), dict(pool_type='l2')):
        with pytest.raises(NotImplementedError):
            convnet.Network(16, 256, 2, 2, **bad)
    # the branches that ARE built: stand-alone layers, executed by nlt_amd/generic.py
    from nlt_amd.networks import elements as E
    pooled = convnet.Network(16, 32, 2, 2, norm_type='pixel', act_type='elu', pool_type='max')
    down, up = pooled.layers[1], pooled.layers[4]
    assert [type(l).__name__ for l in down.layers] == ['Conv2D', 'PixelNorm', 'Act', 'Conv2D', 'PixelNorm', 'Act', 'Pool2D']
    assert isinstance(up.layers[0], E.Sequential) and [type(l).__name__ for l in up.layers[0].layers] == ['UpSample2D', 'Conv2D']
    # norm = layer / batch (elements.py:51-56): a two-variable layer (gamma, beta) after every conv of a block
    for kind in ('layer', 'batch'):
        normed = convnet.Network(16, 32, 2, 2, norm_type=kind, act_type='leakyrelu')
        blk = normed.layers[1]
        assert [type(l).__name__ for l in blk.layers] == ['Conv2D', 'ChannelNorm', 'Act', 'Conv2D', 'ChannelNorm', 'Act', 'Identity']
        assert [type(l).__name__ for l in blk.all_convs()] == ['Conv2D', 'ChannelNorm', 'Conv2D', 'ChannelNorm'] and not blk.is_plain()
        assert blk.layers[1].eps == 1e-3 and blk.layers[1].name == kind
    assert down.layers[2].kind == 'elu' and not down.is_plain() and len(up.all_convs()) == 3
    assert pooled.spatsize_changes[1] == 0.25 and pooled.spatsize_changes[4] == 4
    assert convnet.Network.str2none('None') is None and convnet.Network.str2none('x') == 'x'


def test_model_registry_and_contract():
    Model = get_model_class('nlt')
    m = Model(nlt_amd.make_config(uvh=64, uvw=64, imh=64, imw=64, loss='barron,2e+0l2'))
    assert len(m.net['query'].layers) == 14 and len(m.net['obs'].layers) == 7     # decoder dropped from obs
    assert [w for w, _ in m.wloss] == [1.0, 2.0]
    assert m._parse_loss_and_weight('1e+0lpips') == ('lpips', 1.0)
    with pytest.raises(AssertionError):
        m.trainable_variables
    m.register_trainable()
    assert hasattr(m, 'net_query_layer13') and hasattr(m, 'net_obs_layer6') and not hasattr(m, 'net_obs_layer7')
    with pytest.raises(ValueError):
        m._validate_mode('infer')
    with pytest.raises(NotImplementedError):
        Model(nlt_amd.make_config(loss='barron,1e+0lpips'))     # frozen AlexNet blob missing upstream
    with pytest.raises(ModuleNotFoundError):
        get_model_class('nope')


def test_product_never_imports_oracle():
    """The shipped package must not route through the CPU oracle (or /root/reference)."""
    pkg = os.path.join(ROOT, 'neural-light-transport_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert '/root/reference' not in src, f
    code = "import sys; sys.path.insert(0, %r); import nlt_amd; assert 'oracle' not in sys.modules" % ROOT
    subprocess.run([sys.executable, '-c', code], check=True)
