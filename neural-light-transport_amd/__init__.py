"""MI355X-native hot path of google/neural-light-transport (import as `nlt_amd`).

Mirrors the reference's plugin surface: `models.get_model_class('nlt')(config)`,
`model(batch, mode)`, `networks.convnet.Network(...)`; all device arithmetic is in
libnlt_hip.so (include/nlt_hip.h), reached through `_capi`.
"""
from . import _capi as capi          # noqa: F401
from . import networks, models, losses, optim, trainvali, nlt_test, metric   # noqa: F401
from .util import net as netutil     # noqa: F401


def make_config(**overrides):
    """A ConfigParser carrying the released dragon_specular.ini model keys
    (reference nlt/config/dragon_specular.ini:43-64), overridable by keyword."""
    from configparser import ConfigParser
    cfg = ConfigParser()
    d = dict(imh=512, imw=512, uvh=512, uvw=512, use_obs=True, skip_connect_base=True, depth0=16, depth=256,
             kernel=2, stride=2, norm='None', act='leakyrelu', pool='None', loss='l2', lr=1e-3, mgm=-1, bs=4,
             model='nlt', linear_space=False)
    d.update(overrides)
    for k, v in d.items():
        cfg.set('DEFAULT', k, str(v))
    return cfg
