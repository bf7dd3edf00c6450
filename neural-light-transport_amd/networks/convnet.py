"""Per-texel conv encoder-decoder (mirrors reference nlt/networks/convnet.py:30-90)."""
import numpy as np

from ..util import net as netutil
from .seq import Network as BaseNetwork
from .elements import conv, norm, act, pool, iden, deconv, upconv, Sequential


class Network(BaseNetwork):
    def __init__(self, depth0, depth, kernel, stride, norm_type=None, act_type='relu', pool_type=None):
        super().__init__()
        norm_type = self.str2none(norm_type) if isinstance(norm_type, str) else norm_type
        pool_type = self.str2none(pool_type) if isinstance(pool_type, str) else pool_type
        n_feat = netutil.gen_feat_n(depth0, depth)
        self.is_contracting, self.spatsize_changes = [], []
        # 1x1 conv producing the original-resolution feature map
        self.layers.append(conv(1, n_feat[0], stride=1))
        self.is_contracting.append(True)
        self.spatsize_changes.append(1)
        prev_n = 0
        for n in n_feat[:-1]:
            if n >= prev_n:     # spatially contracting (64 -> 64 counts as contracting)
                self.layers.append(Sequential([
                    conv(kernel, n, stride=stride), norm(norm_type), act(act_type),
                    conv(kernel, n, stride=1), norm(norm_type), act(act_type),
                    pool(pool_type)]))
                self.is_contracting.append(True)
                change = 1 / stride
                if pool_type is not None:
                    change *= 1 / 2
            else:               # spatially expanding
                self.layers.append(Sequential([
                    iden() if pool_type is None else upconv(n),
                    deconv(kernel, n, stride=stride), norm(norm_type), act(act_type),
                    deconv(kernel, n, stride=1), norm(norm_type), act(act_type)]))
                self.is_contracting.append(False)
                change = stride
                if pool_type is not None:
                    change *= 2
            self.spatsize_changes.append(change)
            prev_n = n
        # back at the original resolution: 1x1 conv to the output channel count
        self.layers.append(conv(1, n_feat[-1], stride=1))
        self.is_contracting.append(False)
        self.spatsize_changes.append(1)
        assert np.cumprod(self.spatsize_changes)[-1] == 1, "Resolution doesn't return to the original value"
