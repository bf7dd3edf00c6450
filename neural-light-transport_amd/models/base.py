"""Model contract (mirrors reference nlt/models/base.py:26-140): config, `net` dict,
loss-string parsing, `register_trainable`, mode validation."""
from ..networks import base as basenet


class Model:
    def __init__(self, config):
        self.config = config
        self.net = {'main': basenet.Network()}   # trainable networks go in here
        self.trainable_registered = False        # call register_trainable() before training
        self.wloss = self._init_loss()           # [(weight, loss callable)]

    def _init_loss(self):
        raise NotImplementedError

    @staticmethod
    def _parse_loss_and_weight(weight_loss_str):
        """'1e+2lpips' -> ('lpips', 100.); 'barron' -> ('barron', 1.): the longest prefix
        that parses as a float is the weight."""
        for i in range(len(weight_loss_str), -1, -1):
            try:
                weight = float(weight_loss_str[:i])
            except ValueError:
                continue
            return weight_loss_str[i:], weight
        return weight_loss_str, 1.

    def register_trainable(self):
        """Aliases every layer of every net as attribute `net_<name>_layer<i>` (the names the
        reference's checkpoints use) and freezes the trainable-variable list."""
        registered = []
        for net_name, net in self.net.items():
            attr = 'net_' + net_name
            assert attr.isidentifier(), "network name '%s' does not make a valid identifier" % net_name
            for i, layer in enumerate(net.layers):
                full = attr + '_layer%d' % i
                assert not hasattr(self, full), "Can't register `%s`: already an attribute" % full
                setattr(self, full, layer)
                registered.append(full)
        self._registered = registered
        self.trainable_registered = True

    @property
    def trainable_variables(self):
        assert self.trainable_registered, "Register the trainable layers before using `trainable_variables`"
        out = []
        for name in self._registered:
            out += getattr(self, name).variables()
        return out

    @staticmethod
    def _validate_mode(mode):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError(mode)

    def __call__(self, batch, mode, **kwargs):
        return self.call(batch, mode, **kwargs)

    def call(self, batch, mode):
        raise NotImplementedError

    def compute_loss(self, pred, gt, **kwargs):
        raise NotImplementedError

    def vis_batch(self, data_dict, outdir, mode, dump_raw_to=None):
        raise NotImplementedError

    def compile_batch_vis(self, batch_vis_dirs, outpref, mode):
        raise NotImplementedError
