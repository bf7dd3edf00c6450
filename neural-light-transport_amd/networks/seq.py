"""Sequential-flow network (mirrors reference nlt/networks/seq.py:27-41)."""
from .base import Network as BaseNetwork
from .elements import Sequential


class Network(BaseNetwork):
    def build(self, input_shape, device='cuda'):
        """input_shape: (N, H, W, C) as in Keras; creates every layer's variables."""
        seq = Sequential(self.layers)
        seq.build(input_shape[-1], device)
        for layer in self.layers:
            assert layer.built, "Some layers not built"

    def __call__(self, tensor):
        x = tensor
        y = None
        for layer in self.layers:
            y = layer(x)
            x = y
        return y
