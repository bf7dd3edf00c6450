"""Network base (mirrors reference nlt/networks/base.py:26-40)."""


class Network:
    def __init__(self):
        self.layers = []

    def __call__(self, x):
        raise NotImplementedError

    @staticmethod
    def str2none(str_):
        """There is no `config.getnone()`: the string 'none' (any case) means None."""
        assert isinstance(str_, str), "Call this only on strings"
        return None if str_.lower() == 'none' else str_
