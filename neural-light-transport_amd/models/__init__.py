"""Model registry (mirrors reference nlt/models/__init__.py:17-20)."""
from importlib import import_module


def get_model_class(name):
    mod = import_module('.' + name, __name__)
    return mod.Model
