"""Dataset registry (mirrors reference nlt/datasets/__init__.py:15-20)."""
from importlib import import_module


def get_dataset_class(name):
    mod = import_module('.' + name, __name__)
    return mod.Dataset
