"""Host mirrors of data_gen/ and nlt/datasets/nlt.py (nlt_amd.data_gen, nlt_amd.datasets) on CPU:
the C-ABI adapters are replaced by tests/fake_capi.py (oracle emulation), so what is checked here
is the host logic -- argument plumbing, sample ordering of calc_bidir_mapping, the hold-out split,
neighbour lookup, error behaviour.  The kernels themselves are checked by tests/test_gpu_buffers.py."""
import numpy as np
import pytest
import torch

import fake_capi
import test_gpu_buffers as G
from nlt_amd import _capi


@pytest.fixture()
def cpu_capi(monkeypatch):
    fake_capi.install(monkeypatch)
    monkeypatch.setattr(G, 'DEV', 'cpu')
    return _capi


def test_data_gen_mirrors_on_cpu(cpu_capi, monkeypatch):
    G.test_data_gen_mirrors(cpu_capi)


def test_dataset_load_batch_on_cpu(cpu_capi, monkeypatch):
    G.test_dataset_load_batch(cpu_capi)


def test_get_neighbors_matches_reference_golden(cpu_capi, monkeypatch):
    from nlt_amd.data_gen import get_neighbors as gn
    named = lambda a, p: [{'name': '%s%03d' % (p, i), 'position': list(map(float, x))} for i, x in enumerate(a)]
    nn = gn.get_neighbors(named(G.G['knn_ref'], 'r'), named(G.G['knn_cand'], 'c'), device='cpu')
    assert [int(nn['r%03d' % i][1:]) for i in range(len(G.G['knn_ref']))] == G.G['knn_nn'].tolist()
    k2 = gn.get_neighbors(named(G.G['knn_ref'], 'r'), named(G.G['knn_cand'], 'c'), k=2, device='cpu')
    assert all(v[0] == nn[key] and len(v) == 2 for key, v in k2.items())
