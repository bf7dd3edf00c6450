"""data_gen/get_neighbors.py:52-71 on HIP (generalised from 1 to k neighbours)."""
import numpy as np
import torch

from .. import _capi as C


def get_neighbors(phys_and_virt, phys, k=1, device='cuda'):
    """Each of `phys_and_virt` ({'name', 'position'} dicts) -> name of its nearest `phys` entry at
    non-zero distance (first minimum wins).  k > 1 returns a list of k names, nearest first."""
    pos = lambda objs: torch.tensor(np.array([o['position'] for o in objs], np.float64).reshape(-1, 3), device=device)
    idx = C.knn_indices(pos(phys_and_virt), pos(phys), k).cpu().numpy()
    neighbors = {}
    for ref, row in zip(phys_and_virt, idx):
        assert row[0] >= 0                                     # get_neighbors.py:68
        names = [phys[i]['name'] for i in row if i >= 0]
        neighbors[ref['name']] = names[0] if k == 1 else names
    return neighbors
