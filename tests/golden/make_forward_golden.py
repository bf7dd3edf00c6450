"""Writes tests/golden/nlt_forward_64.npz: oracle outputs for a seeded 64x64 batch.
(TensorFlow is not installable here, so this fixture pins the ORACLE, not TF itself.)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import nlt_oracle as O

WSEED, BSEED = 21, 22
om = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=WSEED)
batch, nn = O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=BSEED)
with torch.no_grad():
    pred_c, gt_c, _, vis = om.call(batch, 'train', nn_list=nn)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nlt_forward_64.npz'),
                    weight_seed=WSEED, batch_seed=BSEED, pred=vis['pred'].numpy(), pred_camspc=pred_c.numpy(),
                    gt_camspc=gt_c.numpy())
print('ok', float(vis['pred'].abs().mean()))
