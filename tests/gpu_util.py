"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch

import nlt_amd
from nlt_amd.models import get_model_class
from oracle import nlt_oracle as O


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def make_pair(depth=256, uv=64, im=64, loss='l2', seed=0, use_obs=True, skip_connect_base=True, act='leakyrelu', **product_only):
    """(oracle model, product model) sharing the same Keras-layout weights."""
    om = O.OracleModel(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, loss=loss, seed=seed,
                       use_obs=use_obs, skip_connect_base=skip_connect_base, act=act)
    cfg = nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=im, imw=im, loss=loss,
                              use_obs=use_obs, skip_connect_base=skip_connect_base, act=act, **product_only)
    pm = get_model_class('nlt')(cfg)
    pm.load_weights(om.numpy_weights())
    pm.register_trainable()
    return om, pm


def to_device_batch(batch, nn_list, device='cuda'):
    """Oracle batch (+ list of k neighbours) -> product batch with [N,k,H,W,3] neighbour tensors."""
    dev = lambda t: None if t is None else t.to(device).contiguous()
    b = list(batch)
    nn_base = torch.stack([x[0] for x in nn_list], 1)
    nn_rgb = torch.stack([x[1] for x in nn_list], 1)
    out = [dev(t) if torch.is_tensor(t) else t for t in b]
    out[8], out[9] = dev(nn_base), dev(nn_rgb)
    return tuple(out)


def hip_activation_masks(pm, dtype=torch.bool):
    """The branch every LeakyReLU of the LAST TRAIN FORWARD took on the HIP side, in OracleModel.act_masks' keys: read off
    the post-activation maps the plan keeps for its backward pass (y > 0 <=> pre-activation > 0 for alpha >= 0; the
    backward kernels test exactly `y > 0`).  Level l: qtmp[l] / fm[l][..., :C] (query), otmp[l][:, j] / obs[l][:, j]
    (observation j); expanding block j: dtmp[j] / dec[j]."""
    (b,) = pm.plan._bufs.values()
    D, U, cl = pm.plan.n_down, pm.plan.n_up, b['C']
    cpu = lambda t: (t > 0).cpu()
    masks = {}
    for l in range(1, D + 1):
        masks[('q', l)] = (cpu(b['qtmp'][l]), cpu(b['fm'][l][..., :cl[l]]))
        for j in range(b['obs'][l].shape[1]):
            masks[('o', l, j)] = (cpu(b['otmp'][l][:, j]), cpu(b['obs'][l][:, j]))
    for j in range(U):
        masks[('q', D + 1 + j)] = (cpu(b['dtmp'][j]), cpu(b['dec'][j]))
    return masks
