"""Helpers shared by the golden-vector tests (fixtures are produced by oracle/ref_harness/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), 'golden')
MAXN = 8192


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def check_sub(g, name, t, rtol=1e-4, atol=1e-4):
    """Compare tensor `t` with the stored subsample/checksums of golden entry `name`."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    assert tuple(a.shape) == tuple(g[name + '.shape']), (name, a.shape, g[name + '.shape'])
    flat = a.reshape(-1).astype(np.float32)
    stride = int(g[name + '.stride'])
    ref = g[name + '.sample']
    got = flat[::stride]
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f'{name}: max err {err.max():.3e} (ref scale {np.abs(ref).max():.3e})'
    asum = float(g[name + '.asum'])
    assert abs(np.abs(flat.astype(np.float64)).sum() - asum) <= 1e-4 * asum + 1e-3, name
    return float(err.max())


def seeded_sd(g):
    from nuhtc_amd import weights
    sd = weights.seeded_state_dict(int(g['seed']))
    sd['roi_head.semantic_head.conv_logits.bias'] = torch.tensor([float(g['sem_bias'])])
    for key in g.files:                                   # tensors the generator replaced (five-class head)
        if key.startswith('override.'):
            sd[key[len('override.'):]] = torch.from_numpy(g[key])
    for k in range(3):
        sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] = sd[f'roi_head.bbox_head.{k}.fc_cls.bias'] + torch.from_numpy(g['cls_bias_add'])
    return sd


def canon_rows(d):
    """(n,5) [x1,y1,x2,y2,score] rows sorted by score desc, exact ties broken by coordinates."""
    d = np.asarray(d)
    if len(d) == 0:
        return d
    key = np.lexsort((d[:, 3], d[:, 2], d[:, 1], d[:, 0], -d[:, 4]))
    return d[key]


def check_prob_vs_logits(g, name, prob, atol=1e-6):
    """sigmoid(stored logit subsample) vs the same subsample of a probability tensor."""
    a = prob.detach().cpu().numpy() if isinstance(prob, torch.Tensor) else np.asarray(prob)
    assert tuple(a.shape) == tuple(g[name + '.shape']), (name, a.shape, g[name + '.shape'])
    got = a.reshape(-1)[::int(g[name + '.stride'])]
    ref = 1.0 / (1.0 + np.exp(-g[name + '.sample'].astype(np.float64)))
    err = np.abs(got - ref).max() if got.size else 0.0
    assert err <= atol, f'{name}: max prob err {err:.3e}'
    return float(err)
