"""Helpers to compare tensors with the pinned fixtures under tests/golden."""
import os

import numpy as np
import torch

from oracle import recipe

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def check_pin(store, key, t: torch.Tensor, rtol: float, what: str = ""):
    """Compare ``t`` with the pin ``key``: shape, subsampled values and L2 norm.

    The metric is max|a-b| / max|b| over the pinned values (relative to the
    largest magnitude of the tensor, so entries near zero carry an absolute
    floor), the tolerance named in each test."""
    a = t.detach().to(torch.float64).cpu().numpy().reshape(-1)
    assert tuple(store[key + "/shape"]) == tuple(t.shape), (what, key, "shape")
    idx = recipe.subsample_index(a.size)
    ref = store[key + "/v"].astype(np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(a[idx] - ref).max() / scale
    assert err <= rtol, f"{what} {key}: max rel err {err:.3e} > {rtol:.1e}"
    l2 = float(store[key + "/l2"])
    assert abs(np.sqrt((a * a).sum()) - l2) <= rtol * max(l2, 1e-30) * 4, (what, key, "l2 norm")
    return err
