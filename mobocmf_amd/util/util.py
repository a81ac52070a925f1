"""Helpers of mobocmf/util/util.py used by the path (same names and results)."""
import os

import numpy as np
import torch


def triu_indices(n, offset=0):
    """util.py:27-30 -- returns a 2 x K index tensor (used to index ROWS at mfdgp.py:144)."""
    rows, cols = torch.triu_indices(n, n, offset=offset)
    return torch.stack((rows, cols), dim=0)


def compute_dist(x):
    """util.py:32-33 -- squared distances by the expanded form."""
    return torch.sum(x ** 2, 1, keepdims=True) - 2.0 * x.mm(x.T) + torch.sum(x ** 2, 1, keepdims=True).T


def create_path(folder):
    if not os.path.exists(folder):
        os.makedirs(folder)


def save_pickle(folder, filename, content):
    import dill
    create_path(folder)
    with open(os.path.join(folder, filename), "wb") as fw:
        dill.dump(content, fw)


def read_pickle(folder, filename):
    import dill
    with open(os.path.join(folder, filename), "rb") as fr:
        return dill.load(fr)


def preprocess_outputs(*args):
    """util.py:36-51 -- identity standardisation (mean 0 / std 1) kept as in the reference."""
    y_mean, y_std = 0.0, 1.0
    y_train = [torch.from_numpy((y - y_mean) / y_std).double() for y in args]
    y_train.extend([y_mean, y_std])
    return y_train


def reset_random_state(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
