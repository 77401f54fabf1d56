"""Host-side helpers of the BC loop, same contracts as reference src/utils_bc.py:5-49."""
import pickle
import random

import numpy as np


def is_essential_save(epoch, max_epochs, eval_frequency):
    """True inside +-5 evaluation periods around 1 %, 10 %, 50 % and 97 % of training (utils_bc.py:5-12)."""
    window = 5 * eval_frequency
    return any(int(frac * max_epochs) - window <= epoch < int(frac * max_epochs) + window
               for frac in (0.01, 0.1, 0.5, 0.97))


def ranks(sample):
    order = sorted(range(len(sample)), key=sample.__getitem__)
    out = [0] * len(sample)
    for rank, idx in enumerate(order):
        out[idx] = rank
    return out


def sample_with_minimum_distance(n=40, k=4, d=10, rng=random):
    """k start indices from range(n), pairwise at least d apart (utils_bc.py:24-29).  Consumes the Python
    `random` stream exactly like the reference (one random.sample call), so seeds reproduce its batches."""
    sample = rng.sample(range(n - (k - 1) * (d - 1)), k)
    return [s + (d - 1) * r for s, r in zip(sample, ranks(sample))]


def gather_unrolls(arrays, starting_i, unroll_length, n_samples):
    """main_bc_2.py:194-201: for each start index take unroll_length consecutive rows (wrapping), stack on axis 1."""
    idx = np.mod(np.asarray(starting_i)[None, :] + np.arange(unroll_length)[:, None], n_samples)   # (T,B)
    return [a[idx] for a in arrays]


def read_habitat_data(data_path):
    """Per-scene trajectory pickle -> concatenated arrays (utils_bc.py:33-49)."""
    print('loading %s ...' % data_path)
    with open(data_path, 'rb') as f:
        data = pickle.load(f)
    n_trajectories = len(data['reward'])
    for k in ('obs', 'action', 'reward', 'done', 'true_state'):
        data[k] = np.concatenate(data[k])
    n_samples = len(data['reward'])
    print('  ', '%d trajectories for a total of %d samples' % (n_trajectories, n_samples))
    print('  ', 'avg. return is', data['reward'].sum() / n_trajectories)
    return data


def shard_bounds(n, rank, world):
    """Contiguous row range [lo, hi) of rank `rank`: concatenating shard outputs in rank order reproduces the
    reference's row order."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi
