"""Host-side helpers with the names and behaviour of `ns_gym/utils.py` (the non-MuJoCo, non-CLI part of it).

The kernels compute the same quantities on the device (W1 of a distribution update: `csrc/nsg_theta.hip.h`, `w1_n`);
these functions are for user code that called the reference's helpers directly."""
from __future__ import annotations

import math

import numpy as np


def wasserstein_distance(u, v) -> float:
    """1-Wasserstein distance between two pmfs on the indices {0..n-1} (`ns_gym/utils.py:55-94`).

    The reference calls SciPy with `u_values = v_values = arange(n)` and the inputs as weights; SciPy's `_cdf_distance`
    then sums |CDF_u - CDF_v| * delta over the merged support [0, 0, 1, 1, ...], whose deltas alternate 0, 1.  The same
    array - zeros included, so that NumPy's pairwise summation associates identically - is built here: results equal
    SciPy's bit for bit (`tests/test_host_api.py`), without importing SciPy on every call as the reference does.
    Weights need not be normalised; they must have one shape, be finite, non-negative and not sum to zero."""
    a = np.asarray(u, dtype=float)
    b = np.asarray(v, dtype=float)
    if a.shape != b.shape:
        raise ValueError(f"wasserstein_distance: u and v must have the same shape, got {a.shape} vs {b.shape}")
    a, b = a.ravel(), b.ravel()
    if a.size == 0:
        raise ValueError("Distribution can't be empty.")
    for w in (a, b):
        if np.any(w < 0):
            raise ValueError("All weights must be non-negative.")
        if not 0 < np.sum(w) < np.inf:
            raise ValueError("Weight array-like sum must be positive and finite. Set as None for an equal distribution of weight.")
    ca = np.concatenate(([0.0], np.cumsum(a)))
    cb = np.concatenate(([0.0], np.cumsum(b)))
    terms = np.zeros(2 * a.size - 1)
    # merged support point 2k+1 sits between index k and k+1: both CDFs have absorbed weights 0..k there
    terms[1::2] = np.abs(ca[1:-1] / ca[-1] - cb[1:-1] / cb[-1])
    return float(np.sum(terms))


def n_choose_k(n: int, k: int) -> int:
    """Binomial coefficient (`ns_gym/utils.py:41-52`)."""
    return math.factorial(n) // (math.factorial(k) * math.factorial(n - k))


def state_action_update(transitions: list, new_probs: list) -> list:
    """Replace the probabilities of one `P[s][a]` transition list in place and return it (`ns_gym/utils.py:12-38`)."""
    for k, entry in enumerate(transitions):
        transitions[k] = (new_probs[k],) + tuple(entry[1:])
    return transitions


def categorical_sample(probs) -> int:
    """Index drawn from a categorical distribution with NumPy's global generator, like the reference (`ns_gym/utils.py:97-104`)."""
    return np.random.choice(len(probs), p=probs)


def type_mismatch_checker(observation=None, reward=None):
    """NS observation dict -> its `state`, `Reward` -> its scalar; anything else passes through (`ns_gym/utils.py:122-152`)."""
    from .base import Reward

    obs = observation["state"] if isinstance(observation, dict) and "state" in observation else observation
    rew = reward.reward if isinstance(reward, Reward) else reward
    assert not isinstance(obs, dict), "Observation is still a dict after type checking."
    return obs, rew
