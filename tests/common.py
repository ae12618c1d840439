"""Seeded synthetic inputs shared by the tests (trees, alignments, models)."""
import json
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NT = "ACGT"
AA = "ARNDCQEGHILKMFPSTWYV"


def rand_tree(tips, rng, depth=0.01):
    """random coalescent-like binary tree: tips 0..tips-1, inner nodes after, root last"""
    n = 2 * tips - 1
    left, right, times = [-1] * n, [-1] * n, [0.0] * n
    active, t, nxt = list(range(tips)), 0.0, tips
    while len(active) > 1:
        t += rng.exponential(depth / tips)
        i, j = rng.choice(len(active), 2, replace=False)
        a, b = active[i], active[j]
        left[nxt], right[nxt], times[nxt] = a, b, t
        active = [x for x in active if x not in (a, b)] + [nxt]
        nxt += 1
    return left, right, times, n - 1


def rand_seqs(tips, sites, alphabet, rng, extra="", pmut=0.2):
    alpha = alphabet + extra
    base = rng.integers(0, len(alphabet), sites)
    out = []
    for _ in range(tips):
        s = base.copy()
        m = rng.random(sites) < pmut
        s[m] = rng.integers(0, len(alpha), m.sum())
        out.append("".join(alpha[c] for c in s))
    return out


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def lg_model():
    g = load_golden("lg_model.json")
    return np.array(g["rates"]), np.array(g["freqs"])


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)
