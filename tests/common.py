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


def experimental_build():
    """the library was built with -DBPA_EXPERIMENTAL (csrc/experimental/ compiled, the A/B switches of superseded variants alive):
    tests of those variants skip on the default build"""
    import bpp_amd
    return bool(bpp_amd.lib().bpa_experimental_build())


# switches (and BPA_S20_KERNEL values) that only an experimental build reads
EXPERIMENTAL_SWITCHES = ("BPA_GS_FUSEA", "BPA_GS_FUSEPM", "BPA_GS_PINOUT", "BPA_S20_PMGROUP", "BPA_GS_ROOTSTORE", "BPA_GS_NOSPLIT", "BPA_KLANE_V2")
EXPERIMENTAL_S20 = ("waverl", "wave2", "pipe", "generic")


def skip_unless_experimental(setting):
    """setting: 'NAME=value' or a BPA_S20_KERNEL value"""
    import pytest
    name = (setting or "").split("=")[0]
    if (name in EXPERIMENTAL_SWITCHES or setting in EXPERIMENTAL_S20) and not experimental_build():
        pytest.skip(f"{setting}: a variant of csrc/experimental/ (build with BPA_EXPERIMENTAL=1 python -m bpp_amd.build --force)")
