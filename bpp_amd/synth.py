"""Synthetic multi-locus alignments of the shapes BASELINE.json names:
gene trees drawn from the multispecies coalescent on a fixed species tree (one
sequence per species), sequences evolved down each gene tree under JC69 /
GTR+Gamma / an amino-acid model + Gamma, then site-pattern compression with the
semantics of compress_site_patterns (compress.c:218; JC69 relabel-merge for JC69).

This is the bench/test input generator (the reference's own simulator is DNA-only
and does not travel to the GPU box).  Plain numpy; deterministic for a seed.
"""
import os

import numpy as np

from . import api

NT = "ACGT"
AA = "ARNDCQEGHILKMFPSTWYV"

# species trees as nested tuples (name | (left, right, tau)); theta is global
SPECIES_TREES = {
    # (((A,B):0.001, C):0.002, D):0.003   — SURVEY.md App. B sim.ctl
    4: ((("A", "B", 0.001), "C", 0.002), "D", 0.003),
    # balanced 8 taxa, tau_root 0.005
    8: (((("A", "B", 0.001), ("C", "D", 0.0012), 0.0025), (("E", "F", 0.0011), ("G", "H", 0.0013), 0.003), 0.005)),
    # ((((A,B),C),(D,E)),F), tau_root 0.05 (amino-acid set)
    6: (((("A", "B", 0.01), "C", 0.02), ("D", "E", 0.015), 0.035), "F", 0.05),
}


def species_tree_arrays(taxa, theta=None):
    """(parent, tau, theta) of SPECIES_TREES[taxa] in stree->nodes order: the species in the order
    their names appear (= the gene-tree tip order of make_dataset), then the inner populations with
    children before parents, the root last; one theta for all populations"""
    stree = SPECIES_TREES[taxa]
    if theta is None:
        theta = 0.002 if taxa != 6 else 0.02
    tips = []

    def names(t):
        if isinstance(t, str):
            tips.append(t)
        else:
            names(t[0]); names(t[1])
    names(stree)
    inner = []                      # (tau, left id, right id) in post-order

    def walk(t):
        if isinstance(t, str):
            return tips.index(t)
        l, r = walk(t[0]), walk(t[1])
        inner.append((t[2], l, r))
        return len(tips) + len(inner) - 1
    walk(stree)
    n = len(tips) + len(inner)
    parent, tau = [-1] * n, [0.0] * n
    for k, (tv, l, r) in enumerate(inner):
        me = len(tips) + k
        tau[me] = tv
        parent[l] = parent[r] = me
    return parent, tau, [theta] * n


def _msc_gene_tree(stree, theta, rng):
    """one gene tree under the MSC; returns (left, right, times, root) with tips in
    species order and inner nodes numbered by increasing age"""
    tips = []

    def names(t):
        if isinstance(t, str):
            tips.append(t)
        else:
            names(t[0]); names(t[1])
    names(stree)
    n = len(tips)
    left, right, times = [-1] * (2 * n - 1), [-1] * (2 * n - 1), [0.0] * (2 * n - 1)
    events = []          # (time, a, b)
    tip_id = {name: i for i, name in enumerate(tips)}

    def pop(t, end):
        """coalesce the lineages of subtree t inside its own population up to `end`"""
        if isinstance(t, str):
            lin, start = [tip_id[t]], 0.0
        else:
            lin = pop(t[0], t[2]) + pop(t[1], t[2])
            start = t[2]
        now = start
        while len(lin) > 1:
            k = len(lin)
            now += rng.exponential(theta / (k * (k - 1)))     # rate k(k-1)/theta
            if end is not None and now >= end:
                break
            i, j = rng.choice(k, 2, replace=False)
            events.append((now, lin[i], lin[j]))
            new = ("ev", len(events) - 1)
            lin = [x for q, x in enumerate(lin) if q not in (i, j)] + [new]
        return lin

    pop(stree, None)
    order = sorted(range(len(events)), key=lambda e: events[e][0])
    ident = {}
    for rank, e in enumerate(order):
        ident[e] = n + rank

    def nid(x):
        return x if isinstance(x, int) else ident[x[1]]
    for e in order:
        t, a, b = events[e]
        i = ident[e]
        left[i], right[i], times[i] = nid(a), nid(b), t
    return left, right, times, 2 * n - 2


def _discrete_gamma(alpha, cats):
    return api.compute_gamma_cats(alpha, alpha, cats) if cats > 1 else np.ones(1)


def _q_matrix(freqs, exch):
    S = len(freqs)
    Q = np.zeros((S, S))
    k = 0
    for i in range(S):
        for j in range(i + 1, S):
            Q[i, j] = exch[k] * freqs[j]
            Q[j, i] = exch[k] * freqs[i]
            k += 1
    Q[np.diag_indices(S)] = -Q.sum(1)
    return Q / -(freqs * np.diag(Q)).sum()


def _evolve(left, right, times, root, sites, rng, freqs, Q, site_rate, eig=None):
    """states[node, site].  eig: (lam, V, Vi) of Q when the caller has them (the same Q for every locus)"""
    S = len(freqs)
    n = len(left)
    states = np.zeros((n, sites), dtype=np.int8)
    states[root] = rng.choice(S, sites, p=freqs)
    if eig is None:
        lam, V = np.linalg.eig(Q)
        Vi = np.linalg.inv(V)
    else:
        lam, V, Vi = eig
    # the uniforms of a branch are drawn rate class by rate class (classes by increasing rate): one draw of `sites`
    # numbers dealt out in that order is the same stream
    urates, ridx = np.unique(site_rate, return_inverse=True)
    order = np.argsort(ridx, kind="stable")
    u_site = np.empty(sites)
    cum = np.empty((len(urates), S, S))
    stack = [root]
    while stack:
        p = stack.pop()
        for c in (left[p], right[p]):
            if c < 0:
                continue
            t = times[p] - times[c]
            for k, r in enumerate(urates):
                P = np.real((V * np.exp(lam * t * r)) @ Vi)
                P = np.clip(P, 0, None)
                cum[k] = np.cumsum(P / P.sum(1, keepdims=True), axis=1)
            u_site[order] = rng.random(sites)
            states[c] = np.minimum((u_site[:, None] > cum[ridx, states[p]]).sum(1), S - 1)
            stack.append(c)
    return states


def lg_model():
    """(exchangeabilities[190], frequencies[20]) of the LG model (data table)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "lg_model.json")) as f:
        g = json.load(f)
    return np.array(g["rates"]), np.array(g["freqs"])


def scaled_tree(t, f):
    """SPECIES_TREES entry with every divergence time x f"""
    return t if isinstance(t, str) else (scaled_tree(t[0], f), scaled_tree(t[1], f), t[2]*f)


def make_dataset(nloci, sites, taxa=4, model="jc69", rate_cats=1, alpha=0.5, theta=None, seed=12345,
                 freqs=None, exch=None, divergence=1.0):
    """returns a list of loci: dict(seqs (compressed), weights, left, right, times, root, ...).
    divergence: every tau of SPECIES_TREES[taxa] and the default theta times this factor (more substitutions per site:
    more distinct site patterns per locus)"""
    cached = _cache_file(nloci, sites, taxa, model, rate_cats, alpha, theta, seed, divergence) if freqs is None and exch is None else None
    if cached and os.path.exists(cached):
        import pickle
        with open(cached, "rb") as f:
            return pickle.load(f)
    rng = np.random.default_rng(seed)
    stree = scaled_tree(SPECIES_TREES[taxa], divergence)
    dna = model in ("jc69", "gtr")
    S = 4 if dna else 20
    if theta is None:
        theta = (0.002 if dna else 0.02)*divergence
    if model == "jc69":
        freqs, exch = np.full(4, 0.25), np.ones(6)
    elif model == "gtr":
        freqs = np.array([0.3, 0.2, 0.2, 0.3]) if freqs is None else np.asarray(freqs)
        exch = np.array([1, 2, 1, 0.5, 1.5, 1.0]) if exch is None else np.asarray(exch)
    else:
        if freqs is None or exch is None:
            exch, freqs = lg_model()
        freqs, exch = np.asarray(freqs), np.asarray(exch)
    Q = _q_matrix(freqs, exch)
    rates = _discrete_gamma(alpha, rate_cats)
    alphabet = NT if dna else AA
    letters = np.frombuffer(alphabet.encode(), dtype="S1")
    lam, V = np.linalg.eig(Q)
    eig = (lam, V, np.linalg.inv(V))
    out = []
    for _ in range(nloci):
        left, right, times, root = _msc_gene_tree(stree, theta, rng)
        site_rate = rates[rng.integers(0, rate_cats, sites)]
        st = _evolve(left, right, times, root, sites, rng, freqs, Q, site_rate, eig)
        seqs = [letters[st[i]].tobytes().decode() for i in range(taxa)]
        pats, w = api.compress_site_patterns(seqs, dna, model == "jc69")
        out.append(dict(seqs=pats, weights=w, left=left, right=right, times=times, root=root,
                        states=S, rate_cats=rate_cats, model=model, freqs=freqs, exch=exch,
                        rates=rates, raw_sites=sites))
    return out


def _cache_file(*key):
    """BPP_AMD_SYNTH_CACHE=<dir>: data sets made ahead of time (precompute, below) are read from there — the GPU test
    session starts the 10 000-locus sets of the full-size tests in background processes while the other tests run"""
    d = os.environ.get("BPP_AMD_SYNTH_CACHE")
    if not d:
        return None
    import hashlib
    return os.path.join(d, "synth_" + hashlib.sha1(repr(key).encode()).hexdigest()[:20] + ".pkl")


def precompute(nloci, sites, taxa, model, rate_cats, seed, divergence=1.0):
    """make one data set and leave it in the cache directory (called in a process of its own: tests/conftest.py)"""
    import pickle
    path = _cache_file(nloci, sites, taxa, model, rate_cats, 0.5, None, seed, divergence)
    if not path or os.path.exists(path):
        return
    os.environ.pop("BPP_AMD_SYNTH_CACHE")
    data = make_dataset(nloci, sites, taxa, model, rate_cats, seed=seed, divergence=divergence)
    with open(path + ".tmp%d" % os.getpid(), "wb") as f:
        pickle.dump(data, f, protocol=pickle.HIGHEST_PROTOCOL)
    os.replace(path + ".tmp%d" % os.getpid(), path)


def msc_start_tree(tip_species, parent, tau, theta, rng):
    """a gene tree drawn from the MSC for tips with the given species (several per species allowed): tips
    0..n-1, inner nodes by increasing age, root last"""
    npop, n = len(parent), len(tip_species)
    kids = {p: [c for c in range(npop) if parent[c] == p] for p in range(npop)}
    events = []

    def run(p):
        lin = [k for k in range(n) if tip_species[k] == p]
        for c in kids[p]:
            lin += run(c)
        now, end = tau[p], (tau[parent[p]] if parent[p] >= 0 else None)
        while len(lin) > 1:
            k = len(lin)
            now += rng.exponential(theta[p] / (k * (k - 1)))
            if end is not None and now >= end:
                break
            i, j = rng.choice(k, 2, replace=False)
            events.append((now, lin[i], lin[j]))
            lin = [x for q, x in enumerate(lin) if q not in (i, j)] + [("ev", len(events) - 1)]
        return lin

    run(npop - 1)
    order = sorted(range(len(events)), key=lambda e: events[e][0])
    ident = {e: n + rank for rank, e in enumerate(order)}
    nid = lambda x: x if isinstance(x, int) else ident[x[1]]
    left, right, times = [-1] * (2 * n - 1), [-1] * (2 * n - 1), [0.0] * (2 * n - 1)
    for e in order:
        t, a, b = events[e]
        left[ident[e]], right[ident[e]], times[ident[e]] = nid(a), nid(b), t
    return left, right, times, 2 * n - 2
