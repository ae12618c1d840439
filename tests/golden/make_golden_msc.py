"""Golden vectors for the MSC density: the REAL reference's gtree_logprob (gtree.c:3957, through
oracle/ref_shim_input.c: ref_msc_logpr) on gene trees drawn from the multispecies coalescent.
    python tests/golden/make_golden_msc.py   ->  tests/golden/msc_density.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bpp_amd import synth          # noqa: E402


def ref_logpr(L, parent, tau, theta, tips, left, right, time, pop):
    np_ = len(parent)
    n = 2 * tips - 1
    contrib = (C.c_double * np_)()
    L.ref_msc_logpr.restype = C.c_double
    L.ref_msc_logpr.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                C.POINTER(C.c_double)]
    v = L.ref_msc_logpr((np_ + 1) // 2, (C.c_int * np_)(*parent), (C.c_double * np_)(*tau), (C.c_double * np_)(*theta),
                        tips, (C.c_int * n)(*left), (C.c_int * n)(*right), (C.c_double * n)(*time), (C.c_int * n)(*pop),
                        contrib)
    return v, list(contrib)


def pops_of(parent, tau, tips_species, left, right, time):
    """population of every gene node: climb from the common population of the children"""
    np_ = len(parent)
    anc = [set() for _ in range(np_)]
    for p in range(np_):
        q = p
        while q >= 0:
            anc[p].add(q)
            q = parent[q]
    n = len(left)
    pop = list(tips_species) + [-1] * (n - len(tips_species))
    for v in sorted(range(len(tips_species), n), key=lambda k: time[k]):
        c = pop[left[v]]
        while c not in anc[pop[right[v]]]:
            c = parent[c]
        assert time[v] >= tau[c]
        while parent[c] >= 0 and tau[parent[c]] <= time[v]:
            c = parent[c]
        pop[v] = c
    return pop


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbppref.so"))
    rng = np.random.default_rng(2024)
    cases = []
    for taxa in (4, 8, 6):
        parent, tau, _ = synth.species_tree_arrays(taxa)
        for rep in range(6):
            theta = [float(x) for x in rng.uniform(0.0005, 0.02, len(parent))]
            left, right, time, root = synth._msc_gene_tree(synth.SPECIES_TREES[taxa], float(np.mean(theta)), rng)
            pop = pops_of(parent, tau, list(range(taxa)), left, right, time)
            v, contrib = ref_logpr(L, parent, tau, theta, taxa, left, right, time, pop)
            cases.append(dict(parent=parent, tau=[x.hex() for x in tau], theta=[x.hex() for x in theta], tips=taxa,
                              tip_species=list(range(taxa)), left=left, right=right,
                              time=[float(x).hex() for x in time], root=root, pop=pop, logpr=float(v).hex(),
                              contrib=[float(x).hex() for x in contrib]))
    # several sequences per species: a 3-species tree with 2+3+1 sequences, coalescent drawn by hand
    parent, tau, theta = [3, 3, 4, 4, -1], [0, 0, 0, 0.002, 0.005], [0.003, 0.001, 0.002, 0.004, 0.006]
    tip_species = [0, 0, 1, 1, 1, 2]
    left = [-1] * 6 + [0, 2, 7, 6, 9]
    right = [-1] * 6 + [1, 3, 4, 8, 5]
    time = [0.0] * 6 + [0.0011, 0.0004, 0.0031, 0.0042, 0.0093]
    pop = pops_of(parent, tau, tip_species, left, right, time)
    v, contrib = ref_logpr(L, parent, tau, theta, 6, left, right, time, pop)
    cases.append(dict(parent=parent, tau=[float(x).hex() for x in tau], theta=[float(x).hex() for x in theta], tips=6,
                      tip_species=tip_species, left=left, right=right, time=[float(x).hex() for x in time], root=10,
                      pop=pop, logpr=float(v).hex(), contrib=[float(x).hex() for x in contrib]))
    with open(os.path.join(HERE, "msc_density.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
