"""A00-shaped proposal schedule for the batched engine (bench / test input).

One MCMC iteration of BPP's method A00 (cmd_run loop, method.c:5343-5870) touches
the likelihood path, per locus with n tips, through:

  GAGE  n-1 gene-node age proposals   gtree.c:4585-5532   2-3 branch P-matrices, root-path partials, lnL
  GSPR  2n-2 prune/regraft proposals  gtree.c:6531-7610   3-4 branch P-matrices, 1-2 root paths, lnL
  TAU   one all-loci step per species-tree inner node  stree.c:4338-4779, 5512-6440 (dirty-subtree partials,
        ONE accept/reject for all loci from the summed lnL difference)
  MIX   one all-loci step             prop_mixing.c:52-221 (all ages scaled: every matrix, every partial)

This module plays the *caller* of the hot path for every locus in lock-step ("step j of
every locus" = one batched plan): it keeps the gnode_t fields the reference keeps
(left/right/parent/time/clv_index/scaler_index/pmatrix_index), toggles the double
buffers exactly as the reference does before calling the update API
(SWAP_CLV_INDEX / SWAP_PMAT_INDEX / SWAP_SCALER_INDEX, locus.c:24-26) and toggles back
on rejection.  The MSC prior and the accept/reject rule are host MCMC control and out
of scope: here proposals are drawn from simple valid kernels and accepted
by a seeded coin, which fixes the tape in advance so that the same tape can be replayed
by the GPU engine, by the oracle, and by the real reference (oracle/ref_shim.c
ref_run_tape) and their log-likelihoods compared step by step.
"""
import numpy as np

from .api import OP_DTYPE, SCALE_BUFFER_NONE

P_ACCEPT = 0.3          # typical of BPP's tuned GAGE/GSPR acceptance (finetune target 0.2-0.4)


class TreeState:
    """gene tree of one locus with the reference's buffer-index bookkeeping"""

    def __init__(self, left, right, times, root, scaling=False):
        n = len(left)
        self.n = n
        self.tips = (n + 1) // 2
        self.inner = self.tips - 1
        self.edges = 2 * self.tips - 2
        self.left, self.right = list(left), list(right)
        self.time = [float(t) for t in times]
        self.parent = [-1] * n
        for i in range(n):
            if left[i] >= 0:
                self.parent[left[i]] = i
                self.parent[right[i]] = i
        self.root = root
        self.clv = list(range(n))
        self.pmat = list(range(n))
        self.scaler = [(i - self.tips) if (scaling and i >= self.tips) else SCALE_BUFFER_NONE for i in range(n)]

    # -- the reference's toggles (locus.c:24-26)
    def swap_clv(self, i):
        self.clv[i] = self.tips + (self.clv[i] - self.tips + self.inner) % (2 * self.inner)
        if self.scaler[i] != SCALE_BUFFER_NONE:
            self.scaler[i] = (self.scaler[i] + self.inner) % (2 * self.inner)

    def swap_pmat(self, i):
        self.pmat[i] = (self.pmat[i] + self.edges) % (2 * self.edges)

    def path_to_root(self, i):
        out = []
        while i >= 0:
            out.append(i)
            i = self.parent[i]
        return out

    def op(self, i):
        l, r = self.left[i], self.right[i]
        return (self.clv[i], self.scaler[i], self.clv[l], self.pmat[l], self.scaler[l],
                self.clv[r], self.pmat[r], self.scaler[r])

    def snapshot(self, nodes):
        return [(i, self.left[i], self.right[i], self.parent[i], self.time[i], self.clv[i],
                 self.scaler[i], self.pmat[i]) for i in nodes]

    def restore(self, snap, root):
        for i, l, r, p, t, c, s, m in snap:
            self.left[i], self.right[i], self.parent[i], self.time[i] = l, r, p, t
            self.clv[i], self.scaler[i], self.pmat[i] = c, s, m
        self.root = root

    def record(self, i):
        """state of node i as the reference replay needs it (ref_shim.c ref_run_tape)"""
        return (i, self.left[i], self.right[i], self.parent[i], self.clv[i], self.scaler[i], self.pmat[i],
                self.time[i])

    def swap_ids(self, a, b):
        """exchange the tree positions of node ids a and b (buffer indices stay with the ids):
        what gtree.c:6129-6175 does so that the root node object stays the root"""
        m = lambda x: b if x == a else a if x == b else x
        n = self.n
        left = [m(self.left[m(i)]) if self.left[m(i)] >= 0 else -1 for i in range(n)]
        right = [m(self.right[m(i)]) if self.right[m(i)] >= 0 else -1 for i in range(n)]
        parent = [m(self.parent[m(i)]) if self.parent[m(i)] >= 0 else -1 for i in range(n)]
        time = [self.time[m(i)] for i in range(n)]
        self.left, self.right, self.parent, self.time = left, right, parent, time
        self.root = m(self.root)
        return m

    def inner_nodes(self):
        return [i for i in range(self.n) if self.left[i] >= 0]

    def subtree(self, i):
        out, st = [], [i]
        while st:
            x = st.pop()
            out.append(x)
            if self.left[x] >= 0:
                st += [self.left[x], self.right[x]]
        return out


class Step:
    """one batched proposal step over a subset of loci (explicit indices, ready for bpa_plan)"""
    __slots__ = ("kind", "loci", "mat_off", "mat_pmatrix", "mat_length", "op_off", "ops", "root_clv",
                 "root_scaler", "pre", "post", "global_decision", "_index", "params")

    def __init__(self, kind):
        self.kind = kind
        self.loci = []
        self.mat_off, self.mat_pmatrix, self.mat_length = [0], [], []
        self.op_off, self.ops = [0], []
        self.root_clv, self.root_scaler = [], []
        self.pre, self.post = [], []        # per locus: node records before compute / after (reverts)
        self.global_decision = None
        self._index = None
        # substitution parameters to install BEFORE this step is evaluated: [(which, values[nloci, len])] with
        # which = 1 base frequencies, 2 exchangeabilities, 4 category rates (bpa_plan_set_params); rows of ALL loci
        self.params = []

    def finish(self):
        self.ops = np.array(self.ops, dtype=OP_DTYPE) if self.ops else np.zeros(0, dtype=OP_DTYPE)
        return self


class A00Schedule:
    """builds the tape: `iterations` x (GAGE, GSPR, TAU, MIX) steps for all loci"""

    def __init__(self, trees, rate_mui=None, seed=1, taus=(0.001, 0.002, 0.003), subst=None):
        """subst: dict(freqs[nloci,S], exch[nloci,S(S-1)/2], alpha[nloci], rate_cats, gamma=fn(alpha, cats)) turns on the
        per-locus substitution-parameter proposals of a GTR+Gamma analysis (3 frequency, 5 exchangeability and 1
        alpha move per locus and iteration: locus.c:2782-3419, prop_gamma.c:52-224), each a full recompute"""
        self.trees = trees
        self.nloci = len(trees)
        self.mui = [1.0] * self.nloci if rate_mui is None else list(rate_mui)
        self.rng = np.random.default_rng(seed)
        self.taus = list(taus)
        self.subst = None
        self.revert = set()               # parameter kinds whose current values must be re-installed (rejections)
        self.keep_records = True          # the node records a replay on the REFERENCE's API needs (tests/tape.py): off for launch-only use
        if subst is not None:
            self.subst = dict(gamma=subst["gamma"], rate_cats=int(subst["rate_cats"]))
            self.cur = {1: np.array(subst["freqs"], dtype=np.float64), 2: np.array(subst["exch"], dtype=np.float64),
                        "alpha": np.array(subst["alpha"], dtype=np.float64)}
            self.cur[4] = np.array([self.subst["gamma"](a, self.subst["rate_cats"]) for a in self.cur["alpha"]])

    # ---- helpers
    def _emit(self, step, li, tr, branches, nodes, touched, root_before, snap):
        """toggle buffers, emit descriptors for locus li; returns post (revert) records"""
        for b in branches:
            tr.swap_pmat(b)
        nodes = sorted(set(nodes), key=lambda i: tr.time[i])        # children before parents
        for i in nodes:
            tr.swap_clv(i)
        step.loci.append(li)
        for b in branches:
            step.mat_pmatrix.append(tr.pmat[b])
            step.mat_length.append((tr.time[tr.parent[b]] - tr.time[b]) * self.mui[li])
        step.mat_off.append(len(step.mat_pmatrix))
        for i in nodes:
            step.ops.append(tr.op(i))
        step.op_off.append(len(step.ops))
        step.root_clv.append(tr.clv[tr.root])
        step.root_scaler.append(tr.scaler[tr.root])
        if not self.keep_records:
            return None
        allnodes = sorted(set(touched) | set(branches) | set(nodes))
        step.pre.append(dict(records=[tr.record(i) for i in allnodes], root=tr.root,
                             branches=list(branches), nodes=list(nodes)))
        return allnodes

    def _decide(self, step, tr, snap, root_before, allnodes, accept):
        if not accept:
            tr.restore(snap, root_before)
        if not self.keep_records:
            return
        step.post.append(dict(records=[] if accept else [tr.record(i) for i in allnodes], root=tr.root))

    # ---- GAGE: gtree.c:4585 propose_ages — one inner node per step
    def gage_step(self, k):
        step = Step("GAGE")
        u = self.rng.random((self.nloci, 2))
        for li, tr in enumerate(self.trees):
            inner = tr.inner_nodes()
            if k >= len(inner):
                continue
            v = inner[k]
            lo = max(tr.time[tr.left[v]], tr.time[tr.right[v]])
            p = tr.parent[v]
            snap = tr.snapshot(tr.path_to_root(v) + [tr.left[v], tr.right[v]])
            root_before = tr.root
            if p >= 0:
                tr.time[v] = lo + (0.02 + 0.96 * u[li, 0]) * (tr.time[p] - lo)
            else:
                tr.time[v] = lo + (tr.time[v] - lo) * np.exp(0.6 * (u[li, 0] - 0.5))
            branches = [tr.left[v], tr.right[v]] + ([v] if p >= 0 else [])
            allnodes = self._emit(step, li, tr, branches, tr.path_to_root(v), [v], root_before, snap)
            self._decide(step, tr, snap, root_before, allnodes, u[li, 1] < P_ACCEPT)
        return step.finish()

    # ---- GSPR: gtree.c:6531 propose_spr — one non-root node per step
    def gspr_step(self, k):
        step = Step("GSPR")
        u = self.rng.random((self.nloci, 4))
        for li, tr in enumerate(self.trees):
            cand = [i for i in range(tr.n) if i != tr.root]
            if k >= len(cand):
                continue
            a = cand[k]
            p = tr.parent[a]
            s = tr.left[p] if tr.right[p] == a else tr.right[p]
            g = tr.parent[p]
            root_before = tr.root
            snap = tr.snapshot(range(tr.n))
            # prune: sibling takes p's place
            tr.parent[s] = g
            if g >= 0:
                if tr.left[g] == p:
                    tr.left[g] = s
                else:
                    tr.right[g] = s
            else:
                tr.root = s
            # regraft target: any remaining node not in a's subtree (incl. the remaining root)
            banned = set(tr.subtree(a)) | {p}
            targets = [c for c in range(tr.n) if c not in banned]
            c = targets[int(u[li, 0] * len(targets)) % len(targets)]
            pc = tr.parent[c]
            lo = max(tr.time[a], tr.time[c])
            if pc >= 0:
                if tr.time[pc] <= lo:            # no room on that branch: regraft back onto the sibling
                    c, pc = s, tr.parent[s]
                    lo = max(tr.time[a], tr.time[c])
                tnew = lo + (0.02 + 0.96 * u[li, 1]) * (tr.time[pc] - lo) if pc >= 0 else \
                    lo + (0.1 + u[li, 1]) * max(lo, 1e-4) * 0.5
            else:
                tnew = lo + (0.1 + u[li, 1]) * max(lo, 1e-4) * 0.5
            tr.time[p] = tnew
            tr.left[p], tr.right[p] = a, c
            tr.parent[a] = tr.parent[c] = p
            tr.parent[p] = pc
            if pc >= 0:
                if tr.left[pc] == c:
                    tr.left[pc] = p
                else:
                    tr.right[pc] = p
            else:
                tr.root = p
            nodes = set(tr.path_to_root(p))
            if g >= 0:
                nodes |= set(tr.path_to_root(g))
            bset = [a, c, p, s]
            if tr.root != root_before:
                # the root node keeps its identity (and its never-used P-matrix slot): the new
                # top node and the old root object trade places (gtree.c:6129-6175); the node
                # that lands in the old root's position must be recomputed into its own buffers
                newtop = tr.root
                m = tr.swap_ids(newtop, root_before)
                nodes = {m(x) for x in nodes} | set(tr.path_to_root(newtop))
                bset = [m(x) for x in bset]
            branches = list(dict.fromkeys(x for x in bset if tr.parent[x] >= 0))
            allnodes = self._emit(step, li, tr, branches, nodes, range(tr.n), root_before, snap)
            self._decide(step, tr, snap, root_before, allnodes, u[li, 2] < P_ACCEPT)
        return step.finish()

    # ---- TAU: stree.c:5512 propose_tau — rubber-band rescaling around one species divergence
    def tau_step(self, j):
        step = Step("TAU")
        u = self.rng.random(2)
        taus = self.taus
        tau = taus[j]
        lo = taus[j - 1] if j > 0 else 0.0
        hi = taus[j + 1] if j + 1 < len(taus) else None
        new = lo + (0.05 + 0.9 * u[0]) * ((hi if hi is not None else 2 * tau - lo) - lo)
        accept = bool(u[1] < P_ACCEPT)
        step.global_decision = accept
        for li, tr in enumerate(self.trees):
            moved = []
            snap = tr.snapshot(range(tr.n))
            root_before = tr.root
            for v in tr.inner_nodes():
                t = tr.time[v]
                if lo < t <= tau:
                    tr.time[v] = lo + (t - lo) * (new - lo) / (tau - lo)
                    moved.append(v)
                elif t > tau and (hi is None or t < hi):
                    tr.time[v] = (new + (t - tau)) if hi is None else hi - (hi - t) * (hi - new) / (hi - tau)
                    moved.append(v)
            if not moved:
                continue
            # keep the tree valid whatever the rescaling did
            for v in sorted(tr.inner_nodes(), key=lambda i: tr.time[i]):
                m = max(tr.time[tr.left[v]], tr.time[tr.right[v]])
                if tr.time[v] <= m:
                    tr.time[v] = m * (1 + 1e-9) + 1e-12
                    if v not in moved:
                        moved.append(v)
            branches, nodes = set(), set()
            for v in moved:
                branches |= {tr.left[v], tr.right[v]}
                if tr.parent[v] >= 0:
                    branches.add(v)
                nodes |= set(tr.path_to_root(v))            # gtree_return_partials, gtree.c:145-175
            allnodes = self._emit(step, li, tr, sorted(branches), nodes, moved, root_before, snap)
            self._decide(step, tr, snap, root_before, allnodes, accept)
        if accept:
            self.taus[j] = new
        return step.finish()

    # ---- MIX: prop_mixing.c:52 — all ages times c, everything recomputed
    def mix_step(self):
        step = Step("MIX")
        u = self.rng.random(2)
        c = float(np.exp(0.3 * (u[0] - 0.5)))
        accept = bool(u[1] < P_ACCEPT)
        step.global_decision = accept
        for li, tr in enumerate(self.trees):
            snap = tr.snapshot(range(tr.n))
            root_before = tr.root
            for v in tr.inner_nodes():
                tr.time[v] *= c
            branches = [i for i in range(tr.n) if tr.parent[i] >= 0]
            allnodes = self._emit(step, li, tr, branches, tr.inner_nodes(), [], root_before, snap)
            self._decide(step, tr, snap, root_before, allnodes, accept)
        if accept:
            self.taus = [t * c for t in self.taus]
        return step.finish()

    # ---- FREQ / QRATE / ALPHA: per-locus substitution-parameter proposals, each evaluated by a full recompute
    def _param_step(self, kind, which, proposed, alpha=None):
        step = Step(kind)
        # rejected proposals of OTHER kinds are put back first; this kind's own array already carries current values
        step.params = [(w, self.cur[w].copy()) for w in sorted(self.revert) if w != which] + [(which, proposed)]
        self.revert.clear()
        u = self.rng.random(self.nloci)
        rejected = False
        for li, tr in enumerate(self.trees):
            snap = tr.snapshot(range(tr.n))
            root_before = tr.root
            branches = [i for i in range(tr.n) if tr.parent[i] >= 0]
            allnodes = self._emit(step, li, tr, branches, tr.inner_nodes(), [], root_before, snap)
            accept = bool(u[li] < P_ACCEPT)
            self._decide(step, tr, snap, root_before, allnodes, accept)
            if accept:
                self.cur[which][li] = proposed[li]
                if alpha is not None:
                    self.cur["alpha"][li] = alpha[li]
            else:
                rejected = True
        if rejected:
            self.revert.add(which)
        return step.finish()

    def freq_step(self, k):
        """move mass between frequency k and the last one (locus.c:2782-2900)"""
        f = self.cur[1].copy()
        S = f.shape[1]
        d = 0.04 * (self.rng.random(self.nloci) - 0.5)
        d = np.clip(d, 0.01 - f[:, k], f[:, S - 1] - 0.01)
        f[:, k] += d
        f[:, S - 1] -= d
        return self._param_step("FREQ", 1, f)

    def qrate_step(self, k):
        """multiplier on exchangeability k; the last one stays the reference (locus.c:3100-3250)"""
        q = self.cur[2].copy()
        q[:, k] *= np.exp(0.3 * (self.rng.random(self.nloci) - 0.5))
        return self._param_step("QRATE", 2, q)

    def alpha_step(self):
        """multiplier on alpha; the category rates follow through pll_compute_gamma_cats (prop_gamma.c:52-224)"""
        a = self.cur["alpha"] * np.exp(0.4 * (self.rng.random(self.nloci) - 0.5))
        r = np.array([self.subst["gamma"](x, self.subst["rate_cats"]) for x in a])
        return self._param_step("ALPHA", 4, r, alpha=a)

    def flush_params(self, step):
        """a tree step after rejected parameter proposals: the current values go back in first"""
        if self.revert:
            step.params = [(w, self.cur[w].copy()) for w in sorted(self.revert)]
            self.revert.clear()
        return step

    def initial_step(self):
        """start-up: all matrices, all partials, lnL (method.c:4285-4297) — no toggling"""
        step = Step("INIT")
        for li, tr in enumerate(self.trees):
            step.loci.append(li)
            for b in range(tr.n):
                if tr.parent[b] >= 0:
                    step.mat_pmatrix.append(tr.pmat[b])
                    step.mat_length.append((tr.time[tr.parent[b]] - tr.time[b]) * self.mui[li])
            step.mat_off.append(len(step.mat_pmatrix))
            for i in sorted(tr.inner_nodes(), key=lambda i: tr.time[i]):
                step.ops.append(tr.op(i))
            step.op_off.append(len(step.ops))
            step.root_clv.append(tr.clv[tr.root])
            step.root_scaler.append(tr.scaler[tr.root])
            step.pre.append(dict(records=[tr.record(i) for i in range(tr.n)], root=tr.root,
                                 branches=[b for b in range(tr.n) if tr.parent[b] >= 0],
                                 nodes=sorted(tr.inner_nodes(), key=lambda i: tr.time[i])))
            step.post.append(dict(records=[], root=tr.root))
        return step.finish()

    def iteration(self):
        """the steps of one A00 iteration, in the order of method.c:5490-5602"""
        nmax = max(tr.tips for tr in self.trees)
        steps = [self.gage_step(k) for k in range(nmax - 1)]
        steps += [self.gspr_step(k) for k in range(2 * nmax - 2)]
        if self.subst is not None:
            nex = self.cur[2].shape[1]
            steps += [self.freq_step(k) for k in range(self.cur[1].shape[1] - 1)]
            steps += [self.qrate_step(k) for k in range(nex - 1)]
            if self.subst["rate_cats"] > 1:
                steps.append(self.alpha_step())
        steps += [self.flush_params(self.tau_step(j)) for j in range(len(self.taus))]
        steps.append(self.flush_params(self.mix_step()))
        return [s for s in steps if s.loci]
