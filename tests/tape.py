"""Replay of an A00 proposal tape (bpp_amd/schedule.py) by the three evaluators:
the GPU engine (product), the oracle restatement and the real reference
(oracle/ref_shim.c ref_run_tape).  Test/bench-baseline infrastructure."""
import ctypes as C
import numpy as np

import bpp_amd
from bpp_amd.schedule import A00Schedule, TreeState
import oraclelib as O

REC_DTYPE = np.dtype([("node", "<i4"), ("left", "<i4"), ("right", "<i4"), ("parent", "<i4"),
                      ("clv", "<i4"), ("scaler", "<i4"), ("pmat", "<i4"), ("pad", "<i4"),
                      ("time", "<f8")])


def make_engine_loci(engine, data, scaling=False):
    """device loci for a synth dataset (buffer counts of method.c:4110-4146)"""
    loci = []
    for d in data:
        S, R = d["states"], d["rate_cats"]
        tips, sites = len(d["seqs"]), len(d["seqs"][0])
        inner, edges = tips - 1, 2 * tips - 2
        dtype = bpp_amd.DATA_DNA if S == 4 else bpp_amd.DATA_AA
        mdl = {"jc69": bpp_amd.MODEL_JC69, "gtr": bpp_amd.MODEL_GTR}.get(d["model"], bpp_amd.MODEL_LG)
        loc = bpp_amd.Locus(engine, dtype, mdl, tips, 2 * inner, S, sites, 1, 2 * edges, R,
                            2 * inner if scaling else 0)
        for i, s in enumerate(d["seqs"]):
            loc.set_tip_states(i, s)
        loc.set_pattern_weights(d["weights"])
        if d["model"] != "jc69":
            loc.set_frequencies(0, d["freqs"])
            loc.set_subst_params(0, d["exch"])
        loc.set_category_rates(d["rates"])
        loci.append(loc)
    return loci


def make_schedule(data, seed=1, scaling=False, taus=None, subst=False):
    """subst: add the per-locus frequency / exchangeability / alpha proposals of a GTR(+Gamma) analysis"""
    trees = [TreeState(d["left"], d["right"], d["times"], d["root"], scaling) for d in data]
    kw = {} if taus is None else {"taus": taus}
    if subst:
        R = data[0]["rate_cats"]
        kw["subst"] = dict(freqs=[d["freqs"] for d in data], exch=[d["exch"] for d in data],
                           alpha=[0.5] * len(data), rate_cats=R,
                           gamma=lambda a, cats: bpp_amd.compute_gamma_cats(a, a, cats) if cats > 1 else np.ones(1))
    return A00Schedule(trees, seed=seed, **kw)


def apply_params(all_loci_plan, step):
    """the step's substitution-parameter installs (rows for ALL loci) through a plan that holds every locus"""
    for which, vals in step.params:
        all_loci_plan.set_params(which, vals)


def plan_for_step(engine, loci, step):
    return bpp_amd.Plan(engine, [loci[i] for i in step.loci], step.mat_off, step.mat_pmatrix,
                        step.mat_length, step.op_off, step.ops, step.root_clv, step.root_scaler)


def root_lnl_all(engine, loci, sch):
    """the root term of every locus on the schedule's current trees as ONE batched launch with empty matrix and
    update lists: what locus_root_loglikelihood (locus.c:2573) returns for the CLVs that are there"""
    n = len(loci)
    zeros = np.zeros(n + 1, dtype=np.uint32)
    p = bpp_amd.Plan(engine, list(loci), zeros, np.zeros(0, dtype=np.uint32), np.zeros(0), zeros,
                     np.zeros(0, dtype=bpp_amd.OP_DTYPE),
                     np.array([t.clv[t.root] for t in sch.trees], dtype=np.uint32),
                     np.array([t.scaler[t.root] for t in sch.trees], dtype=np.int32))
    p.launch()
    v = p.lnl()
    p.close()
    return v


def locus_subtape(steps, li):
    """the steps that involve locus li, flattened for ref_run_tape / oracle replay"""
    out = []
    carried = []                  # installs of steps this locus sits out (TAU): they ride on its next step
    for si, st in enumerate(steps):
        idx = getattr(st, "_index", None)
        if idx is None:
            idx = {l: i for i, l in enumerate(st.loci)}
            try:
                st._index = idx
            except AttributeError:
                pass
        if li not in idx:
            carried += [(w, v[li]) for w, v in st.params]
        if li in idx:
            t = idx[li]
            out.append(dict(kind=st.kind, pre=st.pre[t], post=st.post[t], params=carried + [(w, v[li]) for w, v in st.params],
                            mat=(st.mat_pmatrix[st.mat_off[t]:st.mat_off[t + 1]],
                                 st.mat_length[st.mat_off[t]:st.mat_off[t + 1]]),
                            ops=st.ops[st.op_off[t]:st.op_off[t + 1]],
                            root_clv=st.root_clv[t], root_scaler=st.root_scaler[t], task=t, step=si))
            carried = []
    return out


def oracle_replay(d, sub, scaling=False):
    """explicit-index replay with the oracle kernels; returns lnL per step"""
    S, R = d["states"], d["rate_cats"]
    ol = O.OracleLocus(S, R, d["seqs"], d["weights"], model=d["model"],
                       freqs=None if d["model"] == "jc69" else d["freqs"],
                       qrates=None if d["model"] == "jc69" else d["exch"], rates=d["rates"],
                       scaling=scaling)
    tips = len(d["seqs"])
    clv = {i: ol.clv[i] for i in range(tips)}
    pm, sc, out = {}, {}, []
    for s in sub:
        for which, v in s.get("params", ()):
            if which == 1:
                ol.freqs = np.asarray(v, float)
            elif which == 2:
                ol.qrates = np.asarray(v, float)
            else:
                ol.rates = np.asarray(v, float)
            if which in (1, 2):
                ol.eig = O.orc_eigen(ol.freqs, ol.qrates)
        for p, t in zip(*s["mat"]):
            pm[p] = ol.pmatrix(t)
        for op in s["ops"]:
            pc, ps, lc, lp, ls, rc, rp, rs = [int(x) for x in op]
            clv[pc], scal = O.orc_partial(clv[lc], clv[rc], pm[lp], pm[rp],
                                          sc.get(ls) if ls >= 0 else None, sc.get(rs) if rs >= 0 else None,
                                          ps >= 0, ol.order)
            if ps >= 0:
                sc[ps] = scal
        out.append(O.orc_lnl(clv[s["root_clv"]], ol.freqs, ol.rw, ol.weights,
                             sc.get(s["root_scaler"]) if s["root_scaler"] >= 0 else None, ol.order))
    return np.array(out)


def _recs(lst):
    a = np.zeros(len(lst), dtype=REC_DTYPE)
    for i, (node, l, r, p, clv, scaler, pmat, time) in enumerate(lst):
        a[i] = (node, l, r, p, clv, scaler, pmat, 0, time)
    return a


def ref_tape_arrays(sub):
    pre_off, post_off, br_off, op_off = [0], [0], [0], [0]
    pre, post, pre_root, post_root, br, opn = [], [], [], [], [], []
    par_off, par_which, par_voff, par_val = [0], [], [], []
    for s in sub:
        for which, v in s.get("params", ()):
            par_which.append(which)
            par_voff.append(len(par_val))
            par_val += [float(x) for x in v]
        par_off.append(len(par_which))
        pre += s["pre"]["records"]
        pre_off.append(len(pre))
        pre_root.append(s["pre"]["root"])
        post += s["post"]["records"]
        post_off.append(len(post))
        post_root.append(s["post"]["root"])
        br += s["pre"]["branches"]
        br_off.append(len(br))
        opn += s["pre"]["nodes"]
        op_off.append(len(opn))
    u = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    return dict(n=len(sub), pre_off=u(pre_off), pre=_recs(pre), pre_root=i32(pre_root),
                post_off=u(post_off), post=_recs(post), post_root=i32(post_root),
                br_off=u(br_off), br=u(br), op_off=u(op_off), op=u(opn),
                par_off=u(par_off), par_which=i32(par_which), par_voff=u(par_voff),
                par_val=np.ascontiguousarray(par_val, dtype=np.float64), has_params=bool(par_which))


def ref_locus_for(d, scaling=False, arch=O.ARCH_AVX2):
    rl = O.RefLocus(d["states"], d["rate_cats"], d["seqs"], d["weights"],
                    model={"jc69": "jc69", "gtr": "gtr"}.get(d["model"], "lg"),
                    freqs=None if d["model"] == "jc69" else d["freqs"],
                    qrates=None if d["model"] == "jc69" else d["exch"],
                    rates=d["rates"], scaling=scaling, arch=arch)
    rl.set_tree(d["left"], d["right"], d["times"], d["root"])
    return rl


def ref_replay(rl, arrays, repeats=1):
    """run the tape through the reference's own update API; returns (lnl per step, seconds)"""
    L = O.ref()
    a = arrays
    out = np.zeros(a["n"])
    up, ip = C.POINTER(C.c_uint), C.POINTER(C.c_int)
    vp = C.c_void_p
    dp = C.POINTER(C.c_double)
    L.ref_run_tape_params.restype = C.c_double
    par = a.get("has_params", False)
    secs = L.ref_run_tape_params(rl.h, a["n"], a["pre_off"].ctypes.data_as(up), a["pre"].ctypes.data_as(vp),
                                 a["pre_root"].ctypes.data_as(ip), a["post_off"].ctypes.data_as(up),
                                 a["post"].ctypes.data_as(vp), a["post_root"].ctypes.data_as(ip),
                                 a["br_off"].ctypes.data_as(up), a["br"].ctypes.data_as(up),
                                 a["op_off"].ctypes.data_as(up), a["op"].ctypes.data_as(up),
                                 a["par_off"].ctypes.data_as(up) if par else None,
                                 a["par_which"].ctypes.data_as(ip) if par else None,
                                 a["par_voff"].ctypes.data_as(up) if par else None,
                                 a["par_val"].ctypes.data_as(dp) if par else None,
                                 out.ctypes.data_as(dp), repeats)
    return out, secs


ORC_OP_DTYPE = bpp_amd.OP_DTYPE


def oracle_tape_run(d, sub, repeats=1, scaling=False):
    """the same tape through oracle.c's own C loop (orc_run_tape): (lnl per step, seconds)"""
    L = O.oracle()
    L.orc_run_tape.restype = C.c_double
    S, R = d["states"], d["rate_cats"]
    tips, sites = len(d["seqs"]), len(d["seqs"][0])
    inner, edges = tips - 1, 2 * tips - 2
    dna = S == 4
    clv = [O.orc_tipclv(S, R, s, dna) for s in d["seqs"]] + [np.zeros((sites, R, S)) for _ in range(2 * inner)]
    pm = [np.zeros((R, S, S)) for _ in range(2 * edges)]
    sc = [np.zeros(sites, dtype=np.uint32) for _ in range(2 * inner)]
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint)
    pclv = (dp * len(clv))(*[a.ctypes.data_as(dp) for a in clv])
    ppm = (dp * len(pm))(*[a.ctypes.data_as(dp) for a in pm])
    psc = (up * len(sc))(*[a.ctypes.data_as(up) for a in sc])
    jc = d["model"] == "jc69"
    freqs = np.full(4, 0.25) if jc else np.ascontiguousarray(d["freqs"], dtype=np.float64)
    if jc:
        ev = iev = evals = np.zeros(1)
    else:
        ev, iev, evals = O.orc_eigen(freqs, d["exch"])
    rates = np.ascontiguousarray(d["rates"], dtype=np.float64)
    rw = np.full(R, 1.0 / R)
    w = np.ascontiguousarray(d["weights"], dtype=np.uint32)
    mat_off, mp, ml, op_off, ops, rc, rs = [0], [], [], [0], [], [], []
    for s in sub:
        mp += list(s["mat"][0]); ml += list(s["mat"][1]); mat_off.append(len(mp))
        ops += [tuple(int(x) for x in o) for o in s["ops"]]; op_off.append(len(ops))
        rc.append(s["root_clv"]); rs.append(s["root_scaler"])
    u = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    mat_off, mp, op_off, rc = u(mat_off), u(mp), u(op_off), u(rc)
    ml = np.ascontiguousarray(ml, dtype=np.float64)
    rs = np.ascontiguousarray(rs, dtype=np.int32)
    ops = np.array(ops, dtype=ORC_OP_DTYPE) if ops else np.zeros(0, dtype=ORC_OP_DTYPE)
    out = np.zeros(len(sub))
    order = O.ORDER_PAIR if S == 4 else O.ORDER_FMA4
    secs = L.orc_run_tape(S, sites, R, int(jc), order, pclv, ppm, psc, rates.ctypes.data_as(dp),
                          rw.ctypes.data_as(dp), freqs.ctypes.data_as(dp), evals.ctypes.data_as(dp),
                          ev.ctypes.data_as(dp), iev.ctypes.data_as(dp), w.ctypes.data_as(up), len(sub),
                          mat_off.ctypes.data_as(up), mp.ctypes.data_as(up), ml.ctypes.data_as(dp),
                          op_off.ctypes.data_as(up), ops.ctypes.data_as(C.c_void_p),
                          rc.ctypes.data_as(up), rs.ctypes.data_as(C.POINTER(C.c_int)),
                          out.ctypes.data_as(dp), repeats)
    return out, secs
