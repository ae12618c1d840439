"""usedata = 0 with ALL moves on: the joint is p(taus) p(thetas) p(G | taus, thetas), so every theta's marginal is its gamma
prior and the root tau's its gamma prior, exactly — whatever the loci.  Checks the device samplers' THETA / TAU / MIX moves
(windows, rubber-band Jacobian, mixing factor, Gibbs draws and theta re-draws of the program's moves) against those."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape


def batch_se(x, nb=40):
    m = len(x) // nb
    bm = np.array([x[i*m:(i+1)*m].mean() for i in range(nb)])
    return bm.std(ddof=1) / np.sqrt(nb)


def run(taxa, model, R, nloci, mode, iters, thin):
    data = synth.make_dataset(nloci, 100, taxa, model, R, seed=3)
    eng = bpp_amd.Engine(0)
    eng.set_options(usedata=0, bfbeta=1.0)
    dev = bpp_amd.Sampler(eng, tape.make_engine_loci(eng, data), data, seed=17)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    a_th, b_th = 3.0, 3.0 / 0.002
    a_tau, b_tau = 4.0, 4.0 / tau0[-1]
    dev.set_species_tree(parent, tau0, thetas)
    dev.set_tau_prior(a_tau, b_tau)
    dev.set_theta_prior(a_th, b_th, 0.002)
    dev.set_finetune(0.004, 0.004, 0.5 * tau0[-1], 0.6)
    if mode >= 1:
        dev.set_proposal_kernel(1)
    if mode == 2:
        dev.set_program_moves(True, 0.1)
    dev.initialize()
    dev.iterate(2000)
    S = []
    for _ in range(iters):
        dev.iterate(thin)
        S.append(dev.thetas() + dev.taus())
    S = np.array(S)
    npop = len(parent)
    out = []
    worst = 0.0
    for p in range(npop):
        x = S[:, p]
        if x.std() < 1e-12:
            continue
        se = batch_se(x)
        z = (x.mean() - a_th / b_th) / se
        rs = x.std() / (np.sqrt(a_th) / b_th)
        worst = max(worst, abs(z))
        out.append(f"theta[{p}] mean {x.mean():.6f} (prior {a_th/b_th:.6f}, z {z:+.1f}) sd ratio {rs:.3f}")
    x = S[:, npop + npop - 1]
    se = batch_se(x)
    z = (x.mean() - a_tau / b_tau) / se
    worst = max(worst, abs(z))
    out.append(f"tau_root mean {x.mean():.6f} (prior {a_tau/b_tau:.6f}, z {z:+.1f}) sd ratio {x.std()/(np.sqrt(a_tau)/b_tau):.3f}")
    sm = dev.summary()
    print(f"--- {taxa} taxa {model} x{nloci} loci, kind {dev.kind()}, mode {('uniform', 'bpp kernel', 'program moves')[mode]}: acceptance {sm['accepted']/sm['proposals']:.3f}, worst |z| {worst:.1f}")
    for o in out:
        print("   ", o)
    dev.close(); eng.close()
    return worst


if __name__ == "__main__":
    w = []
    for mode in (0, 1, 2):
        w.append(run(4, "jc69", 1, 3, mode, 6000, 5))
    w.append(run(8, "gtr", 4, 3, 0, 1500, 2))
    print("worst |z| over all runs:", max(w))
