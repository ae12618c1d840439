"""persistent iteration kernel (sweep2.hpp) against the one-launch-per-step path (BPA_SMP_V1=1), proposal by proposal:
same per-locus streams, so every tree must agree after every prefix of the proposal list"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bpp_amd
from bpp_amd import synth
import tape


def make(eng, data, taxa, seed, v1, steps=None, nomix=False):
    for k in ("BPA_SMP_V1", "BPA_SMP_STEPS", "BPA_SMP_NOMIX"):
        os.environ.pop(k, None)
    if v1:
        os.environ["BPA_SMP_V1"] = "1"
    if steps is not None:
        os.environ["BPA_SMP_STEPS"] = "%d,%d" % steps
    if nomix:
        os.environ["BPA_SMP_NOMIX"] = "1"
    loci = tape.make_engine_loci(eng, data)
    smp = bpp_amd.Sampler(eng, loci, data, seed=seed)
    parent, tau0, thetas = synth.species_tree_arrays(taxa)
    smp.set_species_tree(parent, tau0, thetas)
    smp.set_tau_prior(3.0, 3.0 / tau0[-1])
    smp.set_theta_prior(2.0, 1000.0, 0.001)
    smp.set_finetune(0.003, 0.005, 0.0008, 0.2)
    smp.initialize()
    return smp, loci


def compare(a, b, nloci, tag, limit=4):
    bad = 0
    for i in range(nloci):
        x, y = a.tree(i), b.tree(i)
        diff = [k for k in ("left", "right", "parent", "clv", "pmat", "pop") if [int(v) for v in x[k]] != [int(v) for v in y[k]]]
        if x["root"] != y["root"]:
            diff.append("root")
        if not np.allclose(x["time"], y["time"], rtol=1e-13, atol=0):
            diff.append("time")
        if abs(x["lnl"] - y["lnl"]) > 1e-9 * abs(y["lnl"]):
            diff.append("lnl")
        if abs(x["logpr"] - y["logpr"]) > 1e-9 * abs(y["logpr"]):
            diff.append("logpr")
        if diff:
            bad += 1
            if bad <= limit:
                print(f"  [{tag}] locus {i}: differs in {diff}")
                for k in diff:
                    print(f"     new {k}: {x[k]}\n     old {k}: {y[k]}")
    sa, sb = a.summary(), b.summary()
    ok = bad == 0 and (sa["proposals"], sa["accepted"]) == (sb["proposals"], sb["accepted"])
    print(f"[{tag}] loci differing: {bad}/{nloci}; proposals/accepted new {sa['proposals']}/{sa['accepted']} old {sb['proposals']}/{sb['accepted']}; "
          f"lnL new {sa['total_lnl']:.6f} old {sb['total_lnl']:.6f}  {'OK' if ok else 'MISMATCH'}")
    return ok


def main():
    eng = bpp_amd.Engine(0)
    allok = True
    for taxa, nloci in ((4, 300), (8, 60)):
        data = synth.make_dataset(nloci, 400, taxa, "jc69", 1, seed=17)
        ng, nq = taxa - 1, 2 * taxa - 2
        # prefixes of the per-locus proposal list, all-loci steps off
        for g, q in [(1, 0), (ng, 0), (ng, 1), (ng, 2), (ng, nq)]:
            new, _ = make(eng, data, taxa, 23, False, (g, q), True)
            old, _ = make(eng, data, taxa, 23, True, (g, q), True)
            new.iterate(1); old.iterate(1)
            ok = compare(new, old, nloci, f"taxa {taxa} gage {g} gspr {q}")
            allok &= ok
            new.close(); old.close()
            if not ok:
                break
        # full iterations
        new, _ = make(eng, data, taxa, 23, False)
        old, _ = make(eng, data, taxa, 23, True)
        for it in range(4):
            new.iterate(1); old.iterate(1)
            ok = compare(new, old, nloci, f"taxa {taxa} full iteration {it}")
            print("   taus new", new.taus(), "\n   taus old", old.taus())
            print("   thetas new", new.thetas(), "\n   thetas old", old.thetas())
            allok &= ok
            if not ok:
                break
        new.close(); old.close()
    print("ALL OK" if allok else "FAILED")
    eng.close()


if __name__ == "__main__":
    main()
