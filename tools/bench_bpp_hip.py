"""The reference's OWN program on the two back-ends, same control file: oracle/_ref/bpp (CPU, AVX2) and
oracle/_ref/bpp_hip (the same objects linked against libbpp_amd.so through integration/locus_hip.c) — whole MCMC
iterations/s from the differential wall time of two run lengths.  This is north_star's architecture taken literally:
method.c's control flow unchanged, one launch and one synchronisation per locus per proposal.
usage: bench_bpp_hip.py [c2 loci] [gtr loci]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bpphip as B

def rate(binary, ctl, files, n1, n2, env=None):
    ts = []
    for ns in (n1, n2):
        t0 = time.perf_counter()
        rc, out, _ = B.run_program(binary, ctl.format(nsample=ns), files, timeout=1800, env=env)
        assert rc == 0, out[-500:]
        ts.append(time.perf_counter() - t0)
    return (n2 - n1)/max(ts[1] - ts[0], 1e-9)

G = B.GOLDEN
cases = []
n_c2 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_gtr = int(sys.argv[2]) if len(sys.argv) > 2 else 100
files = B.simulate(B.SIM_CTL.format(seed=12345, species=B.SPECIES4_SIM, phase="0 0 0 0", nloci=n_c2, sites=1000, simmodel=0, extra=""))
cases.append((f"c2-like {n_c2} loci x 1000 sites, 4 taxa, JC69", B.A00_CTL.format(species=B.SPECIES4, phase="0 0 0 0", nloci=n_c2, model="jc69", alpha="", taub=500, burnin=0, sampfreq=1, nsample="{nsample}", extra=""), files, 10, 40))
EX = "alpha_siterate = 1 0.5 4\nqrates = 1 1 2 1 0.5 1.5 1\nbasefreqs = 1 0.3 0.2 0.2 0.3\nmodelparafile = syn.para.txt\n"
files = B.simulate(B.SIM_CTL.format(seed=7, species=B.SPECIES8_SIM, phase="0 0 0 0 0 0 0 0", nloci=n_gtr, sites=1000, simmodel=7, extra=EX))
cases.append((f"c3-like {n_gtr} loci x 1000 sites, 8 taxa, GTR+G4", B.A00_CTL.format(species=B.SPECIES8, phase="0 0 0 0 0 0 0 0", nloci=n_gtr, model="gtr", alpha="alphaprior = 1 1 4", taub=300, burnin=0, sampfreq=1, nsample="{nsample}", extra=""), files, 5, 20))
cases.append(("c1 frogs (5 loci, phased diploids)", B.FROGS_CTL.format(burnin=0, sampfreq=1, nsample="{nsample}", extra=""),
              {"frogs.txt": os.path.join(G, "frogs", "frogs.txt"), "frogs.Imap.txt": os.path.join(G, "frogs", "frogs.Imap.txt")}, 100, 400))
cases.append(("c5 anopheles MSC-I (100 loci x 12 sequences)", B.ANOPHELES_CTL.format(tree=B.ANOPHELES_MSCI_TREE, phiprior="phiprior = 1 1", burnin=0, sampfreq=1, nsample="{nsample}", extra=""),
              {"loci_realign.txt": os.path.join(G, "anopheles", "loci_realign.txt"), "Imap.txt": os.path.join(G, "anopheles", "Imap.txt")}, 20, 80))
for name, ctl, files, n1, n2 in cases:
    cpu = rate(B.REF_BIN, ctl, files, n1, n2)
    hip = rate(B.HIP_BIN, ctl, files, n1, n2)
    print(f"{name}: bpp (1 thread, AVX2) {cpu:.2f} it/s | bpp_hip {hip:.2f} it/s | ratio {hip/cpu:.2f}", flush=True)
