"""throughput of the device-resident sampler on config 2 (10 000 loci, 4 taxa, JC69)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bpp_amd
from bpp_amd import synth
import tape
eng = bpp_amd.Engine(0)
data = synth.make_dataset(10000, 1000, 4, "jc69", 1, seed=12345)
loci = tape.make_engine_loci(eng, data)
smp = bpp_amd.Sampler(eng, loci, data, seed=1)
smp.initialize()
print("start lnL", smp.summary()["total_lnl"])
smp.iterate(50); eng.synchronize()
for K in (200, 1000):
    t0 = time.perf_counter(); smp.iterate(K); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{K} iterations: {dt*1e3:.2f} ms -> {K/dt:.0f} iterations/s ({dt/K*1e6:.1f} us/iteration)")
s = smp.summary()
print(s, "acceptance", s["accepted"]/s["proposals"])
