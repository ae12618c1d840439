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
smp.initialize(); smp.iterate(20); eng.synchronize()
t0 = time.perf_counter(); smp.iterate(200); eng.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("BPA_SMP_STEPS"), os.environ.get("BPA_SMP_NOMIX"), f"{dt/200*1e6:.1f} us")
