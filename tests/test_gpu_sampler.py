"""Device-resident proposal control (bpa_sampler_t) walks the same trajectory as the C host
driver on libbpp_amd.so with the same seed — which in turn equals the host driver on the real
reference (test_gpu_host_driver.py).  Same proposals (integer random streams), same
accept/reject history, same trees; ages and log-likelihoods to ~1e-12 (device exp/log vs glibc
in the root-age and mixing multipliers)."""
import numpy as np
import pytest

import bpp_amd
from bpp_amd import synth
import oraclelib as O
import hostdrv
import tape
from common import rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("taxa,nloci,iters", [(4, 300, 6), (8, 60, 4)])
def test_sampler_equals_host_driver(taxa, nloci, iters):
    eng = bpp_amd.Engine(0)
    data = synth.make_dataset(nloci, 400, taxa, "jc69", 1, seed=17)
    loci_a = tape.make_engine_loci(eng, data)
    loci_b = tape.make_engine_loci(eng, data)
    host = hostdrv.hip_driver(eng, loci_a, data, seed=23)
    dev = bpp_amd.Sampler(eng, loci_b, data, seed=23)
    taus = (0.001, 0.002, 0.003) if taxa == 4 else (0.0011, 0.0025, 0.005)
    host.set_taus(taus); dev.set_taus(taus)
    host.initialize(); dev.initialize()
    s = dev.summary()
    assert rel(s["total_lnl"], host.total_lnl()) < 1e-13
    for it in range(iters):
        host.iterate(); dev.iterate(1)
        s = dev.summary()
        hp, ha, _ = host.counters()
        assert (s["proposals"], s["accepted"]) == (hp, ha), it
        assert rel(s["total_lnl"], host.total_lnl()) < 1e-11, it
    assert np.allclose(dev.taus(), host.taus(), rtol=1e-12, atol=0) and dev.taus() != list(taus)
    for i in range(nloci):
        a, b = dev.tree(i), host.tree(i)
        assert a["root"] == b["root"]
        for key in ("left", "right", "parent", "clv", "pmat"):
            assert [int(x) for x in a[key]] == [int(x) for x in b[key]], (i, key)
        assert np.allclose(a["time"], b["time"], rtol=1e-12, atol=0)
        assert rel(a["lnl"], b["lnl"]) < 1e-11
    # the device state is the state of the explicit-index API: buffers hold what the indices say
    for i in range(0, nloci, max(1, nloci // 10)):
        t = dev.tree(i)
        d = data[i]
        have = loci_b[i].root_loglikelihood(int(t["clv"][t["root"]]), -1)
        full = O.OracleLocus(4, 1, d["seqs"], d["weights"]).full_lnl(list(t["left"]), list(t["right"]), list(t["time"]), t["root"])
        assert rel(have, full) < 1e-12 and rel(t["lnl"], full) < 1e-12
    # 4 launches per iteration instead of 3 tips - 2 host round trips
    assert dev.summary()["launches"] <= 1 + (2 + len(taus)) * iters + 2 * (iters + 2) + nloci
    host.close(); dev.close(); eng.close()


def test_sampler_rejects_unsupported_loci(engine):
    data = synth.make_dataset(2, 200, 8, "gtr", 4, seed=1)
    loci = tape.make_engine_loci(engine, data)
    with pytest.raises(bpp_amd.BpaError, match="JC69"):
        bpp_amd.Sampler(engine, loci, data)
