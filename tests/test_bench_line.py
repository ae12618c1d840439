"""The line bench.py prints must stay readable by the driver: round 4's 26 KB line was cut by its capture and the round had no
driver-recorded figure.  The compact line is built from the full record by bench.compact_line; here: the committed full records
of earlier rounds -> a line of <= 8 192 bytes that parses and carries the contract's keys, `roofline` and `cpu_baseline`."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_full.json")))


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.relpath(p, ROOT) for p in RECORDS])
def test_compact_line_of_a_committed_record(path):
    b = _bench()
    full = json.load(open(path))
    line = json.dumps(b.compact_line(full), separators=(",", ":"))
    assert len(line) <= 8192, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k


def test_an_oversized_record_still_gives_a_short_line():
    """whatever a section grows to, the line drops optional parts rather than outgrow the capture"""
    b = _bench()
    # (the newest FULL record: a committed bench_default.json of round 5 on is already the compact line)
    full = [d for d in (json.load(open(p)) for p in RECORDS) if "other_configs" in d][-1]
    full["other_configs"] = {f"x{i}": dict(full["other_configs"]["c3"]) for i in range(60)}
    line = json.dumps(b.compact_line(full), separators=(",", ":"))
    assert len(line) <= 8192
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and d["value"] == full["value"]


def test_bench_prints_the_compact_line_only():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("print(json.dumps(") == 1 and "print(json.dumps(compact_line(out)" in src


# ---- round 6 (task 5): the line's figures share their denominators
def _newest_round_dir():
    import re
    ds = [d for d in glob.glob(os.path.join(ROOT, "profiles", "r*")) if re.fullmatch(r"r\d+", os.path.basename(d))]
    d = max(ds, key=lambda x: int(os.path.basename(x)[1:]))
    return d, int(os.path.basename(d)[1:])


def test_the_committed_line_is_self_consistent():
    """(i) `frac_pmc` of config 3 / 4 divides the bytes of FULL-batch launches (tools/profile_cfg.sh buckets a kernel's dispatches by
    grid size) by the time of full-batch launches — the same file's own quotient (`moved_same_file`) agrees with the line's;
    (ii) `site_lnl_updates_per_s` comes from the run `value` comes from (the sampler's device counters), the tape's stays under
    `likelihood_only`; (iii) `metric` is BASELINE.json's string, the qualifier sits in `config.moves`"""
    d_, n = _newest_round_dir()
    path = os.path.join(d_, "bench_default.json")
    if n < 6 or not os.path.exists(path):
        pytest.skip("no round-6 line committed yet")
    line = json.load(open(path))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"]
    assert line["config"]["moves"] == "program" and len(line["config"]["workload"]) <= 120
    lo = line["likelihood_only"]
    per_it, per_it_tape = line["site_lnl_updates_per_s"] / line["value"], lo["site_lnl_updates_per_s"] / lo["it_s"]
    assert 0.6 < per_it / per_it_tape < 1.6, (per_it, per_it_tape)          # the same proposals per iteration, drawn differently
    assert abs(line["site_lnl_updates_per_s"] - lo["site_lnl_updates_per_s"]) > 1e-6 * lo["site_lnl_updates_per_s"]
    for key in ("c3", "c4"):
        prof = json.load(open(os.path.join(d_, f"profile_{key}.json")))
        assert prof.get("pmc_buckets", "").startswith("largest grid") and prof["full_batch"] and prof["moved_same_file"]
        c = line["configs"][key]
        kern = c["kernel"]
        same = [v for k, v in prof["moved_same_file"].items() if kern in k]
        assert same, (kern, list(prof["moved_same_file"]))
        assert abs(c["frac_pmc"] / same[0]["frac_of_8TBps"] - 1) < 0.02, (key, c["frac_pmc"], same[0])     # (bytes AND time are the profile's)
