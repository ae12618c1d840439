import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (the .so files are git-ignored): build in-tree, as __graft_entry__.build()
    # does (hipcc cross-compiles for gfx950 without a GPU); the oracle and the in-place reference build too when possible
    from bpp_amd import build as _build
    if not os.path.exists(_build.OUT) or not os.path.exists(_build.HOST_OUT):
        import shutil
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            try:
                import __graft_entry__
                __graft_entry__.build()
            except Exception as exc:       # the tests that need the library will say so
                print(f"[conftest] build failed: {exc}", file=sys.stderr)


@pytest.fixture(scope="session")
def engine():
    import bpp_amd
    e = bpp_amd.Engine(0)
    yield e
    e.close()


# the 10 000-locus data sets of the full-size tests take 3-8 s each to make (numpy, one core): started in background
# processes when the session's selection holds those tests, read back from BPP_AMD_SYNTH_CACHE when the tests get there
_BIG_SETS = {"test_gpu_fullsize.py::test_full_size_properties[C2": (10000, 1000, 4, "jc69", 1, 777, 1.0),
             "test_gpu_fullsize.py::test_full_size_properties[C3": (10000, 1000, 8, "gtr", 4, 777, 1.0),
             "test_gpu_fullsize.py::test_full_size_properties[C4": (2000, 500, 6, "lg", 4, 777, 3.0),
             "test_gpu_tape.py::test_chain_launch_equals_step_by_step[4-False-10000]": (10000, 500, 4, "jc69", 1, 33, 1.0)}
_bg = []
_program_runs = {}          # the whole-program runs of test_gpu_bpp_hip.py, started at collection (see its fixture)


def pytest_collection_modifyitems(session, config, items):
    # the whole-program module goes last: its ~30 runs of `bpp` / `bpp_hip` (subprocesses, started below when the
    # collection is done) then overlap the rest of the session instead of being 30 s of waiting of their own
    last = [it for it in items if it.nodeid.split("::")[0].endswith("test_gpu_bpp_hip.py")]
    if last and len(last) < len(items):
        items[:] = [it for it in items if it not in last] + last


def _start_program_runs(session, cpu_only=True):
    """cpu_only (the pass started at collection, overlapping the rest of the session): only the runs of the unmodified CPU
    program are made; the bpp_hip runs — GPU processes, which must not share the device with the persistent-kernel tests
    (a launch that waits for a co-tenant's workgroups gives up and is run again: correct, but it moves the rate tests) —
    wait for the second pass, which tests/test_gpu_bpp_hip.py starts when its turn comes (last)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import bpphip
    items = [it for it in session.items if it.nodeid.split("::")[0].endswith("test_gpu_bpp_hip.py")]
    if not items or any(it.get_closest_marker("skipif") and it.get_closest_marker("skipif").args[0] for it in items):
        return
    if cpu_only and _program_runs:
        return

    def dry(it):
        try:
            it.obj(**(it.callspec.params if hasattr(it, "callspec") else {}))
        except (AssertionError, bpphip.DeferredHip):
            pass                   # (the test proper reports a failed assertion; a deferred run is the second pass's)
        except BaseException as ex:      # noqa: BLE001  (the test proper will meet it again: say so now)
            print(f"[conftest] dry run of {it.nodeid}: {type(ex).__name__}: {ex}", file=sys.stderr)

    def run():
        bpphip.CPU_ONLY = cpu_only
        try:
            # (the second pass is GPU processes only: more than four of them at once time-slice the device so badly that a
            #  run of seconds takes minutes — the first pass had CPU runs between them)
            with ThreadPoolExecutor(max_workers=(8 if len(session.items) > len(items) else 12) if cpu_only else 4) as ex:
                list(ex.map(dry, items))
        finally:
            bpphip.CPU_ONLY = False
    t = threading.Thread(target=run, daemon=True)
    _program_runs["thread"] = t
    t.start()


def pytest_collection_finish(session):
    import subprocess
    import tempfile
    if session.config.getoption("collectonly", False):
        return
    _start_program_runs(session)
    want = [a for k, a in _BIG_SETS.items() if any(k in it.nodeid for it in session.items)]
    if not want or len(session.items) < 8:
        return
    d = tempfile.mkdtemp(prefix="bpp_amd_synth_")
    os.environ["BPP_AMD_SYNTH_CACHE"] = d
    for a in want:
        _bg.append(subprocess.Popen([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); from bpp_amd import synth; synth.precompute(*{a!r})"],
                                    env=dict(os.environ), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))


def pytest_sessionfinish(session, exitstatus):
    import shutil
    for p in _bg:
        if p.poll() is None:
            p.kill()
        p.wait()          # (our own children, by handle)
    d = os.environ.pop("BPP_AMD_SYNTH_CACHE", None)
    if d and _bg:
        shutil.rmtree(d, ignore_errors=True)
