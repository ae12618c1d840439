"""Helpers for the drop-in check (test infrastructure): run the unmodified reference program (oracle/_ref/bpp) and the
same objects linked against libbpp_amd.so through integration/locus_hip.c (oracle/_ref/bpp_hip) on one control file and
compare what they print.  Control files are written here (options of the reference's examples with `seed = 1`,
`finetune = 1` — v4.8.7 rejects the examples' own finetune lines, BASELINE.md — and short chains)."""
import os
import re
import shutil
import subprocess
import tempfile
import threading
from concurrent.futures import Future

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bpp")
HIP_BIN = os.path.join(ROOT, "oracle", "_ref", "bpp_hip")
GOLDEN = os.path.join(ROOT, "tests", "golden")

FROGS_CTL = """seed = 1
seqfile = frogs.txt
Imapfile = frogs.Imap.txt
jobname = out
speciesdelimitation = 0
speciestree = 0
species&tree = 4  K  C  L  H
                  9  7 14  2
                 (((K, C), L), H);
phase =   1  1  1  1
usedata = 1
nloci = 5
cleandata = 0
thetaprior = gamma 2 2000
tauprior = gamma 2 1000
finetune = 1
print = 1 0 0 0
burnin = {burnin}
sampfreq = {sampfreq}
nsample = {nsample}
{extra}
"""

ANOPHELES_CTL = """seed = 1
seqfile = loci_realign.txt
Imapfile = Imap.txt
jobname = out
speciesdelimitation = 0
species&tree = 6  G  C  R  L  A  Q
                  2  2  2  2  2  2
      {tree}
usedata = 1
nloci = 100
cleandata = 1
thetaprior = gamma 2 100
tauprior = gamma 2 10
{phiprior}
finetune = 1
print = 1 0 0 0
burnin = {burnin}
sampfreq = {sampfreq}
nsample = {nsample}
{extra}
"""
ANOPHELES_MSCI_TREE = ("((R, (Q)h[&phi=0.3,&tau-parent=no]) g, (f[&tau-parent=yes,&phi=0.3], "
                       "(((((G, C)b)f[&tau-parent=no], A)e, h[&tau-parent=yes])d, L)c)a)o;")
ANOPHELES_MSC_TREE = "(R, ((C, G), ((A, Q), L)));"

SIM_CTL = """seed = {seed}
seqfile = syn.txt
Imapfile = syn.Imap.txt
species&tree = {species}
phase = {phase}
loci&length = {nloci} {sites}
clock = 1
locusrate = 0
model = {simmodel}
{extra}
"""
SPECIES4_SIM = """4  A B C D
                  1 1 1 1
                  (((A #0.002, B #0.002):0.001 #0.002, C #0.002):0.002 #0.002, D #0.002):0.003 #0.002;"""
SPECIES4 = """4  A B C D
                  1 1 1 1
                  (((A, B), C), D);"""
SPECIES8_SIM = """8  A B C D E F G H
                  1 1 1 1 1 1 1 1
                  (((A #0.002, B #0.002):0.001 #0.002, (C #0.002, D #0.002):0.0015 #0.002):0.003 #0.002, ((E #0.002, F #0.002):0.001 #0.002, (G #0.002, H #0.002):0.002 #0.002):0.0035 #0.002):0.005 #0.002;"""
SPECIES8 = """8  A B C D E F G H
                  1 1 1 1 1 1 1 1
                  (((A, B), (C, D)), ((E, F), (G, H)));"""

A00_CTL = """seed = 1
seqfile = syn.txt
Imapfile = syn.Imap.txt
jobname = out
speciesdelimitation = 0
speciestree = 0
species&tree = {species}
phase = {phase}
usedata = 1
nloci = {nloci}
model = {model}
{alpha}
cleandata = 0
thetaprior = gamma 2 1000
tauprior = gamma 2 {taub}
finetune = 1
print = 1 0 0 0
burnin = {burnin}
sampfreq = {sampfreq}
nsample = {nsample}
{extra}
"""


_MEMO, _MEMO_LOCK = {}, threading.Lock()


def _memo(fn):
    """a program run is a function of its arguments (binary, control file, input files, seed inside the control file): one
    run per distinct call in a test session, whoever asks first — tests/test_gpu_bpp_hip.py starts all of its tests' runs
    side by side before the first test, the tests then read the results"""
    def key_of(x):
        if isinstance(x, dict):
            return tuple(sorted((k, key_of(v)) for k, v in x.items()))
        return x

    def wrapped(*args, **kw):
        # the early pass of tests/conftest.py (runs overlapping the rest of the GPU session) is for the CPU program only:
        # a bpp_hip process next to the persistent-kernel tests is the shared-device condition include/bpp_amd.h warns about
        if CPU_ONLY and (HIP_BIN in args or kw.get("binary") == HIP_BIN):
            raise DeferredHip()
        key = (fn.__name__, tuple(key_of(a) for a in args), key_of(kw))
        with _MEMO_LOCK:
            fut = _MEMO.get(key)
            mine = fut is None
            if mine:
                fut = _MEMO[key] = Future()
        if mine:
            try:
                fut.set_result(fn(*args, **kw))
            except BaseException as ex:       # noqa: BLE001
                fut.set_exception(ex)
        return fut.result()
    wrapped.__doc__ = fn.__doc__
    wrapped.__name__ = fn.__name__
    return wrapped


CPU_ONLY = False          # set by tests/conftest.py for its early pass


class DeferredHip(Exception):
    """a run on the GPU library asked for during the CPU-only early pass: it is made when the module's turn comes"""


def have_binaries():
    return os.path.exists(REF_BIN) and os.path.exists(HIP_BIN)


@_memo
def run_program(binary, ctl_text, files, timeout=900, env=None):
    """run `binary --cfile a.ctl` in a fresh directory holding `files` (name -> source path or text); returns
    (stdout, {name: text of the output files})"""
    d = tempfile.mkdtemp(prefix="bpphip_")
    try:
        for name, src in files.items():
            if os.path.exists(src):
                shutil.copy(src, os.path.join(d, name))
            else:
                open(os.path.join(d, name), "w").write(src)
        open(os.path.join(d, "a.ctl"), "w").write(ctl_text)
        e = dict(os.environ)
        e.update(env or {})
        r = subprocess.run([binary, "--cfile", "a.ctl"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout, env=e, text=True)
        outs = {}
        for fn in os.listdir(d):
            if fn.startswith("out."):
                outs[fn] = open(os.path.join(d, fn), errors="replace").read()
        return r.returncode, r.stdout, outs
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_both(ctl_text, files, timeout=900):
    """the unmodified program (host CPU) and bpp_hip (the same program on the library) side by side — they share nothing"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as ex:
        a = ex.submit(run_program, REF_BIN, ctl_text, files, timeout)
        b = ex.submit(run_program, HIP_BIN, ctl_text, files, timeout)
        return a.result(), b.result()


@_memo
def simulate(ctl_text, timeout=600, binary=None):
    """the reference's own simulator (`bpp --simulate`; binary = HIP_BIN: the same simulator with its P-matrices made by
    the library): returns {file name: text} of the alignment and the Imap (+ the model-parameter file if one is written)"""
    d = tempfile.mkdtemp(prefix="bppsim_")
    try:
        open(os.path.join(d, "sim.ctl"), "w").write(ctl_text)
        subprocess.run([binary or REF_BIN, "--simulate", "sim.ctl"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=timeout)
        return {fn: open(os.path.join(d, fn)).read() for fn in ("syn.txt", "syn.Imap.txt", "syn.para.txt")
                if os.path.exists(os.path.join(d, fn))}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def log_l0(stdout):
    m = re.search(r"log-L0[^=]*=\s*(-?[0-9.eE+-]+)", stdout)
    if m:
        return float(m.group(1))
    m = re.search(r"Initial MSC density and log-likelihood of observing data:\s*\n?\s*log-PG0\s*=\s*\S+\s+log-L0\s*=\s*(\S+)", stdout)
    return float(m.group(1)) if m else None


def mcmc_table(text):
    """the sample file: header row + one row per sample; returns (header list, rows of floats)"""
    lines = [ln for ln in text.splitlines() if ln.strip()]
    head = lines[0].split()
    rows = [[float(x) for x in ln.split()] for ln in lines[1:]]
    return head, rows


def compare_runs(ctl_text, files, timeout=900):
    """both programs on the same control file; returns a dict with the two log-L0, the lnL columns' largest relative
    difference, the largest relative difference over all columns, and whether mcmc.txt is byte-identical"""
    (rc0, out0, f0), (rc1, out1, f1) = run_both(ctl_text, files, timeout)
    assert rc0 == 0, out0[-2000:]
    assert rc1 == 0, out1[-2000:]
    assert "Likelihood back-end: bpp_amd" in out1
    m0, m1 = f0["out.mcmc.txt"], f1["out.mcmc.txt"]
    h0, r0 = mcmc_table(m0)
    h1, r1 = mcmc_table(m1)
    assert h0 == h1 and len(r0) == len(r1) and len(r0) > 0
    k = h0.index("lnL")
    rel = lambda a, b: abs(a - b)/max(abs(a), abs(b), 1e-300)
    lnl_err = max(rel(a[k], b[k]) for a, b in zip(r0, r1))
    all_err = max(rel(x, y) for a, b in zip(r0, r1) for x, y in zip(a, b))
    return dict(logl0_ref=log_l0(out0), logl0_hip=log_l0(out1), lnl_err=lnl_err, all_err=all_err,
                identical=(m0 == m1), samples=len(r0), stdout_hip=out1)


_NUM = re.compile(r"-?\d+\.\d+(?:[eE][-+]?\d+)?")


def compare_text_runs(ctl_text, files, timeout=900):
    """as compare_runs for sample files that are not tables (species trees of A01 / A10 / A11 with their annotations):
    the text around the decimal numbers must be the same, the numbers are compared relatively"""
    (rc0, out0, f0), (rc1, out1, f1) = run_both(ctl_text, files, timeout)
    assert rc0 == 0, out0[-2000:]
    assert rc1 == 0, out1[-2000:]
    assert "Likelihood back-end: bpp_amd" in out1
    m0, m1 = f0["out.mcmc.txt"], f1["out.mcmc.txt"]
    assert _NUM.sub("#", m0) == _NUM.sub("#", m1), "the sample files differ in more than their decimal numbers"
    n0, n1 = [float(x) for x in _NUM.findall(m0)], [float(x) for x in _NUM.findall(m1)]
    assert len(n0) == len(n1) and len(n0) > 0
    err = max(abs(a - b)/max(abs(a), abs(b), 1e-300) for a, b in zip(n0, n1))
    return dict(logl0_ref=log_l0(out0), logl0_hip=log_l0(out1), all_err=err, identical=(m0 == m1),
                samples=len([ln for ln in m0.splitlines() if ln.strip()]), stdout_hip=out1)


@_memo
def run_checkpointed(first_bin, resume_bin, ctl_text, files, timeout=900):
    """`first_bin --cfile a.ctl` with a `checkpoint = ...` line (writes out.1.chk on the way and runs to the end), then
    `resume_bin --resume out.1.chk` in the same directory (runs from the checkpoint to the end, rewriting the sample
    file from there).  Returns (sample file after the first run, sample file after the resumed run, resumed stdout)."""
    d = tempfile.mkdtemp(prefix="bppchk_")
    try:
        for name, src in files.items():
            if os.path.exists(src):
                shutil.copy(src, os.path.join(d, name))
            else:
                open(os.path.join(d, name), "w").write(src)
        open(os.path.join(d, "a.ctl"), "w").write(ctl_text)
        r = subprocess.run([first_bin, "--cfile", "a.ctl"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        assert os.path.exists(os.path.join(d, "out.1.chk")), sorted(os.listdir(d))
        m0 = open(os.path.join(d, "out.mcmc.txt")).read()
        r2 = subprocess.run([resume_bin, "--resume", "out.1.chk"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            timeout=timeout, text=True)
        assert r2.returncode == 0, r2.stdout[-2000:]
        m1 = open(os.path.join(d, "out.mcmc.txt")).read()
        return m0, m1, r2.stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)
