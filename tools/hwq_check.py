"""Does the library's own GPU_MAX_HW_QUEUES default (engine.hip: hw_queues_default) take effect when nothing else sets the variable?
The default bench run without the CPU legs, the variable removed from the environment after the imports (the HIP runtime has not
started yet): config 3's sampler inside it reads ~235 it/s with eight queues, ~188 with HIP's four.
    python tools/hwq_check.py [keep]        (keep: leave the package's setting in place; HWQ_FORCE=4: the user's value, which is kept)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
sys.argv = ["bench.py", "--no-bpp-program", "--no-host-control", "--no-scale-projection", "--no-cpu-baseline", "--no-efficiency",
            "--full-record", os.path.join(ROOT, "gpurun_out", "hwq_%s.json" % ("keep" if keep else os.environ.get("HWQ_FORCE") or "lib"))]
import bpp_amd   # noqa: E402,F401
import bench     # noqa: E402
if not keep:
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
if os.environ.get("HWQ_FORCE"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["HWQ_FORCE"]
print("GPU_MAX_HW_QUEUES before the first HIP call:", os.environ.get("GPU_MAX_HW_QUEUES"), file=sys.stderr)
bench.main()
