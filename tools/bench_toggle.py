"""Same-box A/B of a sta.fused switch on the whole bench: python tools/bench_toggle.py LINEAR_ROWS=0 CONV3X3=1 -- [bench.py arguments]
(boxes of the pool differ by more than most single changes move the headline: only runs of one gpurun call compare)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
from sta import fused  # noqa: E402

for kv in sys.argv[1:sep]:
    k, v = kv.split("=")
    mod = fused
    if "." in k:                      # e.g. pipeline.VAE_NHWC_TRACKED=0
        import importlib
        mname, k = k.split(".")
        mod = importlib.import_module("sta." + mname)
    assert hasattr(mod, k), k
    setattr(mod, k, type(getattr(mod, k))(int(v)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
