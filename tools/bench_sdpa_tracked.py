"""bench.py with the differentiable self-attention forced back to PyTorch SDPA (the round-1 tracked path) — the A side of
the A/B behind profiles/r02_selfattn_bwd_bench.txt:   python tools/bench_sdpa_tracked.py --opt-epochs 3 --steps 2 --warmup 1 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-spacetime-attn_amd"))
from sta import ops  # noqa: E402

ops.self_attention_train_supported = lambda x, heads: False
__file__ = os.path.join(ROOT, "bench.py")
sys.argv[0] = __file__
exec(compile(open(__file__).read(), __file__, "exec"))
