"""sta — host glue between PyTorch tensors and the gfx950 fused cross-attention library."""
from . import lib  # noqa: F401
