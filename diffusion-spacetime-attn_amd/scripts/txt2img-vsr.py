"""Entry point for the vsr prompt set (reference: scripts/txt2img-vsr.py). See _txt2img_common.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _txt2img_common import run  # noqa: E402

if __name__ == "__main__":
    run("vsr", "../../datasets/vsr.txt")
