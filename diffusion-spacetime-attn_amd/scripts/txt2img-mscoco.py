"""Entry point for the mscoco prompt set (reference: scripts/txt2img-mscoco.py). See _txt2img_common.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _txt2img_common import run  # noqa: E402

if __name__ == "__main__":
    run("mscoco", "../../datasets/mscoco.txt")
