"""Entry point for the gpt prompt set (reference: scripts/txt2img-gpt.py). See _txt2img_common.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _txt2img_common import run  # noqa: E402

if __name__ == "__main__":
    run("gpt", "../../datasets/gpt.txt")
