"""Prompt sources of the three entry points and the layout input that replaces the layout predictor.

Parsing rules restated from the reference scripts:
  gpt     500 records of 4 lines ("Objects:", "Relation:", "Sentence:", blank); prompt = rows[4*i+2][10:]
          (scripts/txt2img-gpt.py:255-261)
  mscoco / vsr   one prompt per line, first 500 (scripts/txt2img-mscoco.py:255-261, txt2img-vsr.py:255-261)
The layout predictor (LP/inference/inference_coco.py:486-544, `inference_sentence(prompt) -> {name: [x, y]}`
or None) is out of scope: its OUTPUT FORMAT is the input here — a JSON file mapping each prompt (or its
index as a string) to `{noun_chunk: [x, y]}` with x, y in [0, 1].
"""
import json


def parse_prompts(text, kind, limit=500):
    rows = text.split("\n")
    if kind == "gpt":
        rows = rows[: 4 * limit]
        n = min(limit, (len(rows) + 1) // 4)
        return [rows[4 * i + 2][10:] for i in range(n) if 4 * i + 2 < len(rows)]
    if kind in ("mscoco", "vsr"):
        return [r for r in rows[:limit]]
    raise ValueError("unknown dataset kind %r" % kind)


def load_prompts(path, kind, limit=500):
    with open(path, "r") as f:
        return parse_prompts(f.read(), kind, limit)


def load_layouts(path):
    """{prompt or str(index): {object name: [x, y]}}"""
    with open(path, "r") as f:
        return json.load(f)


def layout_for(layouts, prompt, index):
    """The reference's `result = inference_sentence(prompt)`: dict name -> [x, y], or None when the
    predictor found no object (the reference scripts then crash, txt2img-gpt.py:317; here the prompt is
    sampled as plain SD — plms.py:207-208 already creates an empty weight tensor for K = 0)."""
    if layouts is None:
        return None
    return layouts.get(prompt, layouts.get(str(index)))
