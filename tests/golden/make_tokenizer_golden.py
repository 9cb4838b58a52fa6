#!/usr/bin/env python
"""Golden ids for the CLIP BPE tokenizer: the reference's own vendored ``SimpleTokenizer``
(``/root/reference/src/models/components/denseclip/untils.py``) is imported and run on a set of prompts; stored are only
(text, ids) pairs.  ``ftfy`` is not installed here: it is replaced by an identity ``fix_text`` (exact for the ASCII / plain
prompts below, which is all the datasets' ``anns/*.json`` contain).  Run in the build container only."""
import json
import sys
import types
from pathlib import Path

sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda s: s))
sys.path.insert(0, "/root/reference")
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_untils", "/root/reference/src/models/components/denseclip/untils.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
tok = mod.SimpleTokenizer()
PROMPTS = [
    "a photo of a", "polyp", "one small pink round polyp located in center of the image.", "a photo of a polyp .",
    "left ventricular cavity, myocardium, or left atrium cavity of the heart in two-chamber view in the cardiac ultrasound at the end of the diastole cycle.",
    "Skin Melanoma", "bad chest  x-ray;   with\ttabs & ampersand &amp; entity", "it's the patient's 2nd scan, isn't it?", "benign-tumor (2.5cm) #42",
    "naïve café über", "日本語 text", "", "   ", "a" * 40, "the quick brown fox jumps over the lazy dog " * 6,
]
out = {"source": "reference SimpleTokenizer (denseclip/untils.py) with its bpe_simple_vocab_16e6.txt.gz; ftfy = identity",
       "vocab_size": len(tok.encoder), "bos": tok.encoder["<|startoftext|>"], "eos": tok.encoder["<|endoftext|>"],
       "cases": [{"text": p, "ids": tok.encode(p)} for p in PROMPTS]}
path = Path(__file__).resolve().parent / "tokenizer_golden.json"
path.write_text(json.dumps(out, ensure_ascii=False, indent=0))
print(f"wrote {path}: {len(PROMPTS)} prompts, vocab {out['vocab_size']}, bos {out['bos']}, eos {out['eos']}")
