"""Row f2 (input side): CLIP BPE tokenizer, pad-to-longest collation.  CPU only.

The tokenizer is pinned by ``tests/golden/tokenizer_golden.json`` -- ids produced by the reference's vendored ``SimpleTokenizer``
(``tests/golden/make_tokenizer_golden.py``).  That class lists the two specials right after the 512 byte symbols (ids 512 / 513),
whereas the tokenizer on the reference's hot path (HF ``CLIPTokenizer``: BOS 49406, EOS 49407, ``TextConfig`` of the backbone) keeps
OpenAI's order with the specials last; the BPE segmentation is the same, so golden id ``g`` maps to ``g`` (< 512), to the specials
(512 / 513) or to ``g - 2`` (merges).  The merge table itself is not shipped: the golden comparison runs where the file exists."""
import gzip
import json
from pathlib import Path

import pytest
import torch

from tunevlseg_amd.data import ClipBpeTokenizer, PadToLongestCollator
from tunevlseg_amd.data.tokenizer import N_MERGES, _byte_symbols

REF_BPE = Path("/root/reference/src/models/components/denseclip/bpe_simple_vocab_16e6.txt.gz")
GOLDEN = json.loads((Path(__file__).resolve().parent / "golden" / "tokenizer_golden.json").read_text())


def test_byte_symbol_table():
    s = _byte_symbols()
    assert len(set(s)) == 256 and s[ord("a")] == "a" and s[ord("!")] == "!" and s[0xFF] == "ÿ"
    assert s[ord(" ")] == chr(0x100 + 32) and s[0] == chr(0x100)  # bytes 0..32 are the first re-mapped ones
    assert not any(ch.isspace() for ch in s)


@pytest.fixture(scope="module")
def toy_tokenizer(tmp_path_factory):
    """A merge table with a handful of real rules and filler to the expected length: the algorithm, without the real vocabulary."""
    rules = [("l", "o"), ("lo", "w</w>"), ("e", "r</w>"), ("n", "e"), ("ne", "w"), ("new", "er</w>"), ("w", "i"), ("d", "e")]
    filler = [(f"¤{i}", f"¤{i}") for i in range(N_MERGES - len(rules))]
    path = tmp_path_factory.mktemp("bpe") / "merges.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as fh:
        fh.write("#version: toy\n" + "\n".join(" ".join(r) for r in (*rules, *filler)) + "\n")
    return ClipBpeTokenizer(path)


def test_bpe_merges_lowest_rank_first(toy_tokenizer):
    t = toy_tokenizer
    assert len(t) == 49408 and (t.bos_token_id, t.eos_token_id) == (49406, 49407)
    pieces = [t.decoder[i] for i in t.encode("Low  newer")]  # lower-cased, whitespace collapsed
    assert pieces == ["low</w>", "newer</w>"]
    # 'e r</w>' (rank 2) outranks 'd e' (rank 7): "er</w>" forms first and 'd' is left without a partner
    assert [t.decoder[i] for i in t.encode("wider")] == ["wi", "d", "er</w>"]
    out = t("low newer")
    assert out["input_ids"][0] == 49406 and out["input_ids"][-1] == 49407 and out.input_ids == out["input_ids"]
    assert out["attention_mask"] == [1] * len(out["input_ids"])
    assert t.decode(t.encode("low newer")) == "low newer "


@pytest.mark.skipif(not REF_BPE.exists(), reason="the CLIP merge table is not shipped with this repo (present in the build container)")
def test_tokenizer_matches_reference_segmentation():
    t = ClipBpeTokenizer(REF_BPE)
    assert len(t) == GOLDEN["vocab_size"] == 49408
    remap = lambda g: g if g < 512 else (t.bos_token_id if g == GOLDEN["bos"] else t.eos_token_id if g == GOLDEN["eos"] else g - 2)  # noqa: E731
    for case in GOLDEN["cases"]:
        assert t.encode(case["text"]) == [remap(g) for g in case["ids"]], case["text"]
    # the ids the context_initializer of the configs relies on ("a photo of a" -> 320 1125 539 320)
    assert t.encode("a photo of a") == [320, 1125, 539, 320]
    row = t("a photo of a")
    assert row["input_ids"] == [49406, 320, 1125, 539, 320, 49407]
    long = t("the quick brown fox " * 40, truncation=True)
    assert len(long["input_ids"]) == 77 and long["input_ids"][-1] == 49407


def test_pad_to_longest_collator():
    feats = [{"input_ids": [49406, 5, 49407], "attention_mask": [1, 1, 1], "image": torch.zeros(3, 4, 4), "mask_name": "a.png",
              "mask_shape": torch.tensor([7, 9])},
             {"input_ids": [49406, 5, 6, 7, 49407], "attention_mask": [1] * 5, "image": torch.ones(3, 4, 4), "mask_name": "b.png",
              "mask_shape": torch.tensor([4, 4])}]
    out = PadToLongestCollator(("input_ids", "attention_mask"), pad_token_id=1)(feats)
    assert out["input_ids"].tolist() == [[49406, 5, 49407, 1, 1], [49406, 5, 6, 7, 49407]]
    assert out["attention_mask"].tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]
    assert out["image"].shape == (2, 3, 4, 4) and out["mask_name"] == ["a.png", "b.png"] and out["mask_shape"].tolist() == [[7, 9], [4, 4]]
    assert PadToLongestCollator(pad_token_id=0, pad_to_multiple_of=8)(feats)["input_ids"].shape == (2, 8)
    with pytest.raises(ValueError):
        PadToLongestCollator(())


def _write_toy_dataset(root: Path):
    from PIL import Image
    import numpy as np

    (root / "images").mkdir()
    (root / "masks").mkdir()
    rng = np.random.default_rng(0)
    tasks = []
    for i, (h, w) in enumerate([(20, 30), (17, 17), (40, 24)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / "images" / f"im{i}.png")
        m = (rng.random((h, w)) > 0.6).astype(np.uint8) * 255
        Image.fromarray(m).save(root / "masks" / f"m{i}.png")
        tasks.append({"img_name": f"im{i}.png", "mask_name": f"m{i}.png",
                      "prompts": {"p0": "", "p1": f"polyp number {i}", "p2": [f"a round polyp {i}", f"a pink polyp {i}"]}})
    (root / "anns.json").write_text(json.dumps(tasks))
    return tasks


def test_dataset_items_follow_the_reference_wire_format(tmp_path, toy_tokenizer):
    """image_dir / mask_dir / task file with prompts p0..pN -> item keys, original mask_shape, prompt selection rules
    (reference image_text_mask_dataset.py:46-128), uint8 payloads, pad-to-longest collation across ragged prompts."""
    import random

    import numpy as np

    from tunevlseg_amd.data import ImageTextMaskDataset, ResizeTransform

    _write_toy_dataset(tmp_path)
    kw = dict(image_dir=tmp_path / "images", mask_dir=tmp_path / "masks", task_path=tmp_path / "anns.json", tokenizer=toy_tokenizer,
              transforms=ResizeTransform(16))
    ds = ImageTextMaskDataset(prompt_index=1, **kw)
    assert len(ds) == 3
    it = ds[2]
    assert set(it) == {"image", "mask", "mask_shape", "mask_name", "prompt", "input_ids", "attention_mask"}
    assert it["image"].shape == (16, 16, 3) and it["image"].dtype == torch.uint8
    assert it["mask"].shape == (16, 16) and it["mask"].dtype == torch.uint8 and set(it["mask"].unique().tolist()) <= {0, 255}
    assert it["mask_shape"].tolist() == [40, 24] and it["mask_name"] == "m2.png" and it["prompt"] == "polyp number 2"
    assert it["input_ids"][0] == toy_tokenizer.bos_token_id and it["input_ids"][-1] == toy_tokenizer.eos_token_id
    # list-valued prompt -> one of its elements; negative index -> any key but p0; override; trailing stop
    random.seed(0)
    assert ImageTextMaskDataset(prompt_index=2, **kw)[0]["prompt"] in {"a round polyp 0", "a pink polyp 0"}
    for _ in range(8):
        assert ImageTextMaskDataset(prompt_index=-1, **kw)[1]["prompt"] != ""
    assert ImageTextMaskDataset(prompt_index=1, override_prompt="tumour", insert_stop_at_last=True, **kw)[0]["prompt"] == "tumour."
    # without a transform the sample keeps its own geometry
    raw = ImageTextMaskDataset(prompt_index=1, **{**kw, "transforms": None})[0]
    assert raw["image"].shape == (20, 30, 3) and np.array_equal(raw["mask_shape"], [20, 30])
    with pytest.raises(ValueError, match="Image not found"):
        ImageTextMaskDataset(prompt_index=1, **{**kw, "image_dir": tmp_path / "nowhere"})[0]
    batch = PadToLongestCollator(pad_token_id=toy_tokenizer.eos_token_id)([ds[i] for i in range(3)])
    assert batch["image"].shape == (3, 16, 16, 3) and batch["mask"].shape == (3, 16, 16)
    assert batch["input_ids"].shape == batch["attention_mask"].shape and batch["mask_name"] == ["m0.png", "m1.png", "m2.png"]


@pytest.mark.gpu
def test_device_batch_prep_equals_the_host_transform(tmp_path, toy_tokenizer):
    """uint8 batch -> normalised fp32 image / mask on the device == A.Normalize + ToTensorV2 + mask / 255 done on the host."""
    from tunevlseg_amd.data import DeviceBatchPrep, ImageTextMaskDataset, ResizeTransform

    _write_toy_dataset(tmp_path)
    ds = ImageTextMaskDataset(image_dir=tmp_path / "images", mask_dir=tmp_path / "masks", task_path=tmp_path / "anns.json",
                              prompt_index=1, tokenizer=toy_tokenizer, transforms=ResizeTransform(32))
    batch = PadToLongestCollator(pad_token_id=toy_tokenizer.eos_token_id)([ds[i] for i in range(3)])
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = DeviceBatchPrep(mean, std)(batch)
    ref_img = ((batch["image"].float() / 255 - torch.tensor(mean)) / torch.tensor(std)).permute(0, 3, 1, 2)
    assert out["image"].shape == (3, 3, 32, 32) and (out["image"].cpu() - ref_img).abs().max().item() <= 1e-6
    assert torch.equal(out["mask"].cpu(), (batch["mask"].float() / 255)[:, None])
    assert out["input_ids"].is_cuda and out["mask_name"] == batch["mask_name"]
