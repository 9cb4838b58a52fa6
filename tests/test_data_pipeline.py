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


# ---- device-side transforms (reference configs/experiment/coop/clipseg.yaml:78-120) -------------------------------------------------
def test_oracle_cubic_resize_known_answers():
    """Hand-derived vectors of OpenCV's 8-bit INTER_CUBIC definition (A = -0.75, half-pixel centres, 11-bit weights, replicate border).
    Doubling the width puts every output at fraction 0.25 / 0.75: weights (-216, 1800, 536, -72) / 2048 and their mirror.  Row
    [0, 100, 200, 255] -> e.g. x = 3: taps (0, 100, 200, 255) . (-216, 1800, 536, -72) = 268840 -> 131.27 -> 131.  "Unpinned": derived from
    the published algorithm, cv2 itself is absent."""
    import numpy as np

    from oracle import augment_oracle as A

    w = A.cubic_weights(np.array([0.25, 0.75, 0.0], np.float32))
    assert np.array_equal(np.rint(w.astype(np.float64) * 2048), [[-216, 1800, 536, -72], [-72, 536, 1800, -216], [0, 2048, 0, 0]])
    row = np.array([[0, 100, 200, 255]], np.uint8)
    assert A.resize_cubic_u8(row, 1, 8).tolist() == [[0, 19, 67, 131, 175, 223, 246, 255]]
    img = np.random.default_rng(0).integers(0, 256, (9, 7, 3), dtype=np.uint8)
    assert np.array_equal(A.resize_cubic_u8(img, 9, 7), img)                       # same size: every fraction is 0
    assert np.array_equal(A.resize_cubic_u8(np.full((5, 6, 3), 77, np.uint8), 11, 13), np.full((11, 13, 3), 77, np.uint8))   # weights sum to 2048
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert A.resize_nearest_u8(m, 6, 8).tolist() == np.repeat(np.repeat(m, 2, 0), 2, 1).tolist()   # floor(d * 0.5): each pixel twice
    assert A.resize_nearest_u8(m, 2, 3).tolist() == [[0, 1, 2], [4, 5, 6]]                          # floor(d * 1.5 / d * 1.33)
    # brightness / contrast table: clip(v * 1.1 + 0.05 * 255) truncated
    lut = A.brightness_contrast_u8(np.arange(256, dtype=np.uint8), 1.1, 0.05)
    assert lut[0] == 12 and lut[100] == 122 and lut[255] == 255
    # identity affine parameters give the identity matrix; a pure translation moves the content with replicate padding
    assert np.allclose(A.affine_matrix(10, 20, 1, 1, 0, 0, 0), [[1, 0, 0], [0, 1, 0]])
    M = A.affine_matrix(4, 4, 1, 1, 0, 0.25, 0)   # one pixel to the right
    g = np.arange(16, dtype=np.uint8).reshape(4, 4) * 10
    assert A.warp_affine_nearest_u8(g, A.invert_affine(M)).tolist() == [[0, 0, 10, 20], [40, 40, 50, 60], [80, 80, 90, 100], [120, 120, 130, 140]]
    assert A.warp_affine_cubic_u8(g, A.invert_affine(M)).tolist() == A.warp_affine_nearest_u8(g, A.invert_affine(M)).tolist()   # integer shift: exact


def test_shard_sampler_partitions_like_a_distributed_sampler():
    from tunevlseg_amd.data import ShardSampler

    n = 11
    for shuffle in (False, True):
        shards = []
        for r in range(2):
            s = ShardSampler(n, 2, r, shuffle, seed=5)
            s.set_epoch(3)
            shards.append(list(s))
        assert len(shards[0]) == len(shards[1]) == 6                       # padded by wrap-around: same number of steps on every rank
        assert set(shards[0]) | set(shards[1]) == set(range(n))            # together they cover the dataset
        assert len(set(shards[0]) & set(shards[1])) <= 1                   # the only overlap is the one padding sample
    a, b = ShardSampler(n, 2, 0, True, seed=5), ShardSampler(n, 2, 0, True, seed=5)
    a.set_epoch(0), b.set_epoch(1)
    assert list(a) != list(b) and sorted(list(ShardSampler(n, 1, 0, True, seed=1))) == list(range(n))
    assert list(ShardSampler(4, 2, 1, False)) == [1, 3] and len(ShardSampler(11, 2, 0, True, drop_last=True)) == 5
    with pytest.raises(ValueError):
        ShardSampler(4, 2, 2, False)


def test_reference_data_config_instantiates_the_device_pipeline(tmp_path, toy_tokenizer, monkeypatch):
    """``experiment=coop/clipseg`` of the reference's OWN config tree: cfg.data -> ImageTextDatamodule over ImageTextMaskDataset with the
    albumentations list as a device plan (Resize INTER_CUBIC -> Affine p 0.2 -> RandomBrightnessContrast p 0.2 -> Normalize)."""
    ref = Path("/root/reference/configs")
    if not ref.exists():
        pytest.skip("the reference's config tree is not on this machine")
    from tunevlseg_amd import config_loader as CL
    from tunevlseg_amd.data import Compose, ImageTextDatamodule, ImageTextMaskDataset

    root = tmp_path / "kvasir"
    root.mkdir()
    _write_toy_dataset(root)
    (root / "anns").mkdir()
    for split in ("train", "val", "test"):
        (root / "anns" / f"{split}.json").write_text((root / "anns.json").read_text())
    monkeypatch.setenv("TVL_CLIP_BPE", str(toy_tokenizer.bpe_path))
    cfg = CL.Composer(ref).compose("train", ["experiment=coop/clipseg", f"data_root={tmp_path}", "ds_name=kvasir", "prompt_index=1",
                                             "data.batch_size=2", "data.num_workers=0"])
    dm = CL.instantiate(CL.select(cfg, "data"))
    assert isinstance(dm, ImageTextDatamodule) and isinstance(dm.train_ds, ImageTextMaskDataset) and len(dm.train_ds) == 3
    plan = dm.train_ds.transforms.plan()
    assert isinstance(dm.train_ds.transforms, Compose) and plan["size"] == (352, 352)
    assert plan["affine"].p == 0.2 and plan["affine"].rotate == (-5.0, 5.0) and plan["affine"].scale == (0.98, 1.02) and plan["bc"].p == 0.2
    assert plan["normalize"].mean == (0.485, 0.456, 0.406)
    ev = dm.val_ds.transforms.plan()
    assert ev["affine"] is None and ev["bc"] is None and ev["size"] == (352, 352)
    item = dm.train_ds[1]   # decoded, untouched: the transforms run on the device per batch
    assert item["image"].shape == (17, 17, 3) and item["prompt"] == "polyp number 1."
    dm.setup("fit", world_size=2, rank=1, device="cpu")
    assert dm.batch_size_per_device == 1
    with pytest.raises(ValueError, match="not divisible"):
        ImageTextDatamodule(train_ds=dm.train_ds, batch_size=3).setup("fit", world_size=2, rank=0)


@pytest.mark.gpu
def test_device_resize_and_augment_equal_the_cpu_restatement(tmp_path, toy_tokenizer):
    """tvl_resize_u8 (cubic / nearest over a ragged batch) and tvl_augment_u8 (affine warp + brightness / contrast + normalise) against
    oracle/augment_oracle.py on the same inputs and the same random draws: the integer paths bit for bit, the float warp to one grey level."""
    import numpy as np

    from oracle import augment_oracle as A
    from tunevlseg_amd import hip
    from tunevlseg_amd.data import ImageTextMaskDataset, PadToLongestCollator, RaggedCollator
    from tunevlseg_amd.data import transforms as T

    _write_toy_dataset(tmp_path)
    S = 48
    comp = T.Compose([T.Resize(S, S, interpolation=2), T.Affine(scale=[0.9, 1.1], translate_percent=[-0.05, 0.05], rotate=[-15, 15], interpolation=2, mode=1, p=0.7),
                      T.PadIfNeeded(S, S, border_mode=1), T.CropNonEmptyMaskIfExists(S, S), T.RandomBrightnessContrast(0.2, 0.2, p=0.7),
                      T.Normalize(), T.ToTensorV2(transpose_mask=True)])
    ds = ImageTextMaskDataset(image_dir=tmp_path / "images", mask_dir=tmp_path / "masks", task_path=tmp_path / "anns.json", prompt_index=1,
                              tokenizer=toy_tokenizer, transforms=comp)
    batch = RaggedCollator(PadToLongestCollator(pad_token_id=toy_tokenizer.eos_token_id))([ds[i] for i in range(3)])
    assert batch["image_hw"].tolist() == [[20, 30], [17, 17], [40, 24]] and batch["image_bytes"].numel() == 3 * (600 + 289 + 960)
    dt = T.DeviceTransform(comp, "cuda", seed=3)
    dt.set_epoch(0, 3, 0)
    out = dt(batch)
    # the same draws on the host
    rng = np.random.default_rng([3, 0, 0])
    nz = comp.plan()["normalize"]
    for b in range(3):
        item = ds[b]
        img = A.resize_cubic_u8(item["image"].numpy(), S, S)
        msk = A.resize_nearest_u8(item["mask"].numpy(), S, S)
        if rng.random() < 0.7:
            Minv = comp.plan()["affine"].sample(rng, S, S)
            img_w, msk = A.warp_affine_cubic_u8(img, Minv), A.warp_affine_nearest_u8(msk, Minv)
        else:
            img_w = img
        warped = img_w is not img
        if rng.random() < 0.7:
            al, be = comp.plan()["bc"].sample(rng)
            img_w = A.brightness_contrast_u8(img_w, np.float32(al), np.float32(be))
        ref = torch.from_numpy(A.normalize_chw(img_w, nz.mean, nz.std))
        got = out["image"][b].cpu()
        step = (1 / 255) / min(nz.std)   # one grey level after normalisation
        diff = (got - ref).abs()
        if warped:
            assert diff.max().item() <= 1.3 * step and (diff > 1e-6).float().mean().item() < 0.02, (b, diff.max().item())
        else:
            assert diff.max().item() <= 1e-6, (b, diff.max().item())
        mm = (out["mask"][b, 0].cpu() - torch.from_numpy(msk.astype(np.float32) / 255)).abs()
        assert (mm > 0).float().mean().item() <= (0.01 if warped else 0.0)
    # the resize alone, every sample, bit for bit
    hw = batch["image_hw"]
    px = hw[:, 0].long() * hw[:, 1].long()
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), px.cumsum(0)[:-1]])
    r = hip.resize_u8(batch["image_bytes"].cuda(), (3 * offs).cuda(), hw.cuda(), 3, S, S, hip.INTER_CUBIC).cpu().numpy()
    for b in range(3):
        assert np.array_equal(r[b], A.resize_cubic_u8(ds[b]["image"].numpy(), S, S))


@pytest.mark.gpu
def test_token_ids_outside_the_embedding_table_give_nan_rows_not_a_wild_read():
    """A tokenizer that does not belong to the backbone (CLIP ids up to 49407 against a 64-row table) must not fault the GPU."""
    from tunevlseg_amd import hip

    hip.load()
    table, pos = torch.randn(64, 32, device="cuda"), torch.zeros(8, 32, device="cuda")
    ids = torch.tensor([[62, 5, 49407, 63]], device="cuda")
    out = hip.text_assemble(ids, hip.const_i32([0, 1, 2, 3], "cuda"), table, None, 0, pos, 1, 4, 32)
    assert torch.equal(out[0, 0], table[62]) and torch.equal(out[0, 1], table[5]) and torch.equal(out[0, 3], table[63])
    assert torch.isnan(out[0, 2]).all()


@pytest.mark.skipif(not REF_BPE.exists(), reason="the CLIP merge table is not shipped with the repo (build container only)")
def test_denseclip_class_names_tokenise_to_the_references_own_ids():
    """DenseCLIP's ``self.texts`` (denseclip.py:99-101): [SOT, ids, EOT, 0-pads] per class name in the id order of the reference's vendored
    tokenizer (specials at 512 / 513, merges two places later than OpenAI's) -- the golden file's ids verbatim."""
    from tunevlseg_amd.nets.denseclip import tokenize

    names = ["polyp", "Skin Melanoma", "a photo of a"]
    by_text = {c["text"]: c["ids"] for c in GOLDEN["cases"]}
    t = tokenize(names, 8, REF_BPE)
    assert t.shape == (3, 8) and t.dtype == torch.long
    for row, name in zip(t.tolist(), names):
        want = [GOLDEN["bos"], *by_text[name], GOLDEN["eos"]]
        assert row == want + [0] * (8 - len(want)), name
    hf = tokenize(names, 8, REF_BPE, vendored_ids=False)
    assert hf[0, 0].item() == 49406 and hf[0].max().item() == 49407
    with pytest.raises(RuntimeError, match="too long for context length"):
        tokenize(["one small pink round polyp located in center of the image."], 5, REF_BPE)
