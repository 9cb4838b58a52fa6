"""Update path on the GPU: two fused-AdamW train steps against the CPU oracle + torch.optim.AdamW, and the fit loop."""
from functools import partial

import pytest
import torch

pytestmark = pytest.mark.gpu


def tiny_module(depth=2, n=4, new_last=True, lr=2e-3, weight_decay=0.0, seed=11):
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import VPTContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule, ReduceLROnPlateau

    torch.manual_seed(0)
    net = nets.VPTCLIPSeg(context_learner=partial(VPTContextLearner, prompt_depth=depth, num_context=n),
                          model_cfg={"pretrained_model_name_or_path": f"random:tiny:seed={seed}"}, use_new_last_layer=new_last)
    return ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), optimizer=partial(FusedAdamW, lr=lr),
                               scheduler=partial(ReduceLROnPlateau, mode="min", factor=0.2, patience=5), weight_decay=weight_decay).cuda()


def test_two_adamw_steps_match_oracle():
    from oracle import clipseg_oracle as O
    from tunevlseg_amd.config import CLIPSegConfig
    from tunevlseg_amd.weights import init_clipseg_state_dict

    module = tiny_module(weight_decay=0.01)
    net = module.net
    g = torch.Generator().manual_seed(3)
    pix = torch.randn(3, 3, 64, 64, generator=g)
    ids = torch.tensor([[62, 5, 9, 63, 1, 1], [62, 7, 11, 13, 63, 1], [62, 8, 63, 1, 1, 1]])
    am = (ids != 1).long()
    mask = (torch.rand(3, 1, 64, 64, generator=g) > 0.7).float()
    # CPU reference: oracle forward + torch AdamW with the same decay / no-decay split
    cfg, sd = CLIPSegConfig.tiny(), init_clipseg_state_dict(CLIPSegConfig.tiny(), 11)
    ctx = net.context_learner.context_vectors.detach().cpu().clone().requires_grad_(True)
    conv = net.additive_decoder_layer[1]
    cw, cb = conv.weight.detach().cpu().clone().requires_grad_(True), conv.bias.detach().cpu().clone().requires_grad_(True)
    ropt = torch.optim.AdamW([{"params": [cw], "weight_decay": 0.01}, {"params": [ctx, cb], "weight_decay": 0.0}], lr=2e-3)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    batch = {"image": pix.cuda(), "input_ids": ids.cuda(), "attention_mask": am.cuda(), "mask": mask.cuda()}
    for _ in range(2):
        opt.zero_grad()
        module.training_step(batch).backward()
        opt.step()
        ropt.zero_grad()
        logits = O.vpt_forward(sd, cfg, {"kind": "vpt", "ctx": ctx}, pix, ids, am, (cw, cb, torch.tensor(0.5)))
        O.dice_ce_loss(logits, mask).backward()
        ropt.step()
    for mine, ref, name in ((net.context_learner.context_vectors, ctx, "ctx"), (conv.weight, cw, "conv_w"), (conv.bias, cb, "conv_b")):
        err = (mine.detach().cpu() - ref.detach()).abs().max().item()
        # Adam's first steps move every entry by ~lr regardless of gradient scale: compare against lr
        assert err <= 2e-3 * 2e-2, f"{name}: {err:.3e}"
    m = module.epoch_metrics("train")
    assert 0.0 <= m["train_dice"] <= 1.0 and 0.0 <= m["train_iou"] <= 1.0


def test_fit_reduces_loss_and_checkpoints_round_trip(tmp_path):
    from tunevlseg_amd.trainer import SyntheticImageTextMaskLoader, Trainer

    module = tiny_module(depth=3, n=4, new_last=True, lr=5e-3)
    tr = SyntheticImageTextMaskLoader(4, 4, 64, "cuda", seed=1, vocab=64, bos=62, eos=63, pad=1, max_len=6)
    va = SyntheticImageTextMaskLoader(2, 4, 64, "cuda", seed=2, vocab=64, bos=62, eos=63, pad=1, max_len=6)
    logs = []
    trainer = Trainer(max_epochs=6, min_epochs=1, default_root_dir=str(tmp_path), log_fn=logs.append)
    final = trainer.fit(module, tr, va)
    first = float(logs[0].split("train_loss=")[1].split()[0])
    assert final["train_loss"] < first - 0.01, (first, final)
    assert trainer.best_path is not None and (tmp_path / "last.ckpt").exists()
    before = {k: p.detach().clone() for k, p in module.named_parameters() if p.requires_grad}
    with torch.no_grad():
        for p in module.parameters():
            if p.requires_grad:
                p.add_(1.0)
    trainer.load(module, tmp_path / "last.ckpt")
    for k, p in module.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.detach(), before[k]), k
    test_metrics = trainer.test(module, va, ckpt_path="best")
    assert set(test_metrics) == {"test_dice", "test_iou", "test_loss"}


def test_cris_two_adamw_steps_match_oracle():
    """BASELINE configs[2] update path at reduced width: COOPCRIS + CoOp prompts + new last layer, two AdamW steps."""
    from oracle import clipseg_oracle as O
    from oracle import cris_oracle as OC
    from tunevlseg_amd import nets
    from tunevlseg_amd.cris_config import CRISConfig
    from tunevlseg_amd.nets.context_learner import CoOpContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule
    from tunevlseg_amd.weights import init_cris_state_dict

    cfg = CRISConfig.tiny()
    sd = init_cris_state_dict(cfg, 31)
    torch.manual_seed(0)
    net = nets.COOPCRIS(model_cfg={"clip_pretrain": {"config": cfg, "state_dict": sd}, "img_size": cfg.img_size},
                        context_learner=partial(CoOpContextLearner, prompt_depth=2, num_context=3), use_new_last_layer=True)
    module = ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), optimizer=partial(FusedAdamW, lr=2e-3),
                                 scheduler=None, weight_decay=0.01).cuda()
    g = torch.Generator().manual_seed(5)
    pix = torch.randn(3, 3, cfg.img_size, cfg.img_size, generator=g)
    ids = torch.tensor([[62, 5, 9, 63, 0, 0], [62, 7, 11, 13, 63, 0], [62, 8, 63, 0, 0, 0]])
    mask = (torch.rand(3, 1, cfg.img_size, cfg.img_size, generator=g) > 0.7).float()
    ctx = net.context_learner.context_vectors.detach().cpu().clone().requires_grad_(True)
    c0, c2 = net.additive_decoder_layer[0], net.additive_decoder_layer[2]
    w1 = c0.weight.detach().cpu().clone().requires_grad_(True)
    cw, cb = c2.weight.detach().cpu().clone().requires_grad_(True), c2.bias.detach().cpu().clone().requires_grad_(True)
    ratio = net.residual_ratio.detach().cpu().clone().requires_grad_(True)
    ropt = torch.optim.AdamW([{"params": [w1, cw], "weight_decay": 0.01}, {"params": [ctx, cb, ratio], "weight_decay": 0.0}], lr=2e-3)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    am = (ids != 0).long()
    batch = {"image": pix.cuda(), "input_ids": ids.cuda(), "attention_mask": am.cuda(), "mask": mask.cuda()}
    for _ in range(2):
        opt.zero_grad()
        module.training_step(batch).backward()
        opt.step()
        ropt.zero_grad()
        logits = OC.cris_forward(sd, cfg, {"kind": "coop", "ctx": ctx}, pix, ids, am, {"w1": w1, "w": cw, "b": cb, "ratio": ratio})
        O.dice_ce_loss(logits, mask).backward()
        ropt.step()
    for mine, ref, name in ((net.context_learner.context_vectors, ctx, "ctx"), (c0.weight, w1, "w1"), (c2.weight, cw, "conv_w"),
                            (c2.bias, cb, "conv_b"), (net.residual_ratio, ratio, "ratio")):
        err = (mine.detach().cpu() - ref.detach()).abs().max().item()
        assert err <= 2e-3 * 2e-2, f"{name}: {err:.3e}"


def test_cris_unfrozen_projector_head_two_adamw_steps_match_oracle():
    """no_freeze_last_layer on CRIS (reference coop_cris.py:88-94): proj.txt and the projector's last 1x1 conv train next to the
    prompts; two AdamW steps against the oracle, and the prepared (frozen) matrices are built once, not once per step."""
    from oracle import clipseg_oracle as O
    from oracle import cris_oracle as OC
    from tunevlseg_amd import nets
    from tunevlseg_amd.cris_config import CRISConfig
    from tunevlseg_amd.nets.context_learner import CoOpContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule
    from tunevlseg_amd.weights import init_cris_state_dict

    cfg = CRISConfig.tiny()
    sd = init_cris_state_dict(cfg, 31)
    torch.manual_seed(0)
    net = nets.COOPCRIS(model_cfg={"clip_pretrain": {"config": cfg, "state_dict": sd}, "img_size": cfg.img_size},
                        context_learner=partial(CoOpContextLearner, prompt_depth=2, num_context=3), use_new_last_layer=False,
                        no_freeze_last_layer=True)
    head = ("proj.txt.weight", "proj.txt.bias", "proj.vis.4.weight", "proj.vis.4.bias")
    assert {k for k, p in net.named_parameters() if p.requires_grad} == {"context_learner.context_vectors", *head}
    module = ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), optimizer=partial(FusedAdamW, lr=2e-3),
                                 scheduler=None, weight_decay=0.01).cuda()
    g = torch.Generator().manual_seed(6)
    pix = torch.randn(2, 3, cfg.img_size, cfg.img_size, generator=g)
    ids = torch.tensor([[62, 5, 9, 63, 0, 0], [62, 7, 11, 13, 63, 0]])
    mask = (torch.rand(2, 1, cfg.img_size, cfg.img_size, generator=g) > 0.7).float()
    own = dict(net.named_parameters())
    ctx = net.context_learner.context_vectors.detach().cpu().clone().requires_grad_(True)
    ref = {k: own[k].detach().cpu().clone().requires_grad_(True) for k in head}
    sd_ref = dict(sd)
    sd_ref.update(ref)
    ropt = torch.optim.AdamW([{"params": [ref["proj.txt.weight"], ref["proj.vis.4.weight"]], "weight_decay": 0.01},
                              {"params": [ctx, ref["proj.txt.bias"], ref["proj.vis.4.bias"]], "weight_decay": 0.0}], lr=2e-3)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    am = (ids != 0).long()
    batch = {"image": pix.cuda(), "input_ids": ids.cuda(), "attention_mask": am.cuda(), "mask": mask.cuda()}
    preps = []
    for _ in range(2):
        opt.zero_grad()
        module.training_step(batch).backward()
        opt.step()
        preps.append(net.weights.prepared())
        ropt.zero_grad()
        logits = OC.cris_forward(sd_ref, cfg, {"kind": "coop", "ctx": ctx}, pix, ids, am, None)
        O.dice_ce_loss(logits, mask).backward()
        ropt.step()
    assert preps[0] is preps[1]
    for k in head:
        err = (own[k].detach().cpu() - ref[k].detach()).abs().max().item()
        assert err <= 2e-3 * 2e-2, f"{k}: {err:.3e}"
        assert (own[k].detach().cpu() - sd[k].reshape(own[k].shape)).abs().max().item() > 1e-4, f"{k} did not train"
    err = (net.context_learner.context_vectors.detach().cpu() - ctx.detach()).abs().max().item()
    assert err <= 2e-3 * 2e-2, f"ctx: {err:.3e}"


def test_two_rank_step_on_hip_path_equals_global_batch_step(tmp_path):
    """N > 1 on the product path: two fresh child processes (ranks 0 / 1, gloo, one device) each run the HIP net on their half of
    the global batch; the bucketed all-reduce rides on the backward (tunevlseg_amd.dist.GradExchange) and the fused AdamW
    averages.  The parameters after two steps must equal a single process stepping on the whole batch (DDP semantics of the
    reference's trainer=ddp, configs/trainer/ddp.yaml:4-9).  The scaling curve itself is NOT measured here."""
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TVL_DIST_BACKEND="gloo", TVL_ALLOW_SHARED_DEVICE="1", PYTHONPATH=str(root))
        procs.append(subprocess.Popen([sys.executable, str(root / "tests" / "ddp_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", PYTHONPATH=str(root))
    single = subprocess.run([sys.executable, str(root / "tests" / "ddp_worker.py"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stdout + single.stderr
    two, one = torch.load(tmp_path / "world2_rank0.pt"), torch.load(tmp_path / "world1_rank0.pt")
    other = torch.load(tmp_path / "world2_rank1.pt")
    assert two["launched_in_backward"] >= 1  # the exchange was enqueued from inside backward, not after it
    for k in one["params"]:
        assert torch.equal(two["params"][k], other["params"][k]), f"ranks diverged on {k}"
        err = (two["params"][k] - one["params"][k]).abs().max().item()
        assert err <= 2e-3 * 2e-2, f"{k}: {err:.3e}"  # Adam's first steps move every entry by ~lr: compare against lr (as above)


def test_rccl_group_of_one_rank_runs_the_exchange_and_changes_nothing(tmp_path):
    """The RCCL half of the N > 1 path on the one GPU of this box: ONE process, a process group of a single rank on backend "nccl"
    (TVL_DIST_SINGLE_RANK_GROUP=1): the bucketed all-reduces are enqueued asynchronously from the backward's post-accumulate hooks -- on gradients
    that the side stream produced, too (MaPLe's text tower) --, waited for in the optimiser step, followed by the fused AdamW and a barrier.  A sum
    over one rank is the identity: the parameters after two steps must equal the plain single-process run bit for bit."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    runs = {}
    for tag, extra in (("", {}), ("_rccl1", {"TVL_DIST_SINGLE_RANK_GROUP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29531"})):
        env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", PYTHONPATH=str(root), TVL_WORKER_TAG=tag, **extra)
        env.pop("TVL_DIST_BACKEND", None)
        r = subprocess.run([sys.executable, str(root / "tests" / "ddp_worker.py"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        runs[tag] = torch.load(tmp_path / f"world1_rank0{tag}.pt")
    plain, rccl = runs[""], runs["_rccl1"]
    assert plain["backend"] is None and rccl["backend"] == "nccl"
    assert rccl["launched_in_backward"] >= 1 and plain["launched_in_backward"] == 0
    for k in plain["params"]:
        assert torch.equal(plain["params"][k], rccl["params"][k]), k


def test_train_entry_point_fits_from_a_directory_in_the_reference_layout(tmp_path, monkeypatch):
    """``python -m tunevlseg_amd.train experiment=...`` end to end: a config tree with the reference's structure whose ``data`` node is
    the reference's datamodule schema (``configs/data/image_text_mask.yaml``) over a toy dataset in the reference's wire format
    (images/, masks/, anns/{train,val,test}.json) -> cfg.data is instantiated, decoded samples are resized / augmented / normalised on
    the device, fit + test run.  A config WITHOUT a buildable data node raises instead of training on synthetic batches."""
    import json

    import numpy as np
    from PIL import Image

    from tests.test_config_loader import write
    from tunevlseg_amd import config_loader as CL
    from tunevlseg_amd import train as T

    root = tmp_path / "data" / "toy"
    for d in ("images", "masks", "anns"):
        (root / d).mkdir(parents=True)
    rng = np.random.default_rng(0)
    tasks = []
    for i in range(6):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        m = np.zeros((h, w), np.uint8)
        m[h // 4: h // 2, w // 4: 3 * w // 4] = 255
        img[m > 0] = (img[m > 0] // 4 + 190).astype(np.uint8)   # a bright box: something to learn
        Image.fromarray(img).save(root / "images" / f"im{i}.png")
        Image.fromarray(m).save(root / "masks" / f"m{i}.png")
        tasks.append({"img_name": f"im{i}.png", "mask_name": f"m{i}.png", "prompts": {"p0": "", "p1": "a bright box"}})
    for split in ("train", "val", "test"):
        (root / "anns" / f"{split}.json").write_text(json.dumps(tasks))
    # the tiny backbone has a 64-word vocabulary (BOS 62, EOS 63): a word-level stand-in for the CLIP tokenizer that belongs to it
    class TinyTokenizer:
        pad_token_id, bos_token_id, eos_token_id = 1, 62, 63

        def __call__(self, text, **kw):
            ids = [self.bos_token_id, *[2 + (sum(map(ord, w)) % 58) for w in text.lower().split()], self.eos_token_id]
            return {"input_ids": ids, "attention_mask": [1] * len(ids)}

    from tunevlseg_amd.data import dataset as D

    monkeypatch.setattr(D, "resolve_tokenizer", lambda tokenizer=None, path=None, mml=None: tokenizer or TinyTokenizer())

    cfgdir = tmp_path / "configs"
    write(cfgdir / "train.yaml", "# @package _global_\ndefaults:\n  - _self_\n  - data: image_text_mask\n  - model: vpt\n  - trainer: default\n"
                                 "  - experiment: null\ntask_name: train\ntrain: true\ntest: true\nseed: 7\n")
    write(cfgdir / "trainer" / "default.yaml", "max_epochs: 2\nmin_epochs: 1\naccumulate_grad_batches: 1\ncheck_val_every_n_epoch: 1\n")
    write(cfgdir / "model" / "vpt.yaml", """_target_: src.models.image_text_mask_module.ImageTextMaskModule
net:
  _target_: src.models.core_models.coop.VPTCLIPSeg
  model_cfg: {pretrained_model_name_or_path: "random:tiny:seed=11", freeze_encoder: false, freeze_decoder: false}
  context_learner: {_target_: src.models.core_models.coop.context_learner.VPTContextLearner, _partial_: true, prompt_depth: 2, num_context: 4, vector_std: 0.02}
  freeze_all: true
  use_new_last_layer: true
loss_fn: {_target_: monai.losses.DiceCELoss, sigmoid: true, lambda_dice: 1, lambda_ce: 0.2}
weight_decay: 0.0
optimizer: {_target_: torch.optim.AdamW, _partial_: true, lr: 1.0e-2}
scheduler: null
compile: false
task: binary
threshold: 0.5
""")
    ds = lambda split, tf: f"""  _target_: src.data.core_datasets.ImageTextMaskDataset
  image_dir: ${{dataset_root}}/images
  mask_dir: ${{dataset_root}}/masks
  task_path: ${{dataset_root}}/anns/{split}.json
  tokenizer_pretrained_path: ${{tokenizer_pretrained_path}}
  prompt_index: ${{prompt_index}}
  override_prompt: null
  transforms: ${{{tf}}}
  model_max_length: null
  return_tensors: pt
  collate_fn: ${{collate_fn}}
  insert_stop_at_last: true
"""  # noqa: E731
    write(cfgdir / "data" / "image_text_mask.yaml", "_target_: src.data.image_text_mask_datamodule.ImageTextDatamodule\ntrain_ds:\n" + ds("train", "train_transforms")
          + "val_ds:\n" + ds("val", "val_transforms") + "test_ds:\n" + ds("test", "val_transforms") + "batch_size: 4\nnum_workers: 0\ndrop_last: false\npin_memory: false\n")
    write(cfgdir / "experiment" / "toy.yaml", f"""# @package _global_
data_root: {tmp_path / 'data'}
ds_name: toy
dataset_root: ${{data_root}}/${{ds_name}}
tokenizer_pretrained_path: CIDAS/clipseg-rd64
prompt_index: 1
img_size: 64
train_transforms:
  _target_: albumentations.Compose
  transforms:
    - {{_target_: albumentations.Resize, height: "${{img_size}}", width: "${{img_size}}", interpolation: "${{import_eval:cv2.INTER_CUBIC}}"}}
    - {{_target_: albumentations.Affine, scale: [0.98, 1.02], translate_percent: [-0.02, 0.02], rotate: [-5, 5], interpolation: "${{import_eval:cv2.INTER_CUBIC}}", mode: "${{import_eval:cv2.BORDER_REPLICATE}}", p: 0.5}}
    - {{_target_: albumentations.PadIfNeeded, min_height: "${{img_size}}", min_width: "${{img_size}}", border_mode: "${{import_eval:cv2.BORDER_REPLICATE}}"}}
    - {{_target_: albumentations.CropNonEmptyMaskIfExists, width: "${{img_size}}", height: "${{img_size}}"}}
    - {{_target_: albumentations.RandomBrightnessContrast, contrast_limit: 0.1, brightness_limit: 0.1, p: 0.5}}
    - {{_target_: albumentations.Normalize, mean: [0.485, 0.456, 0.406], std: [0.229, 0.224, 0.225]}}
    - {{_target_: albumentations.pytorch.ToTensorV2, transpose_mask: true}}
val_transforms:
  _target_: albumentations.Compose
  transforms:
    - {{_target_: albumentations.Resize, height: "${{img_size}}", width: "${{img_size}}", interpolation: "${{import_eval:cv2.INTER_CUBIC}}"}}
    - {{_target_: albumentations.Normalize, mean: [0.485, 0.456, 0.406], std: [0.229, 0.224, 0.225]}}
    - {{_target_: albumentations.pytorch.ToTensorV2, transpose_mask: true}}
collate_fn:
  _target_: src.data.components.data_collator.CustomDataCollatorWithPadding
  tokenizer: {{_target_: transformers.AutoTokenizer.from_pretrained, pretrained_model_name_or_path: "${{tokenizer_pretrained_path}}"}}
  padding_keys: ["input_ids", "attention_mask"]
  padding: true
paths: {{output_dir: {tmp_path / 'out'}}}
""")
    cfg = CL.Composer(cfgdir).compose("train", ["experiment=toy"])
    metrics = T.train(cfg)
    assert {"train_loss", "val_loss", "val_dice", "test_dice", "test_iou"} <= set(metrics), sorted(metrics)
    assert all(np.isfinite(v) for v in metrics.values())
    # no data node -> a loud error, not synthetic batches
    cfg2 = CL.Composer(cfgdir).compose("train", ["experiment=toy"])
    cfg2["data"] = {"batch_size": 4}
    with pytest.raises(ValueError, match="no buildable `data` node"):
        T.train(cfg2)


@pytest.mark.gpu
def test_non_finite_steps_raise_instead_of_logging_nan(tmp_path):
    """The loss and optimiser kernels keep sticky device-side NaN / Inf flags (hip.nonfinite_flags); Trainer.fit reads them at the end of an
    epoch and raises, bench.py exits non-zero: a non-finite step cannot pass as a number."""
    from tunevlseg_amd import hip, ops
    from tunevlseg_amd.trainer import SyntheticImageTextMaskLoader, Trainer

    hip.check_finite()   # clear whatever earlier tests left
    logits = torch.randn(2, 1, 16, 16).cuda().requires_grad_(True)
    mask = (torch.rand(2, 1, 16, 16) > 0.5).float().cuda()
    loss, _ = ops.DiceCELossFn.apply(logits, mask, 1.0, 0.2, 0.5)
    assert torch.isfinite(loss).item()
    hip.check_finite()
    bad = logits.detach().clone()
    bad[0, 0, 3, 3] = float("nan")
    loss, _ = ops.DiceCELossFn.apply(bad, mask, 1.0, 0.2, 0.5)
    with pytest.raises(FloatingPointError, match="non-finite loss"):
        hip.check_finite()
    hip.check_finite()   # reading cleared the flags
    # a NaN gradient that reaches the optimiser: finite loss, poisoned parameter gradient
    module = tiny_module(depth=1)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    opt.zero_grad()
    opt.flat.grad[3] = float("inf")
    opt.step()
    with pytest.raises(FloatingPointError, match="non-finite gradients"):
        hip.check_finite()
    # ... and through the trainer: parameters made NaN -> every loss NaN -> fit raises at the end of the first epoch
    module = tiny_module(depth=1)
    with torch.no_grad():
        module.net.context_learner.context_vectors.fill_(float("nan"))
    tr = SyntheticImageTextMaskLoader(2, 2, 64, "cuda", seed=1, vocab=64, bos=62, eos=63, pad=1, max_len=6)
    with pytest.raises(FloatingPointError):
        Trainer(max_epochs=2, min_epochs=1, default_root_dir=str(tmp_path), log_fn=lambda *_: None).fit(module, tr, None)
    hip.check_finite()
