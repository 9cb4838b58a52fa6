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
                   TVL_DIST_BACKEND="gloo", PYTHONPATH=str(root))
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
