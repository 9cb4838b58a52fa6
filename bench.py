#!/usr/bin/env python
"""Headline benchmark: images/sec of the prompt-tuning TRAIN STEP (forward + DiceCE + backward + AdamW on the
prompt parameters) for CLIPSeg ViT-B/16 + VPT-shallow (10 visual prompts), 352x352, bs=32 per GPU
(BASELINE.json configs[1]); weak scaling over the GPUs of one node (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Synthetic data (images N(0,1), masks U(0,1)>0.7, fixed token rows),
seeded random weights of the named architecture (no checkpoint is reachable offline).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time
from functools import partial
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

GFLOP_PER_IMAGE_TRAIN = 167.1  # VPT-10 shallow 352^2, fwd 80.6 + bwd 86.5 (FlopCounterMode on the reference)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (exact-fp32 mode only)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# fp32 products cost this many bf16 MFMA products in each GEMM mode -> ceiling in algorithmic (fp32) TFLOP/s
MODE_PEAK = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16": PEAK_BF16_MFMA_TFLOPS}
MODE_DTYPE = {"f32": "f32", "bf16x6": "f32 (3xbf16-split operands on bf16 MFMA, fp32 accumulate; fp32-equivalent)",
              "bf16x3": "f32 in/out, 2xbf16-split operands (reduced precision)", "bf16": "bf16 operands, fp32 accumulate (reduced precision)"}


def make_batch(B: int, size: int, seed: int, device, pad_id: int = 1):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    L = 8
    ids = torch.full((B, L), pad_id, dtype=torch.long)
    am = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n_words = 2 + (b % 4)
        row = [49406, *torch.randint(320, 40000, (n_words,), generator=g).tolist(), 49407]
        ids[b, : len(row)] = torch.tensor(row)
        am[b, : len(row)] = 1
    mask = (torch.rand(B, 1, size, size, generator=g) > 0.7).float()
    return {"image": img.to(device), "input_ids": ids.to(device), "attention_mask": am.to(device), "mask": mask.to(device)}


class _PromptTokenizer:
    """Stand-in for ``AutoTokenizer("CIDAS/clipseg-rd64")`` (not reachable offline): CLIP BPE ids of "a photo of a"."""

    def __call__(self, text, **kw):
        class R:
            pass

        r = R()
        r.input_ids = torch.tensor([[320, 1125, 539, 320]] * (1 if isinstance(text, str) else len(text)), dtype=torch.long)
        return r


def build_cris_module(device, seed: int = 0):
    """BASELINE configs[2]: CRIS + CoCoOp meta-net (reference configs/model/cocoop/cris.yaml), 416x416."""
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import CoCoOpContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    torch.manual_seed(12345)
    net = nets.COOPCRIS(
        model_cfg={"clip_pretrain": f"random:rn50:seed={seed}", "img_size": 416, "freeze_encoder": True, "cris_pretrain": None},
        context_learner=partial(CoCoOpContextLearner, norm_image_features=False, prompt_depth=1, use_unified_projection=False,
                                intermediate_dim=64, use_proj_norm=True, use_lora_proj=False, num_context=4,
                                context_initializer="a photo of a", vector_std=0.02, tokenizer=_PromptTokenizer()),
        freeze_all=True, no_freeze_last_layer=False, use_new_last_layer=True, new_last_layer_kernel_size=5, residual_ratio=0.5)
    module = ImageTextMaskModule(net=net, loss_fn=DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2),
                                 optimizer=partial(FusedAdamW, lr=2e-5), scheduler=None, compile=False, task="binary",
                                 threshold=0.5, weight_decay=0.0).to(device)
    module.setup("fit")
    return module, module.configure_optimizers()["optimizer"]


def build_maple_module(device, seed: int = 0):
    """BASELINE configs[3]: CLIPSeg + MaPLe coupled prompts, depth 9, 4 context tokens (reference configs/model/maple_clipseg.yaml)."""
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import MapleContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    torch.manual_seed(12345)
    net = nets.MapleCLIPSeg(
        context_learner=partial(MapleContextLearner, prompt_depth=9, num_context=4, vector_std=0.02, use_unified_projection=False,
                                intermediate_dim=64, use_proj_norm=True, use_lora_proj=False),
        model_cfg={"pretrained_model_name_or_path": f"random:rd64:seed={seed}", "freeze_encoder": False, "freeze_decoder": False},
        freeze_all=True, no_freeze_last_layer=False, use_new_last_layer=True, new_last_layer_kernel_size=5, residual_ratio=0.5)
    module = ImageTextMaskModule(net=net, loss_fn=DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2),
                                 optimizer=partial(FusedAdamW, lr=2e-4), scheduler=None, compile=False, task="binary",
                                 threshold=0.5, weight_decay=0.0).to(device)
    module.setup("fit")
    return module, module.configure_optimizers()["optimizer"]


def build_module(device, num_context: int = 10, prompt_depth: int = 1, seed: int = 0, cond_cache: bool = False):
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import VPTContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    torch.manual_seed(12345)  # configs/experiment/coop/clipseg.yaml:19
    net = nets.VPTCLIPSeg(
        context_learner=partial(VPTContextLearner, prompt_depth=prompt_depth, num_context=num_context, vector_std=0.02),
        model_cfg={"pretrained_model_name_or_path": f"random:rd64:seed={seed}", "freeze_encoder": False, "freeze_decoder": False},
        freeze_all=True, no_freeze_last_layer=False, use_new_last_layer=False,  # authors' setting (scripts/schedule_vpt.sh:14,21)
        cache_text_features=cond_cache)
    module = ImageTextMaskModule(net=net, loss_fn=DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2),
                                 optimizer=partial(FusedAdamW, lr=2e-4), scheduler=None, compile=False, task="binary",
                                 threshold=0.5, weight_decay=0.0).to(device)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    return module, opt


PEAK_HBM_TBS = 8.0   # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def rocprof_avg_us(workload: str, kernel: str):
    """Average duration of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of the same command (profiles/), or None."""
    import csv

    stem = {"cris": "cris_kernel_stats.csv", "maple": "maple_kernel_stats.csv", "vit640": "vit640_kernel_stats.csv", "denseclip": "denseclip_kernel_stats.csv"}.get(workload, "kernel_stats.csv")
    f = next((c for c in (ROOT / "profiles" / f"r4_{stem}", ROOT / "profiles" / f"r3_{stem}") if c.exists()), None)
    if f is None:
        return None, None
    want = kernel.replace(" ", "")
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = (row.get("Name") or row.get("KernelName") or "").replace(" ", "")
            if want in name:
                try:
                    return round(float(row.get("AverageNs") or row.get("Average") or 0.0) / 1e3, 1), str(f.relative_to(ROOT))
                except ValueError:
                    return None, None
    return None, None


def cpu_baseline(steps32: int = 3, steps4: int = 10) -> dict:
    """The reference's trainer=cpu path on the host cores of this box: the CPU oracle -- a port pinned by the
    golden fixtures; the reference's own Python cannot travel -- runs the identical train step (forward, DiceCE, backward, AdamW),
    fp32, ``set_float32_matmul_precision("medium")`` (reference src/models/__init__.py:6), all host threads.
    Two bounded samples: the headline workload's own shape (VPT-10 shallow, bs 32) and config C1 (CoOp-4, bs 4)."""
    from oracle import clipseg_oracle as O
    from tunevlseg_amd.config import CLIPSegConfig
    from tunevlseg_amd.weights import init_clipseg_state_dict

    torch.set_float32_matmul_precision("medium")
    cores = torch.get_num_threads()
    cfg = CLIPSegConfig.rd64()
    sd = init_clipseg_state_dict(cfg, 0)

    def timed(step, warm: int, n: int) -> list[float]:
        for _ in range(warm):
            step()
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            step()
            out.append(time.perf_counter() - t0)
        return out

    def median(v):
        v = sorted(v)
        return 0.5 * (v[(len(v) - 1) // 2] + v[len(v) // 2])

    # C2 shape: VPT-10 shallow, bs 32, seed 1
    vctx = (torch.randn(1, 10, 768, generator=torch.Generator().manual_seed(1)) * 0.02).requires_grad_(True)
    vopt = torch.optim.AdamW([vctx], lr=2e-4)
    b32 = make_batch(32, 352, 1, "cpu")

    def vpt_step():
        vopt.zero_grad()
        logits = O.vpt_forward(sd, cfg, {"kind": "vpt", "ctx": vctx}, b32["image"], b32["input_ids"], b32["attention_mask"])
        O.dice_ce_loss(logits, b32["mask"]).backward()
        vopt.step()

    t32 = timed(vpt_step, 1, steps32)  # ~27 s per step on the host cores: 1 warm-up, every timed step reported
    # C1: CoOp-4 (4 context tokens, depth 1), bs 4, seed 0
    cctx = (torch.randn(1, 4, 512, generator=torch.Generator().manual_seed(0)) * 0.02).requires_grad_(True)
    copt = torch.optim.AdamW([cctx], lr=2e-4)
    b4 = make_batch(4, 352, 0, "cpu")

    def coop_step():
        copt.zero_grad()
        logits = O.coop_forward(sd, cfg, {"kind": "coop", "ctx": cctx}, b4["image"], b4["input_ids"], b4["attention_mask"])
        O.dice_ce_loss(logits, b4["mask"]).backward()
        copt.step()

    t4 = timed(coop_step, 3, steps4)   # 3 warm-up + 10 timed steps, median
    torch.set_float32_matmul_precision("highest")
    return {"value": round(32 / median(t32), 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"median of {steps32} train steps (1 warm-up; each timed step's seconds listed) of the headline workload's own shape -- VPT-10 shallow, 352x352, bs 32 -- "
                      "on the fp32 torch CPU oracle",
            "step_seconds_bs32": [round(t, 2) for t in t32],
            "c1_coop4_bs4": {"value": round(4 / median(t4), 3), "unit": "images/s",
                             "sample": f"median of {steps4} train steps (3 warm-up), CLIPSeg + CoOp-4, 352x352, bs 4 (BASELINE configs[0])"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="forward + backward of the step replayed as a hipGraph (tunevlseg_amd/graph.py); opt-in: the eager step is what the default line measures")
    ap.add_argument("--cond-cache", action="store_true",
                    help="vpt only: keep the frozen text tower's conditional embeddings per distinct token row (skips work: NOT the headline number; "
                         "the line is marked cond_cache=true)")
    ap.add_argument("--cpu-steps32", type=int, default=3, help="timed CPU-oracle steps at the headline shape (bs 32)")
    ap.add_argument("--cpu-steps4", type=int, default=10, help="timed CPU-oracle steps of config C1 (CoOp-4, bs 4)")
    ap.add_argument("--workload", choices=("vpt", "cris", "maple", "vit640", "denseclip"), default="vpt",
                    help="vpt = BASELINE configs[1] (the headline line); cris = configs[2] (CRIS + CoCoOp, 416x416) and maple = configs[3] "
                         "(MaPLe depth 9) are reported for DESIGN.md; vit640 = the ViT-B/16 encoder leg of configs[4] (640x640, bs 16: "
                         "forward + data gradient of the 12 layers); denseclip = configs[4] up to the mmseg neck / head: ViT-B/16 backbone with its four FPN taps, "
                         "context text encoder, context decoder, pixel-text score map, backward into contexts / gamma, AdamW (640x640, bs 16, 20 classes)")
    args = ap.parse_args()

    from tunevlseg_amd import dist as tdist
    from tunevlseg_amd import hip

    rank, local_rank, world = tdist.init_distributed("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("TVL_DIST_BACKEND") == "gloo":  # rehearsal on a 1-GPU box: all ranks share device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    hip.load()
    if args.graph:
        from tunevlseg_amd.graph import GraphedStep, use_private_stream

        use_private_stream(device)   # before the parameters exist

    cris, maple, vit640 = args.workload == "cris", args.workload == "maple", args.workload == "vit640"
    denseclip = args.workload == "denseclip"
    if denseclip:
        # BASELINE configs[4] up to what reaches the mmseg neck / head (reference denseclip.py:136-169 over models.py:530-714,805-960): the frozen
        # ViT-B/16 backbone with its four FPN taps (no tape), the context text encoder over the 20 class prompts, the context decoder against
        # the 1 + 40^2 projected tokens, the score map; backward of a fixed cotangent on score map + text embeddings (the head's loss is mmseg's and
        # not built: its gradient enters here) into contexts / gamma; fused AdamW on them.
        from tunevlseg_amd.denseclip_backbone import DenseCLIPWeights
        from tunevlseg_amd.denseclip_config import DenseCLIPConfig
        from tunevlseg_amd.nets import DenseCLIP
        from tunevlseg_amd.task import FusedAdamW

        if "--batch" not in sys.argv:
            args.batch = 16
        dcfg = DenseCLIPConfig.vitb16_640(num_classes=20)
        g = torch.Generator().manual_seed(100 + rank)
        sot, eot = dcfg.vocab_size - 2, dcfg.vocab_size - 1
        texts = torch.zeros(dcfg.num_classes, dcfg.context_length, dtype=torch.long)
        for k in range(dcfg.num_classes):
            row = [sot, *torch.randint(320, 40000, (1 + k % 3,), generator=g).tolist(), eot]
            texts[k, : len(row)] = torch.tensor(row)
        torch.manual_seed(12345)
        dnet = DenseCLIP(pretrained=DenseCLIPWeights(dcfg, None, seed=0), texts=texts).to(device)
        with torch.no_grad():
            dnet.gamma.fill_(0.3)   # (the reference's 1e-4 start would leave the context decoder's path numerically idle)
        opt = FusedAdamW([dnet.contexts, dnet.gamma], lr=1e-4, weight_decay=1e-4)
        img640 = torch.randn(args.batch, 3, 640, 640, generator=g).to(device)
        G640 = 640 // dcfg.patch_size
        gs640 = torch.randn(args.batch, dcfg.num_classes, G640, G640, generator=g).to(device)
        gt640 = (torch.randn(args.batch, dcfg.num_classes, dcfg.embed_dim, generator=g) * 0.1).to(device)
        T640, D640, E640, K640, Dd = 1 + G640 * G640, dcfg.width, dcfg.output_dim, dcfg.num_classes, dcfg.decoder_width
        fl = dcfg.layers * (24.0 * T640 * D640 * D640 + 4.0 * T640 * T640 * D640)                        # the 12 encoder layers, forward only
        fl += 2.0 * (T640 - 1) * D640 * (3 * 16 * 16) + 2.0 * T640 * D640 * E640                         # patch conv, ln_post @ proj
        fl += 2.0 * (T640 - 1) * D640 * 4 * D640 * (1 + 4 + 1)                                           # fpn1 (two transposed convs: HW and 4 HW pixels), fpn2
        fl += 2.0 * T640 * E640 * Dd + dcfg.decoder_layers * (2 * 2.0 * T640 * Dd * Dd + 4.0 * K640 * T640 * Dd)   # memory projection, cross-attention K / V + products
        fl += 2.0 * (T640 - 1) * E640 * K640 * 2                                                         # score map, and its gradient w.r.t. the text side
        gflop_per_image = fl / 1e9   # (text tower / decoder rows over 20 classes are < 1 % and left out)
        module = batch = None

        def step():
            opt.zero_grad()
            text_embeddings, maps, score_map = dnet(img640)
            loss = ops_dot(score_map.permute(0, 2, 3, 1), gs_nhwc) + ops_dot(text_embeddings, gt640)   # (the score map lives as [B, H, W, K])
            loss.backward()
            opt.step()
            return loss

        from tunevlseg_amd import hip as _hip
        from tunevlseg_amd import ops as _ops

        class _DotFn(_ops.Fn):   # sum(x * w) with a fixed w: the stand-in for the head's loss, one launch each way
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(w)
                ctx.shape = x.shape
                return _hip.dot(_ops._c(x).view(-1), w.view(-1)).view(())

            @staticmethod
            def backward(ctx, d):
                (w,) = ctx.saved_tensors
                return _hip.scale_dev(w.view(-1), d.reshape(1).contiguous(), False).view(ctx.shape), None

        gs_nhwc = gs640.permute(0, 2, 3, 1).contiguous()

        def ops_dot(x, w):
            return _DotFn.apply(x, w)
    elif vit640:
        # configs[4]'s encoder leg (reference src/models/components/denseclip/models.py:530-714: 12 ResidualAttentionBlocks of width 768 over
        # 1 + 40^2 tokens): forward + data gradient through the frozen layers -- what a prompt-tuned DenseCLIP step would spend in the tower
        from tunevlseg_amd import ops
        from tunevlseg_amd.backbone import CLIPSegBackbone

        if "--batch" not in sys.argv:
            args.batch = 16
        T640, D640 = 1 + (640 // 16) ** 2, 768
        layers = CLIPSegBackbone.from_spec("random:rd64:seed=0").requires_grad_(False).to(device).prepared()["vision_layers"]
        spec = ops.AttnSpec(heads=12, act=hip.ACT_QUICK_GELU, eps=1e-5)
        g = torch.Generator().manual_seed(100 + rank)
        x640 = torch.randn(args.batch, T640, D640, generator=g).to(device).requires_grad_(True)
        w640 = (torch.randn(args.batch, T640, D640, generator=g) * 1e-3).to(device)
        per_layer_fwd = 24.0 * T640 * D640 * D640 + 4.0 * T640 * T640 * D640          # 8 T D^2 of QKV/out-proj + 16 T D^2 of the MLP, QK^T + PV
        per_layer_bwd = 24.0 * T640 * D640 * D640 + 10.0 * T640 * T640 * D640         # data gradients only; attention backward = 5 products
        gflop_per_image = len(layers) * (per_layer_fwd + per_layer_bwd) / 1e9
        module = opt = batch = None

        def step():
            x640.grad = None
            y = x640
            for lw in layers:
                y = ops.encoder_layer(y, lw, spec)
            loss = (y * w640).sum()
            loss.backward()
            return loss
    else:
        module, opt = build_cris_module(device) if cris else (build_maple_module(device) if maple else build_module(device, cond_cache=args.cond_cache))
        batch = make_batch(args.batch, 416 if cris else 352, 100 + rank, device, pad_id=0 if cris else 1)
        gflop_per_image = 212.8 if cris else (167.8 if maple else GFLOP_PER_IMAGE_TRAIN)  # SURVEY.md §8d (FlopCounterMode on the reference)

        stepper = GraphedStep(module, opt) if args.graph else None

        def eager_step():
            opt.zero_grad()
            loss = module.training_step(batch, 0)
            loss.backward()
            opt.step()
            return loss

        def step():
            if stepper is None:
                return eager_step()
            loss = stepper(batch)
            opt.step()
            return loss

    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()   # world > 1, or a one-rank RCCL group (TVL_DIST_SINGLE_RANK_GROUP=1)

    def barrier():
        if grouped:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    metrics = module.epoch_metrics("train") if module is not None else None
    # sticky device-side flags of the loss / optimiser kernels: a step that went NaN / Inf anywhere in the run ends the benchmark with a non-zero
    # exit code and no result line (a throughput measured on NaNs is not a measurement)
    try:
        hip.check_finite(device, what=f"bench.py rank {rank}")
        if not torch.isfinite(loss.detach()).item():
            raise FloatingPointError(f"bench.py rank {rank}: the last step's loss is {float(loss)}")
    except FloatingPointError as e:
        print(json.dumps({"error": str(e), "rank": rank, "workload": args.workload}), flush=True)
        raise SystemExit(3)

    # ---- roofline of the dominant kernel: HIP events (on torch's current stream = the launch stream) around every GEMM launch of
    # 2 extra, untimed steps.  achieved = algorithmic FLOPs (2*M*N*K per launch) / summed launch time.
    # The profiled steps run on ONE stream (text tower on the main stream): a launch's event pair then brackets that kernel alone -- with the side
    # stream on, the events of the small text-tower GEMMs also count the time they wait for compute units the vision tower's kernels hold.
    from tunevlseg_amd.nets import towers as _towers

    side_was, _towers.TEXT_SIDE_STREAM = _towers.TEXT_SIDE_STREAM, False
    hip.gemm_profile_start()
    for _ in range(2):
        (step if (module is None or not args.graph) else eager_step)()   # per-launch events need the launches: the profiled steps run eagerly
    _towers.TEXT_SIDE_STREAM = side_was
    prof = hip.gemm_profile_stop()
    roofline = None
    if prof:
        # event brackets around back-to-back launches of a few tens of microseconds read ~30 us too long each (rocprofv3: 28 us per launch
        # of the 64x64 tile where the brackets say 57), so the dominant kernel is chosen among those averaging >= 100 us per launch
        long_ones = {k: v for k, v in prof.items() if v["ms"] / v["launches"] >= 0.1} or prof
        name, d = max(long_ones.items(), key=lambda kv: kv[1]["ms"])
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        gemm_ms = sum(v["ms"] for v in prof.values()) / 2
        split = "bf16s" in name or "tp3" in name or "h2m" in name
        two_piece = "gemm_h2m_kernel" in name or re.search(r", 2, (true|false), (true|false)>", name) is not None   # the h2 format: two fp16 pieces, 3 MFMAs per fp32 product
        peak = (PEAK_BF16_MFMA_TFLOPS / 3 if two_piece else MODE_PEAK[hip.GEMM_MODE]) if split else PEAK_F32_MFMA_TFLOPS
        # HBM bytes per launch of that kernel: recorded by two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE; FETCH_SIZE
        # doubled per the gfx950 note of MI355X_MICROARCH.md §HBM) and committed under profiles/.  It is a RECORDED number: reported
        # only when the record names the same kernel instantiation, and always with the file and commit it came from.
        # A record is reported only while the kernel sources are the ones it was taken on (csrc_sha: the GPU box has no .git to ask about
        # ancestry); otherwise traffic is null and traffic_source says which record was refused.
        traffic, traffic_source = None, None
        stem = {"cris": "cris_hbm_traffic.json", "maple": "maple_hbm_traffic.json", "denseclip": "denseclip_hbm_traffic.json"}.get(args.workload, "hbm_traffic.json")
        tf = next((f for f in (ROOT / "profiles" / f"r4_{stem}", ROOT / "profiles" / f"r3_{stem}") if f.exists()), ROOT / "profiles" / f"r4_{stem}")
        if tf.exists():
            rec_all = json.loads(tf.read_text())
            rec = rec_all.get("kernels", rec_all).get(name.replace(", ", ","))
            same = rec_all.get("csrc_sha") is not None and rec_all.get("csrc_sha") == hip.csrc_sha()
            traffic_source = {"file": str(tf.relative_to(ROOT)), "recorded_at_commit": rec_all.get("commit"), "recorded_csrc_sha": rec_all.get("csrc_sha"),
                              "this_build_csrc_sha": hip.csrc_sha(), "kind": "recorded (rocprofv3 --pmc), not measured by this run"}
            if rec and same:
                traffic = rec["hbm_bytes_per_launch"]
            else:
                traffic_source["refused"] = "no record for this kernel" if not rec else "the kernel sources changed since the record was taken"
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "kernel": name,
                    "hbm_algorithmic_bytes": round(d["bytes"] / d["launches"]) if d.get("bytes") else None,
                    "hbm_frac": round(d["bytes"] / (d["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 4) if d.get("bytes") else None,
                    "peak_note": ("algorithmic fp32 FLOP/s ceiling = dense fp16 MFMA peak (2.5 PFLOP/s) / 3 MFMAs per fp32 product" if two_piece else
                                  "algorithmic fp32 FLOP/s ceiling = dense bf16 MFMA peak (2.5 PFLOP/s) / 6 MFMAs per fp32 product") if split
                    else "dense f32-input MFMA peak",
                    "launches_per_step": d["launches"] // 2, "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 1),
                    # the uncorrected event reading, the empty pair's reading that `avg_launch_us` has taken off, and the same kernel's
                    # average in the committed rocprofv3 --kernel-trace --stats summary (profiles/): the three should agree
                    "avg_launch_us_raw_events": round(1e3 * d["raw_ms"] / d["launches"], 1),
                    "event_pair_overhead_us": round(1e3 * hip.last_empty_pair_ms, 1),
                    "rocprof_avg_launch_us": rocprof_avg_us(args.workload, name)[0], "rocprof_source": rocprof_avg_us(args.workload, name)[1],
                    "flops_per_launch": round(d["flops"] / d["launches"]), "all_gemm_ms_per_step": round(gemm_ms, 2),
                    # every GEMM instantiation of the step (one instantiation serves several shapes: the average mixes them)
                    # both rooflines per instantiation: MFMA (flops / the arithmetic's ceiling) and HBM (algorithmic bytes: operands once, every
                    # output once / 8 TB/s), and the bound of a kernel that does the two one after the other (t_mfma + t_hbm): the measured time sits
                    # near that sum where a round of tiles loads, computes and stores in lock-step
                    "gemm_kernels": [dict({"kernel": k.replace("gemm_bf16s_kernel", "bf16s").replace("gemm_f32_kernel", "f32").replace("gemm_tp3_kernel", "tp3").replace("gemm_h2m_kernel", "h2m"),
                                           "ms_per_step": round(v["ms"] / 2, 2), "launches_per_step": v["launches"] // 2, "avg_us": round(1e3 * v["ms"] / v["launches"], 1),
                                           "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)},
                                          **({"mfma_frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3), 3),
                                              "hbm_algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"]),
                                              "hbm_GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 3),
                                              "t_mfma_us": round(v["flops"] / v["launches"] / (PEAK_BF16_MFMA_TFLOPS / 3 * 1e12) * 1e6, 1),
                                              "t_hbm_us": round(v["bytes"] / v["launches"] / (PEAK_HBM_TBS * 1e12) * 1e6, 1),
                                              "frac_of_serial_bound": round((v["flops"] / (PEAK_BF16_MFMA_TFLOPS / 3 * 1e12) + v["bytes"] / (PEAK_HBM_TBS * 1e12)) / (v["ms"] * 1e-3), 3)}
                                             if v.get("bytes") else {}))
                                     for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:6]],
                    "all_gemm_tflops": round(sum(v["flops"] for v in prof.values()) / 2 / (gemm_ms * 1e-3) / 1e12, 2),
                    # the non-GEMM kernels of a vision layer, each against its own roofline: attention on the same MFMA ceiling (its
                    # products are 3 MFMAs on two fp16 pieces), LayerNorm on HBM (algorithmic bytes)
                    "other_kernels": [
                        ({"kernel": k, "bound": "mfma", "ms_per_step": round(v["ms"] / 2, 3), "launches_per_step": v["launches"] // 2,
                          "avg_us": round(1e3 * v["ms"] / v["launches"], 1), "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                          "peak": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3), 4),
                          "hbm_GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} if v["flops"] else
                         {"kernel": k, "bound": "hbm", "ms_per_step": round(v["ms"] / 2, 3), "launches_per_step": v["launches"] // 2,
                          "avg_us": round(1e3 * v["ms"] / v["launches"], 1), "achieved": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "unit": "GB/s",
                          "peak": PEAK_HBM_TBS * 1e3, "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)})
                        for k, v in sorted(hip.last_aux_profile.items(), key=lambda kv: -kv[1]["ms"])]}

    step_ceiling = PEAK_BF16_MFMA_TFLOPS / 3 if (hip.GEMM_MODE == "bf16x6" and hip.GEMM_H2) else MODE_PEAK[hip.GEMM_MODE]
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * args.batch * args.steps / elapsed
        out = {
            "metric": ("images/sec, train step (fwd + DiceCE + bwd + AdamW on prompts), " +
                       ("CRIS (CLIP-RN50) + CoCoOp, 416x416" if cris else "CLIPSeg ViT-B/16 + MaPLe depth 9, 352x352" if maple
                        else "CLIPSeg ViT-B/16 + VPT-10 shallow, 352x352")) if not (vit640 or denseclip) else
                      ("images/sec, DenseCLIP ViT-B/16 FPN 640x640 up to the mmseg neck / head: frozen backbone + 4 FPN taps, context text encoder, context decoder, "
                       "score map; backward of a fixed cotangent into contexts / gamma; AdamW" if denseclip else
                       "images/sec, forward + data gradient of the 12 ViT-B/16 encoder layers at 640x640 (the encoder leg of DenseCLIP ViT-B FPN; no head, no optimizer)"),
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("f32 (fp32-equivalent: operands split into 2 fp16 pieces with exact power-of-two scales [3x3 convs with C_in % 32 == 0 and C_out >= 128, "
                       "Linears / 1x1 convs with N, K >= 256] or 3 bf16 pieces [the rest]; piece products exact on the 16-bit MFMA, fp32 accumulate)" if cris else
                       "f32 (fp32-equivalent: operands split into 2 fp16 pieces with exact power-of-two scales [the 8 GEMMs and the attention of a vision layer, the decoder's "
                       "feed-forward block] or 3 bf16 pieces [text tower, decoder attention and small GEMMs]; piece products exact on the 16-bit MFMA, fp32 accumulate)")
                      if hip.GEMM_MODE == "bf16x6" and hip.GEMM_H2 else MODE_DTYPE[hip.GEMM_MODE]),
            "gemm_mode": hip.GEMM_MODE, "step_graph": bool(args.graph and module is not None),
            "data": "synthetic", "per_gpu": round(value / world, 2),
            "config": {"workload": ("CRIS (CLIP-RN50 + cross-attn decoder) + CoCoOp meta-net, 416x416, bs=32/GPU (BASELINE configs[2])" if cris else
                                    "CLIPSeg ViT-B/16 + MaPLe (coupled V+L prompts, depth=9), 352x352, bs=32/GPU (BASELINE configs[3])" if maple else
                                    "ViT-B/16 encoder leg of DenseCLIP ViT-B FPN 640x640, bs=16/GPU (BASELINE configs[4]: encoder layers only)" if vit640 else
                                    "DenseCLIP ViT-B/16 FPN 640x640, 20 classes, bs=16/GPU (BASELINE configs[4]) up to after_extract_feat; the mmseg FPN neck / FPNHead and "
                                    "their loss are not built (no source in the reference tree)" if denseclip else
                                    "CLIPSeg ViT-B/16 + VPT-shallow (10 visual prompts), 352x352, bs=32/GPU (BASELINE configs[1])"),
                       "global_batch": world * args.batch, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "weights": "seeded random init (RN50 CRIS geometry)" if cris else "seeded random init (rd64 geometry)",
                       "use_new_last_layer": cris or maple},
            "step_tflops": round(value * gflop_per_image / 1e3, 2),
            # whole-step algorithmic TFLOP/s per GPU against the ceiling of the arithmetic the step's FLOPs actually run on: two fp16
            # pieces (3 MFMAs per fp32 product, 833 TFLOP/s) for the layer GEMMs, the attention and the large convs when GEMM_H2 is on
            "step_frac_of_ceiling": round(value / world * gflop_per_image / 1e3 / step_ceiling, 4), "step_ceiling_tflops": round(step_ceiling, 1),
            "loss": round(float(loss.item()), 6), "train_dice": round(metrics["train_dice"], 6) if metrics else None, "train_iou": round(metrics["train_iou"], 6) if metrics else None,
            "roofline": roofline,
        }
        if args.cond_cache and not cris and not maple:
            out["cond_cache"] = True
            out["config"]["workload"] += " -- WITH the conditional-embedding cache (text tower skipped after the first step: not the headline)"
        if not args.no_cpu_baseline and world == 1 and args.workload == "vpt":
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps32, args.cpu_steps4)
        print(json.dumps(out), flush=True)
    if grouped:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
