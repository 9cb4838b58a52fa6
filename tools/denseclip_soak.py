#!/usr/bin/env python
"""DenseCLIP forward / backward repeated on ONE input without parameter updates and without a host sync inside the loop: every step must leave the same
checksums (four FPN maps -- produced on the second stream beside the context decoder --, score map, text embeddings, the two gradients), bit for bit.
A cross-stream reuse of a buffer that is still being read, or a missing stream dependency, shows up as a step that differs.

    python tools/denseclip_soak.py [--steps 60] [--batch 8] [--preset vitb16_640|tiny]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--preset", default="vitb16_640")
    args = ap.parse_args()
    from tunevlseg_amd import hip
    from tunevlseg_amd.denseclip_backbone import DenseCLIPWeights
    from tunevlseg_amd.denseclip_config import DenseCLIPConfig
    from tunevlseg_amd.nets import DenseCLIP

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip.load()
    cfg = DenseCLIPConfig.vitb16_640(num_classes=20) if args.preset == "vitb16_640" else DenseCLIPConfig.tiny()
    g = torch.Generator().manual_seed(7)
    sot, eot = cfg.vocab_size - 2, cfg.vocab_size - 1
    texts = torch.zeros(cfg.num_classes, cfg.context_length, dtype=torch.long)
    for k in range(cfg.num_classes):
        row = [sot, *torch.randint(1, min(cfg.vocab_size - 2, 40000), (1 + k % 3,), generator=g).tolist(), eot]
        texts[k, : len(row)] = torch.tensor(row)
    torch.manual_seed(3)
    net = DenseCLIP(pretrained=DenseCLIPWeights(cfg, None, seed=0), texts=texts).to(dev)
    with torch.no_grad():
        net.gamma.fill_(0.3)
    H = cfg.input_resolution
    img = torch.randn(args.batch, 3, H, H, generator=g).to(dev)
    G = H // cfg.patch_size
    gs = torch.randn(args.batch, cfg.num_classes, G, G, generator=g).to(dev)
    gt = (torch.randn(args.batch, cfg.num_classes, cfg.embed_dim, generator=g) * 0.1).to(dev)
    sums = torch.zeros(args.steps, 8, device=dev, dtype=torch.float64)
    for i in range(args.steps):
        net.contexts.grad = None
        net.gamma.grad = None
        te, maps, score = net(img)
        loss = (score * gs).sum() + (te * gt).sum()
        loss.backward()
        vals = [m.detach().double().abs().sum() for m in maps] + [score.detach().double().abs().sum(), te.detach().double().abs().sum(),
                                                                    net.contexts.grad.double().abs().sum(), net.gamma.grad.double().abs().sum()]
        sums[i] = torch.stack(vals)
        del te, maps, score, loss
    torch.cuda.synchronize()
    assert torch.isfinite(sums).all(), "non-finite checksum"
    s = sums.cpu()
    bad = [i for i in range(args.steps) if not torch.equal(s[i], s[0])]
    names = ["fpn1", "fpn2", "fpn3|score", "fpn4", "score_map", "text_embeddings", "d contexts", "d gamma"]
    print(f"denseclip soak: {args.steps} steps, batch {args.batch}, checksums {dict(zip(names, [f'{v:.6e}' for v in s[0].tolist()]))}")
    if bad:
        i = bad[0]
        diff = [n for n, a, b in zip(names, s[i].tolist(), s[0].tolist()) if a != b]
        print(f"FIRST DIFFERING STEP {i} of {len(bad)}: {diff}")
        sys.exit(1)
    print("every step bit-identical to step 0; all finite")


if __name__ == "__main__":
    main()
