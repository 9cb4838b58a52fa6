#!/usr/bin/env python
"""Section timing of the CRIS train step (HIP events around the stages of COOPCRIS.forward + backward)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    module, opt = bench.build_cris_module(dev)
    net = module.net
    batch = bench.make_batch(32, 416, 100, dev, pad_id=0)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    orig_blocks = None

    def run(timed):
        marks.clear()
        opt.zero_grad()
        mark("start")
        ids, am, img = batch["input_ids"], batch["attention_mask"], batch["image"]
        pad = net.get_pad_mask(ids, am)
        vis = net.encode_image(img)
        mark("image tower")
        x5, H5, W5 = vis[2]
        from tunevlseg_amd import hip
        with torch.no_grad():
            feats = hip.avgpool_fwd(x5, 32, H5, W5, H5)
        words, state = net.encode_text(ids, feats, key_padding_mask=pad)
        mark("text tower")
        fq, H, W = net.neck_forward(vis, state)
        mark("neck")
        fq = net.decoder_forward(fq, H, W, words, pad)
        mark("decoder")
        pred = net.proj_forward(fq, H, W, state)
        mark("projector")
        from tunevlseg_amd import cris_ops as C, ops
        logits = C.BicubicFn.apply(pred, 416, 416)
        w1 = net.additive_decoder_layer[0].weight
        z = ops.linear(fq, w1.view(w1.shape[0], -1))
        conv = net.additive_decoder_layer[2]
        extra = C.UpconvFn.apply(z, conv.weight, conv.bias, 32, H, 16)
        logits = C.MixFn.apply(logits, extra, net.residual_ratio).view(32, 1, 416, 416)
        loss = module.loss_fn(logits, batch["mask"])
        mark("head + loss")
        loss.backward()
        mark("backward")
        opt.step()
        mark("adamw")
        torch.cuda.synchronize()
        return [(n, marks[i - 1][1].elapsed_time(e)) for i, (n, e) in enumerate(marks) if i]

    for _ in range(2):
        run(False)
    acc = {}
    for _ in range(3):
        for n, ms in run(True):
            acc[n] = acc.get(n, 0.0) + ms / 3
    tot = sum(acc.values())
    for n, ms in acc.items():
        print(f"{n:14s} {ms:8.2f} ms  {100 * ms / tot:5.1f}%")
    print(f"{'total':14s} {tot:8.2f} ms")


if __name__ == "__main__":
    main()
