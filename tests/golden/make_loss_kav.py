#!/usr/bin/env python
"""Known-answer vectors for the loss / metric definitions.

monai and torchmetrics are not installed anywhere this build runs and are not vendored by the reference, so their semantics
cannot be pinned by running them.  These vectors are derived BY HAND from the published definitions, with plain ``math`` in
double precision (no import of oracle/ or tunevlseg_amd/), and both the CPU oracle and the HIP kernel are held to them:

  L1  monai.losses.DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2)      (reference configs/model/vpt_clipseg.yaml:21-25)
        p = sigmoid(x);  per sample b (one channel):  dice_b = 1 - (2*sum(p*t) + 1e-5) / (sum(p) + sum(t) + 1e-5)
        (DiceLoss defaults smooth_nr = smooth_dr = 1e-5, include_background, no squared_pred, reduction "mean" over (b, c));
        one output channel -> the CE term is BCEWithLogitsLoss(mean over every element): -(t*ln p + (1-t)*ln(1-p));
        loss = 1 * mean_b(dice_b) + 0.2 * bce.        The mask enters the LOSS as the float it is (values in [0, 1]).
  L2  torchmetrics.Dice(threshold=0.5, zero_division=1, average="samples"), JaccardIndex(task="binary", threshold=0.5,
      zero_division=1)                                                          (reference image_text_mask_module.py:284-302)
        label = p > 0.5;  target = mask.long()  (TRUNCATION: only mask == 1.0 counts, image_text_mask_module.py:87-107);
        Dice  = mean over samples of 2TP / (2TP + FP + FN), a sample with 2TP+FP+FN = 0 scores zero_division = 1;
        IoU   = sum_b TP / sum_b (TP + FP + FN) over everything seen, 1 if the denominator is 0.

Still unpinned by package (stated in DESIGN.md): whether torchmetrics' legacy Dice thresholds with ``>=`` (we use ``>`` for both
metrics; the two differ only where sigmoid(x) == 0.5 exactly, i.e. |x| < 6e-8) -- the vector "knife_edge" records that case and
is NOT asserted.

Run:  python tests/golden/make_loss_kav.py   ->  tests/golden/loss_metric_kav.json
"""
import json
import math
from pathlib import Path


def sig(x):
    return 1.0 / (1.0 + math.exp(-x))


def case(name, logits, mask, derivation, asserted=True):
    """logits / mask: [B][N] python lists.  Everything below is the definition, term by term."""
    B = len(logits)
    dice_terms, bce_sum, n = [], 0.0, 0
    counts, dice_samples = [], []
    for xs, ts in zip(logits, mask):
        ps = [sig(x) for x in xs]
        inter = sum(p * t for p, t in zip(ps, ts))
        dice_terms.append(1.0 - (2.0 * inter + 1e-5) / (sum(ps) + sum(ts) + 1e-5))
        for x, p, t in zip(xs, ps, ts):
            # -(t ln p + (1 - t) ln(1 - p)), written with log1p for accuracy: ln(1 + e^-|x|) + max(x, 0) - x t
            bce_sum += math.log1p(math.exp(-abs(x))) + max(x, 0.0) - x * t
            n += 1
        lab = [p > 0.5 for p in ps]
        tgt = [int(t) == 1 for t in ts]  # .long() truncates: 0.5 -> 0, 1.0 -> 1
        tp = sum(a and b for a, b in zip(lab, tgt))
        fp = sum(a and not b for a, b in zip(lab, tgt))
        fn = sum((not a) and b for a, b in zip(lab, tgt))
        tn = sum((not a) and (not b) for a, b in zip(lab, tgt))
        counts.append([tp, fp, fn, tn])
        den = 2 * tp + fp + fn
        dice_samples.append(2 * tp / den if den else 1.0)
    dice = sum(dice_terms) / B
    bce = bce_sum / n
    TP, FP, FN = (sum(c[i] for c in counts) for i in range(3))
    iou = TP / (TP + FP + FN) if TP + FP + FN else 1.0
    return {"name": name, "logits": logits, "mask": mask, "derivation": derivation, "asserted": asserted,
            "expect": {"dice_term_per_sample": dice_terms, "dice": dice, "bce": bce, "loss": 1.0 * dice + 0.2 * bce,
                       "counts_tp_fp_fn_tn": counts, "metric_dice_samples": sum(dice_samples) / B, "metric_iou": iou}}


L3 = math.log(3.0)  # sigmoid(ln 3) = 3/4, sigmoid(-ln 3) = 1/4
cases = [
    case("single_pixel", [[0.2]], [[1.0]],
         "p = sigmoid(0.2) = 0.549834; dice = 1 - (2p + 1e-5)/(p + 1 + 1e-5); bce = -ln p; label = (p > 0.5) = 1, target 1: TP = 1 -> Dice 1, IoU 1"),
    case("all_ones_mask", [[L3] * 4], [[1.0] * 4],
         "p = 3/4 on 4 pixels: inter = 3, sum p = 3, sum t = 4 -> dice = 1 - (6 + 1e-5)/(7 + 1e-5); bce = -ln(3/4); TP = 4"),
    case("empty_mask_negative_logits", [[-L3] * 4], [[0.0] * 4],
         "p = 1/4: inter = 0, sum p = 1 -> dice = 1 - 1e-5/(1 + 1e-5) (the smooth terms are all that is left); bce = -ln(3/4); "
         "no predicted and no true pixel: 2TP+FP+FN = 0 -> sample Dice = zero_division = 1, IoU denominator 0 -> 1"),
    case("two_samples_average", [[L3, L3, -L3, -L3], [-L3, -L3, -L3, L3]], [[1.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
         "sample 0: labels 1100 vs target 1010: TP 1 FP 1 FN 1 -> 2/4; sample 1: labels 0001 vs 0001: TP 1 -> 1; Dice(samples) = (0.5 + 1)/2 "
         "= 0.75 (NOT the pooled 2*2/(2*2+1+1) = 0.667); IoU pooled = 2/(2+1+1) = 0.5; the loss' dice term is likewise a mean over samples"),
    case("fractional_mask_truncation", [[L3, L3, -L3, L3]], [[0.5, 1.0, 0.5, 0.0]],
         "mask values 0.5 (a grey level 127/255 after /255) enter the LOSS as 0.5 (inter = 0.75*0.5 + 0.75 + 0.25*0.5) but the METRICS as "
         "mask.long() = 0: target = 0100, labels = 1101 -> TP 1, FP 2, FN 0: Dice = 2/4, IoU = 1/3"),
    case("mixed_batch_with_empty_sample", [[2.0, -1.0, 0.5], [-2.0, -3.0, -0.1]], [[1.0, 1.0, 0.0], [0.0, 0.0, 0.0]],
         "sample 1 is empty on both sides -> its Dice is zero_division = 1 and it adds nothing to the pooled IoU counts"),
    case("knife_edge", [[0.0, 0.0]], [[1.0, 0.0]],
         "sigmoid(0) = 0.5 exactly: '>' gives label 0 (TP 0, FN 1), '>=' would give label 1 (TP 1, FP 1). Recorded, not asserted: "
         "which of the two torchmetrics' Dice uses cannot be confirmed without the package", asserted=False),
]
out = Path(__file__).resolve().parent / "loss_metric_kav.json"
out.write_text(json.dumps({"source": "hand-derived from the published monai / torchmetrics definitions (see make_loss_kav.py)",
                           "lambda_dice": 1.0, "lambda_ce": 0.2, "threshold": 0.5, "cases": cases}, indent=1))
print(f"wrote {out} ({len(cases)} cases)")
