"""Predict / offline-eval tail: PNG artefacts and the per-image metric CSV."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tunevlseg_amd import predict as P


def test_binary_scores_follow_monai_ignore_empty_false():
    a = np.zeros((4, 4), bool)
    b = np.zeros((4, 4), bool)
    assert P.binary_scores(a, b) == (1.0, 1.0)          # both empty
    a[0, 0] = True
    assert P.binary_scores(a, b) == (0.0, 0.0)          # empty ground truth, non-empty prediction
    b[0, :2] = True
    iou, dice = P.binary_scores(a, b)
    assert iou == pytest.approx(0.5) and dice == pytest.approx(2 / 3)


def test_eval_metrics_csv(tmp_path):
    from PIL import Image

    seg, gt = tmp_path / "seg", tmp_path / "gt"
    seg.mkdir(), gt.mkdir()
    g = np.zeros((8, 8), np.uint8)
    g[:4] = 255
    p = np.zeros((8, 8), np.uint8)
    p[:2] = 200
    p[2:4] = 100  # below the threshold of 127
    for name in ("b.png", "a.png"):
        Image.fromarray(g).save(gt / name)
        Image.fromarray(np.repeat(p[:, :, None], 3, 2)).save(seg / name)  # RGB with equal channels, as save_image writes
    rows = P.eval_metrics(seg, gt, tmp_path / "m.csv", threshold=127)
    assert [r["filename"].rsplit("/", 1)[-1] for r in rows] == ["a.png", "b.png"]
    assert rows[0]["iou"] == pytest.approx(50.0) and rows[0]["dice"] == pytest.approx(100 * 2 * 16 / 48)
    ones_dice = 100 * 2 * 32 / (32 + 64)
    assert rows[0]["ones_dice_diff"] == pytest.approx(rows[0]["dice"] - ones_dice)
    text = (tmp_path / "m.csv").read_text().splitlines()
    assert text[0] == "filename,iou,dice,ones_dice_diff" and text[1].endswith("50.0000,66.6667,0.0000")


@pytest.mark.gpu
@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(352, 352, 500, 574), (64, 64, 64, 64), (416, 416, 300, 200), (96, 96, 97, 1000)])
def test_bicubic_resize_u8_matches_torch(Hi, Wi, Ho, Wo):
    from tunevlseg_amd import hip

    g = torch.Generator().manual_seed(3)
    pred = torch.sigmoid(4 * torch.randn(Hi, Wi, generator=g))
    ref = F.interpolate(pred[None, None], size=(Ho, Wo), mode="bicubic", align_corners=False, antialias=False)[0, 0]
    ref_u8 = ref.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)  # torchvision.utils.save_image
    out = hip.bicubic_resize_u8(pred.cuda().contiguous(), Ho, Wo).cpu()
    diff = (out.int() - ref_u8.int()).abs()
    # the float result can differ in its last bit between the two bicubic implementations: at most one grey level, rarely
    assert diff.max().item() <= 1 and (diff > 0).float().mean().item() < 2e-3


@pytest.mark.gpu
def test_save_predictions_writes_pngs_at_mask_shape(tmp_path):
    from functools import partial

    from PIL import Image

    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import VPTContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    net = nets.VPTCLIPSeg(context_learner=partial(VPTContextLearner, prompt_depth=1, num_context=2),
                          model_cfg={"pretrained_model_name_or_path": "random:tiny:seed=3"})
    module = ImageTextMaskModule(net, DiceCELoss(sigmoid=True), optimizer=partial(FusedAdamW, lr=1e-3), scheduler=None).cuda()
    g = torch.Generator().manual_seed(0)
    batch = {"image": torch.randn(2, 3, 64, 64, generator=g).cuda(), "input_ids": torch.tensor([[62, 5, 63, 1], [62, 7, 9, 63]]).cuda(),
             "attention_mask": torch.tensor([[1, 1, 1, 0], [1, 1, 1, 1]]).cuda(), "mask_name": ["a.png", "sub/b.png"],
             "mask_shape": [np.array([50, 70]), torch.tensor([64, 64])]}
    out = tmp_path / "masks"
    assert P.save_predictions(module, [batch], out) == 2
    a, b = np.asarray(Image.open(out / "a.png")), np.asarray(Image.open(out / "sub" / "b.png"))
    assert a.shape == (50, 70, 3) and b.shape == (64, 64, 3) and (a[..., 0] == a[..., 1]).all()
    probs = module.predict_step(batch)["preds"][1, 0].cpu()
    assert np.abs(b[..., 0].astype(int) - (probs * 255 + 0.5).clamp(0, 255).to(torch.uint8).numpy().astype(int)).max() <= 1
    assert P.save_predictions(module, [batch], out) == 0  # exists and overwrite_outputs is false: nothing is written
    assert P.save_predictions(module, [batch], out, overwrite_outputs=True) == 2
