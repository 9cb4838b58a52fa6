"""The first vision layer under the visual prompts computes its input gradient for the prompt rows only (ops.EncoderLayerTp3Fn with
``grad_rows``): patches, CLS and position rows are frozen in the reference too (``src/models/components/vpt_clipseg.py``: the learner's
context vectors are the only trainable input of layer 1), so their gradient rows have no consumer.  Dead-work elimination, not an
approximation: the prompt rows must equal the full backward's rows to fp32 rounding, every other row is zero."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def layer():
    from tunevlseg_amd import hip
    from tunevlseg_amd.backbone import CLIPSegBackbone
    from tunevlseg_amd.ops import AttnSpec

    hip.load()
    bb = CLIPSegBackbone.from_spec("random:rd64:seed=5").requires_grad_(False).cuda()
    return hip, bb.prepared()["vision_layers"][0], AttnSpec(heads=12, act=hip.ACT_QUICK_GELU, eps=1e-5)


def run(lw, spec, x, dout, rows):
    from tunevlseg_amd import ops

    h = x.cuda().requires_grad_(True)
    y = ops.encoder_layer(h, lw, spec, rows)
    (g,) = torch.autograd.grad(y, h, dout.cuda())
    return y.detach(), g.detach()


@pytest.mark.parametrize("B,T,n", [(32, 495, 10), (32, 489, 4), (9, 600, 16)])
def test_prompt_rows_gradient_equals_the_full_backward(layer, B, T, n):
    hip, lw, spec = layer
    assert hip.GRAD_ROWS and hip.tp3_path_ok(B * T, 768, 3072, 64, False, None)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, 768, generator=g)
    dout = torch.randn(B, T, 768, generator=g) * torch.logspace(-5, -2, B * T).view(B, T, 1)
    y0, g0 = run(lw, spec, x, dout, None)
    for _ in range(2):   # twice: rows the restricted kernels leave unwritten must never be read
        y1, g1 = run(lw, spec, x, dout, (T - n, n))
        assert torch.equal(y1, y0)
        assert not g1[:, : T - n].any()
        ref = g0[:, T - n:]
        err = (g1[:, T - n:] - ref).abs().amax(-1) / ref.abs().amax(-1)
        assert err.max().item() < 2e-5, err.max().item()


def test_rows_that_straddle_a_block_fall_back_to_the_full_backward(layer):
    hip, lw, spec = layer
    B, T, n = 32, 520, 10   # rows 510 .. 519 touch the 128-row blocks 3 and 4
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, 768, generator=g)
    dout = torch.randn(B, T, 768, generator=g) * 1e-3
    _, g0 = run(lw, spec, x, dout, None)
    _, g1 = run(lw, spec, x, dout, (T - n, n))
    assert torch.equal(g0, g1)


def test_a_hook_that_keeps_the_gradient_sees_it_unmodified():
    """RowsOverwriteFn.backward cuts the overwritten rows of the incoming gradient IN PLACE only when it is the fresh tensor the layer above's
    backward returned and nobody else holds it; a tensor hook that keeps the gradient of the overwritten activations must see the true
    gradient (non-zero prompt rows), and the parameter gradients must not depend on the hook being there."""
    from functools import partial

    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import VPTContextLearner

    def run(with_hook):
        torch.manual_seed(0)
        net = nets.VPTCLIPSeg(context_learner=partial(VPTContextLearner, prompt_depth=3, num_context=10),
                              model_cfg={"pretrained_model_name_or_path": "random:rd64:seed=3"}, use_new_last_layer=False).cuda()
        kept = []
        if with_hook:
            from tunevlseg_amd import ops

            orig = ops.RowsOverwriteFn.apply

            def hooked(h, src, row0):
                out = orig(h, src, row0)
                out.register_hook(lambda g: kept.append((g, g.detach().clone())) and None)
                return out

            ops.RowsOverwriteFn.apply = hooked
        try:
            g = torch.Generator().manual_seed(1)
            pix = torch.randn(4, 3, 352, 352, generator=g).cuda()
            ids = torch.tensor([[49406, 320, 1125, 49407]] * 4).cuda()
            logits = net(text_input={"input_ids": ids, "attention_mask": torch.ones_like(ids)}, image_input=pix)
            (logits * torch.randn(logits.shape, generator=g).cuda()).sum().backward()
        finally:
            if with_hook:
                ops.RowsOverwriteFn.apply = orig
        return net.context_learner.context_vectors.grad.detach().clone(), kept

    g_plain, _ = run(False)
    g_hook, kept = run(True)
    assert kept, "the hook never fired"
    for held, snapshot in kept:
        assert torch.equal(held, snapshot)            # what the hook kept was not edited behind its back
        assert held[:, -10:].abs().max().item() > 0   # ... and still holds the prompt rows' gradient
    assert (g_plain - g_hook).abs().max().item() <= 1e-6 * g_plain.abs().max().item()
