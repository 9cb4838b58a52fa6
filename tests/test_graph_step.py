"""Forward + backward of a train step replayed as a hipGraph (tunevlseg_amd/graph.py) against the eager step: same parameters after every
update, on batches whose contents change from step to step; the trainer's opt-in switch; the refusal to capture on the default stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def batches(n, seed=5, B=4, size=64):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        ids = torch.tensor([[62, 5, 9, 63, 1, 1], [62, 7, 11, 13, 63, 1], [62, 8, 63, 1, 1, 1], [62, 3, 4, 6, 9, 63]])[:B]
        out.append({"image": torch.randn(B, 3, size, size, generator=g).cuda(), "input_ids": ids.cuda(), "attention_mask": (ids != 1).long().cuda(),
                    "mask": (torch.rand(B, 1, size, size, generator=g) > 0.7).float().cuda()})
    return out


def trainable(module):
    return {k: p.detach().clone() for k, p in module.named_parameters() if p.requires_grad}


@pytest.mark.parametrize("depth", [1, 3])
def test_replayed_steps_equal_eager_steps(depth):
    from tests.test_train_gpu import tiny_module
    from tunevlseg_amd.graph import GraphedStep

    data = batches(6)
    with torch.cuda.stream(torch.cuda.Stream()):   # the capture stream is current while the parameters and the optimiser come to exist
        ref = tiny_module(depth=depth)
        ref.setup("fit")
        ropt = ref.configure_optimizers()["optimizer"]
        want = []
        for b in data:
            ropt.zero_grad()
            loss = ref.training_step(b, 0)
            loss.backward()
            ropt.step()
            want.append((loss.item(), trainable(ref)))
        module = tiny_module(depth=depth)
        module.setup("fit")
        opt = module.configure_optimizers()["optimizer"]
        stepper = GraphedStep(module, opt)
        for i, b in enumerate(data):
            loss = stepper(b)
            opt.step()
            assert abs(loss.item() - want[i][0]) <= 1e-6, (i, loss.item(), want[i][0])
            for k, p in trainable(module).items():
                assert (p - want[i][1][k]).abs().max().item() <= 1e-6 * max(1.0, want[i][1][k].abs().max().item()), (i, k)
        torch.cuda.synchronize()
    assert stepper.replays == len(data) - 1   # batch 0 eager (fills the caches), batch 1 captured and replayed, the rest replayed
    m, r = module.epoch_metrics("train"), ref.epoch_metrics("train")
    assert m == pytest.approx(r, abs=1e-6)   # the device-side metric sums are part of the captured step


def test_capture_on_the_default_stream_is_refused():
    from tests.test_train_gpu import tiny_module
    from tunevlseg_amd.graph import GraphedStep

    assert torch.cuda.current_stream() == torch.cuda.default_stream()
    module = tiny_module(depth=1)
    module.setup("fit")
    opt = module.configure_optimizers()["optimizer"]
    stepper = GraphedStep(module, opt)
    b = batches(1)[0]
    stepper(b)   # first sight of the shape: eager
    with pytest.raises(RuntimeError, match="use_private_stream"):
        stepper(b)


def test_trainer_graph_step_matches_the_eager_trainer(tmp_path):
    from tests.test_train_gpu import tiny_module
    from tunevlseg_amd.graph import use_private_stream
    from tunevlseg_amd.trainer import SyntheticImageTextMaskLoader, Trainer

    default = torch.cuda.default_stream()
    try:
        s = use_private_stream()
        assert s != default and use_private_stream() == s   # idempotent
        finals = []
        for graph in (False, True):
            module = tiny_module(depth=3, n=4, new_last=True, lr=5e-3)
            tr = SyntheticImageTextMaskLoader(4, 4, 64, "cuda", seed=1, vocab=64, bos=62, eos=63, pad=1, max_len=6)
            trainer = Trainer(max_epochs=3, min_epochs=1, default_root_dir=str(tmp_path / str(graph)), log_fn=lambda *_: None, graph_step=graph)
            finals.append((trainer.fit(module, tr, None), trainable(module)))
        assert finals[1][0]["train_loss"] == pytest.approx(finals[0][0]["train_loss"], abs=1e-5)
        for k, p in finals[0][1].items():
            assert (p - finals[1][1][k]).abs().max().item() <= 1e-5 * max(1.0, p.abs().max().item()), k
        with pytest.raises(ValueError):
            Trainer(graph_step=True, accumulate_grad_batches=2)
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(default)


def test_host_entries_of_the_batch_stay_out_of_the_capture():
    """The reference's collate leaves ``mask_shape`` on the CPU and puts it FIRST in the dict (RaggedCollator -> DeviceTransform): the capture
    takes its device from the device tensors and neither keys nor copies host entries."""
    from tests.test_train_gpu import tiny_module
    from tunevlseg_amd.graph import GraphedStep

    data = batches(4)
    with torch.cuda.stream(torch.cuda.Stream()):
        module = tiny_module(depth=1)
        module.setup("fit")
        opt = module.configure_optimizers()["optimizer"]
        stepper = GraphedStep(module, opt)
        for i, b in enumerate(data):
            b = {"mask_shape": torch.tensor([[64 + i, 64]] * 4), "name": [f"img{i}"] * 4, **b}   # host tensor first, a non-tensor entry, values change per batch
            loss = stepper(b)
            opt.step()
            assert torch.isfinite(loss).item()
        torch.cuda.synchronize()
    assert stepper.replays == len(data) - 1 and len(stepper._graphs) == 1


def shared_attn_module(dropout: float):
    from functools import partial

    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import SharedAttnLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    torch.manual_seed(0)
    net = nets.SharedAttnCLIPSeg(
        context_learner=partial(SharedAttnLearner, prompt_depth=2, num_context=3, vector_std=0.02, use_unified_projection=False,
                                unified_projector=partial(torch.nn.TransformerEncoderLayer, nhead=4, dim_feedforward=48, dropout=dropout, norm_first=True)),
        model_cfg={"pretrained_model_name_or_path": "random:tiny:seed=11"}, use_new_last_layer=False)
    return ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), optimizer=partial(FusedAdamW, lr=2e-3), scheduler=None).cuda()


def test_a_step_with_train_mode_dropout_is_not_replayed():
    """ops.DropoutFn takes its (seed, call index) from the host: frozen into a graph, every replay would repeat the capture's mask.  Such a shape
    stays eager -- and draws exactly the masks a stepper-free loop draws (the capture attempt puts the call counter back)."""
    from tunevlseg_amd import ops
    from tunevlseg_amd.graph import GraphedStep

    data = batches(4)
    with torch.cuda.stream(torch.cuda.Stream()):
        runs = []
        for graphed in (False, True):
            torch.manual_seed(7)
            ops._dropout_calls = 0
            module = shared_attn_module(dropout=0.25)
            module.setup("fit")
            module.net.context_learner.train()
            opt = module.configure_optimizers()["optimizer"]
            stepper = GraphedStep(module, opt) if graphed else None
            losses = []
            for b in data:
                if stepper is None:
                    opt.zero_grad()
                    loss = module.training_step(b, 0)
                    loss.backward()
                else:
                    loss = stepper(b)
                opt.step()
                losses.append(loss.item())
            runs.append((losses, trainable(module)))
        torch.cuda.synchronize()
    assert stepper.replays == 0 and len(stepper._eager_only) == 1 and not stepper._graphs
    assert runs[0][0] == pytest.approx(runs[1][0], abs=1e-6)
    for k, p in runs[0][1].items():
        assert (p - runs[1][1][k]).abs().max().item() <= 1e-6 * max(1.0, p.abs().max().item()), k
    assert len({round(v, 5) for v in runs[0][0]}) == len(data)   # the masks (and the data) differ from step to step
    # without dropout the same learner IS captured and replayed
    with torch.cuda.stream(torch.cuda.Stream()):
        module = shared_attn_module(dropout=0.0)
        module.setup("fit")
        module.net.context_learner.train()
        opt = module.configure_optimizers()["optimizer"]
        stepper = GraphedStep(module, opt)
        for b in data:
            stepper(b)
            opt.step()
        torch.cuda.synchronize()
    assert stepper.replays == len(data) - 1
