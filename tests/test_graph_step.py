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
