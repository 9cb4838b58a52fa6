"""Child process of tests/test_train_gpu.py::test_two_rank_step_on_hip_path_equals_global_batch_step: one rank of a
data-parallel job on the HIP path (or the single-process reference when WORLD_SIZE=1).  Writes its parameters after two steps."""
import os
import sys
from functools import partial
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main(out_dir: str) -> None:
    from tunevlseg_amd import dist as tdist
    from tunevlseg_amd import nets
    from tunevlseg_amd.nets.context_learner import MapleContextLearner
    from tunevlseg_amd.task import DiceCELoss, FusedAdamW, ImageTextMaskModule

    rank, _, world = tdist.init_distributed("cuda")
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    net = nets.MapleCLIPSeg(context_learner=partial(MapleContextLearner, prompt_depth=3, num_context=2, intermediate_dim=8, use_proj_norm=True),
                            model_cfg={"pretrained_model_name_or_path": "random:tiny:seed=11"}, use_new_last_layer=True)
    module = ImageTextMaskModule(net, DiceCELoss(sigmoid=True, lambda_dice=1, lambda_ce=0.2), optimizer=partial(FusedAdamW, lr=2e-3),
                                 scheduler=None, weight_decay=0.01).cuda()
    module.setup("fit")
    os.environ["TVL_DDP_BUCKET_BYTES"] = "1024"  # several buckets even at this width
    opt = module.configure_optimizers()["optimizer"]
    assert len(opt.exchange.buckets) > 1
    g = torch.Generator().manual_seed(3)
    B = 4
    pix = torch.randn(B, 3, 64, 64, generator=g)
    ids = torch.tensor([[62, 5, 9, 63, 1, 1], [62, 7, 11, 13, 63, 1], [62, 8, 63, 1, 1, 1], [62, 9, 9, 9, 9, 63]])
    am = (ids != 1).long()
    mask = (torch.rand(B, 1, 64, 64, generator=g) > 0.7).float()
    per = tdist.per_device_batch_size(B, world)
    sl = slice(rank * per, (rank + 1) * per)
    batch = {"image": pix[sl].cuda(), "input_ids": ids[sl].cuda(), "attention_mask": am[sl].cuda(), "mask": mask[sl].cuda()}
    for _ in range(2):
        opt.zero_grad()
        module.training_step(batch).backward()
        opt.step()
    torch.cuda.synchronize()
    torch.save({"params": {k: p.detach().cpu() for k, p in module.named_parameters() if p.requires_grad},
                "launched_in_backward": opt.exchange.launched_in_backward,
                "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None},
               Path(out_dir) / f"world{world}_rank{rank}{os.environ.get('TVL_WORKER_TAG', '')}.pt")
    if torch.distributed.is_initialized():
        tdist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
