"""Hydra-shaped composition + instantiate (CPU): the reference's model YAML schema resolves onto this package."""
from functools import partial
from pathlib import Path

import pytest

from tunevlseg_amd import config_loader as CL

REF_CONFIGS = Path("/root/reference/configs")


def write(p: Path, text: str):
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_text(text)


@pytest.fixture()
def tree(tmp_path):
    """A self-authored config tree with the reference's structure (defaults lists, @package _global_, overrides)."""
    write(tmp_path / "train.yaml", """# @package _global_
defaults:
  - _self_
  - model: base
  - trainer: default
  - experiment: null
  - optional local: default
task_name: train
seed: null
""")
    write(tmp_path / "trainer" / "default.yaml", "max_epochs: 10\nmin_epochs: 1\naccumulate_grad_batches: 1\n")
    write(tmp_path / "model" / "base.yaml", "weight_decay: 0.0\noptimizer:\n  lr: 1.0e-3\n")
    write(tmp_path / "model" / "vpt_clipseg.yaml", """_target_: src.models.image_text_mask_module.ImageTextMaskModule
net:
  _target_: src.models.core_models.coop.VPTCLIPSeg
  model_cfg:
    pretrained_model_name_or_path: ${model_pretrained_path}
    freeze_encoder: false
    freeze_decoder: false
  context_learner:
    _target_: src.models.core_models.coop.context_learner.VPTContextLearner
    _partial_: true
    prompt_depth: 1
    num_context: 4
    vector_std: 0.02
  freeze_all: true
  no_freeze_last_layer: false
  use_new_last_layer: true
  new_last_layer_kernel_size: 5
  residual_ratio: 0.5
loss_fn:
  _target_: monai.losses.DiceCELoss
  sigmoid: true
  lambda_dice: 1
  lambda_ce: 0.2
weight_decay: 0.0
optimizer:
  _target_: torch.optim.AdamW
  _partial_: true
  lr: 2.0e-4
scheduler:
  _target_: torch.optim.lr_scheduler.ReduceLROnPlateau
  _partial_: true
  mode: min
  factor: 0.2
  patience: 5
compile: false
task: binary
threshold: 0.5
""")
    write(tmp_path / "experiment" / "coop" / "clipseg.yaml", """# @package _global_
defaults:
  - override /model: vpt_clipseg
  - override /trainer: default
tags: ["coop"]
seed: 12345
trainer:
  min_epochs: 10
  max_epochs: 20
model:
  net:
    use_new_last_layer: false
  optimizer:
    lr: 3.0e-4
ds_name: ??
model_pretrained_path: "random:tiny:seed=11"
img_size: 352
exp_name: "ds_${ds_name}_img_${img_size}_lr${model.optimizer.lr}"
interp: ${import_eval:math.pi}
""")
    return tmp_path


def test_compose_experiment_and_overrides(tree):
    cfg = CL.Composer(tree).compose("train", ["experiment=coop/clipseg", "model.net.context_learner.num_context=10",
                                              "+trainer.accumulate_grad_batches=2", "ds_name=kvasir"])
    assert cfg["seed"] == 12345 and cfg["trainer"]["max_epochs"] == 20 and cfg["trainer"]["accumulate_grad_batches"] == 2
    assert cfg["model"]["net"]["use_new_last_layer"] is False
    assert cfg["model"]["optimizer"]["lr"] == pytest.approx(3.0e-4)
    assert isinstance(cfg["model"]["optimizer"]["lr"], float)
    r = CL.resolve(cfg)
    assert r["exp_name"] == "ds_kvasir_img_352_lr0.0003"
    assert r["interp"] == pytest.approx(3.141592653589793)
    assert r["model"]["net"]["model_cfg"]["pretrained_model_name_or_path"] == "random:tiny:seed=11"
    with pytest.raises(KeyError):
        CL.Composer(tree).compose("train", ["experiment=coop/clipseg", "trainer.no_such_key=1"])


def test_missing_mandatory_value(tree):
    cfg = CL.Composer(tree).compose("train", ["experiment=coop/clipseg"])
    with pytest.raises(CL.MissingMandatoryValue):
        CL.select(cfg, "ds_name")


def test_cli_group_choice_beats_experiment_override(tree):
    cfg = CL.Composer(tree).compose("train", ["experiment=coop/clipseg", "model=base", "ds_name=x"])
    assert "_target_" not in cfg["model"]


def test_instantiate_maps_reference_targets(tree):
    from tunevlseg_amd import nets, task
    from tunevlseg_amd.nets.context_learner import VPTContextLearner

    cfg = CL.Composer(tree).compose("train", ["experiment=coop/clipseg", "ds_name=x"])
    model_cfg = CL.select(cfg, "model")
    module = CL.instantiate(model_cfg)
    assert isinstance(module, task.ImageTextMaskModule)
    assert isinstance(module.net, nets.VPTCLIPSeg) and isinstance(module.net.context_learner, VPTContextLearner)
    assert module.net.context_learner.context_vectors.shape == (1, 4, 32)
    assert isinstance(module.loss_fn, task.DiceCELoss) and module.loss_fn.lambda_ce == pytest.approx(0.2)
    assert isinstance(module.optimizer, partial) and module.optimizer.func is task.FusedAdamW
    assert module.optimizer.keywords["lr"] == pytest.approx(3.0e-4)
    assert module.scheduler.func is task.ReduceLROnPlateau and module.scheduler.keywords["patience"] == 5
    assert module.net.additive_decoder_layer is None  # experiment override use_new_last_layer: false


@pytest.mark.skipif(not REF_CONFIGS.exists(), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("model", ["vpt_clipseg", "coop/clipseg", "cocoop/clipseg", "maple_clipseg", "shared_attn_clipseg",
                                   "shared_separate_clipseg"])
def test_reference_config_tree_composes(model):
    """The reference's own YAMLs compose and their net/learner targets map onto this package."""
    cfg = CL.Composer(REF_CONFIGS).compose("train", ["experiment=coop/clipseg", f"model={model}", "ds_name=kvasir_polyp", "prompt_index=0",
                                                     "logger=null"])
    net = CL.resolve(cfg, cfg["model"]["net"])
    assert CL.map_target(net["_target_"]).startswith("tunevlseg_amd.nets.")
    assert CL.map_target(net["context_learner"]["_target_"]).startswith("tunevlseg_amd.nets.context_learner.")
    assert CL.locate(net["_target_"]) is not None and CL.locate(net["context_learner"]["_target_"]) is not None
    assert net["model_cfg"]["pretrained_model_name_or_path"] == "CIDAS/clipseg-rd64"
    assert cfg["seed"] == 12345 and cfg["trainer"]["max_epochs"] == 20
    assert CL.resolve(cfg, cfg["model"]["optimizer"])["lr"] == pytest.approx(2.0e-4)


@pytest.mark.skipif(not REF_CONFIGS.exists(), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("model", ["coop/cris", "cocoop/cris"])
def test_reference_cris_config_composes(model):
    """BASELINE configs[2]: experiment=coop/cris composes from the reference's YAMLs and maps onto COOPCRIS here."""
    from tunevlseg_amd import nets

    cfg = CL.Composer(REF_CONFIGS).compose("train", ["experiment=coop/cris", f"model={model}", "ds_name=kvasir_polyp", "prompt_index=0",
                                                     "logger=null"])
    net = CL.resolve(cfg, cfg["model"]["net"])
    assert CL.locate(net["_target_"]) is nets.COOPCRIS
    assert CL.map_target(net["context_learner"]["_target_"]).startswith("tunevlseg_amd.nets.context_learner.")
    mc = net["model_cfg"]
    assert mc["img_size"] == 416 and mc["fpn_in"] == [512, 1024, 1024] and mc["num_layers"] == 3 and mc["word_dim"] == 1024
    assert net["use_new_last_layer"] is True and net["context_learner"]["context_initializer"] == "a photo of a"
