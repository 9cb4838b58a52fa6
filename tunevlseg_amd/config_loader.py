"""Hydra-shaped config composition + ``_target_`` instantiation without Hydra/OmegaConf.

The reference is launched as ``python src/train.py experiment=coop/clipseg model=vpt_clipseg ...``
(``src/train.py:139-158``) over the config tree ``configs/**.yaml``.  Hydra, OmegaConf and Lightning are not
installed here, so this module implements the subset those YAMLs use (SURVEY.md §5 "Config / flags"):

* defaults lists (``_self_``, ``group: option``, ``override /group: option``, ``optional group: option``,
  ``group: null``) and ``# @package _global_`` files;
* command-line overrides ``key=value``, ``+key=value``, ``group=option``;
* ``${a.b}`` interpolation, ``${oc.env:VAR[,default]}``, and the reference's custom resolvers
  ``${import_eval:mod.attr}`` / ``${literal_eval:expr}`` (``src/utils/resolvers.py:16-77``);
* ``_target_`` / ``_partial_`` / ``_args_`` instantiation (``hydra.utils.instantiate``), with the reference's import
  paths mapped onto this package, so the reference's model YAMLs are usable unmodified.

Accessing a mandatory ``??`` value raises, like OmegaConf's ``MissingMandatoryValue``.
"""
from __future__ import annotations

import copy
import importlib
import os
import re
from functools import partial
from pathlib import Path
from typing import Any

import yaml

# reference import path -> drop-in here (constructor keywords are identical)
TARGET_MAP = {
    "src.models.core_models.coop.context_learner.": "tunevlseg_amd.nets.context_learner.",
    "src.models.core_models.coop.": "tunevlseg_amd.nets.",
    "src.models.components.hf_clipseg_wrapper.HFCLIPSegWrapper": "tunevlseg_amd.nets.HFCLIPSegWrapper",
    "src.models.image_text_mask_module.ImageTextMaskModule": "tunevlseg_amd.task.ImageTextMaskModule",
    "monai.losses.DiceCELoss": "tunevlseg_amd.task.DiceCELoss",
    "torch.optim.AdamW": "tunevlseg_amd.task.FusedAdamW",
    "torch.optim.lr_scheduler.ReduceLROnPlateau": "tunevlseg_amd.task.ReduceLROnPlateau",
    # input side (configs/data/image_text_mask.yaml, configs/experiment/**: train_transforms / collate_fn)
    "src.data.image_text_mask_datamodule.ImageTextDatamodule": "tunevlseg_amd.data.datamodule.ImageTextDatamodule",
    "src.data.core_datasets.ImageTextMaskDataset": "tunevlseg_amd.data.dataset.ImageTextMaskDataset",
    "src.data.core_datasets.ImageDirTextMaskDataset": "tunevlseg_amd.data.dataset.ImageDirTextMaskDataset",
    "src.data.components.data_collator.CustomDataCollatorWithPadding": "tunevlseg_amd.data.collate.PadToLongestCollator",
    "transformers.AutoTokenizer.from_pretrained": "tunevlseg_amd.data.dataset.load_tokenizer",
    "albumentations.pytorch.ToTensorV2": "tunevlseg_amd.data.transforms.ToTensorV2",
    "albumentations.": "tunevlseg_amd.data.transforms.",
}

MISSING = "??"


class MissingMandatoryValue(KeyError):
    pass


class _Loader(yaml.SafeLoader):
    pass


# YAML 1.1 does not read "2.0e-4" style floats without a sign in the exponent as float: fix like OmegaConf does
_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?|[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?|[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def load_yaml(path: Path) -> tuple[dict, bool]:
    text = Path(path).read_text()
    is_global = bool(re.match(r"\s*#\s*@package\s+_global_", text))
    data = yaml.load(text, Loader=_Loader) or {}
    return data, is_global


def deep_merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: dict, dotted: str, value: Any, must_exist: bool) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if must_exist:
                raise KeyError(f"override key {dotted!r} does not exist (use +{dotted}=... to add it)")
            node[k] = {}
        node = node[k]
    if must_exist and keys[-1] not in node:
        raise KeyError(f"override key {dotted!r} does not exist (use +{dotted}=... to add it)")
    node[keys[-1]] = value


def _parse_value(text: str) -> Any:
    return yaml.load(text, Loader=_Loader)


class Composer:
    def __init__(self, config_dir: str | Path):
        self.root = Path(config_dir)

    def _find(self, group: str, option: str) -> Path | None:
        p = self.root / group / f"{option}.yaml"
        return p if p.exists() else None

    def _load_group(self, group: str, option: str, out: dict, choices: dict[str, str | None], optional: bool = False) -> None:
        path = self._find(group, option)
        if path is None:
            if optional:
                return
            raise FileNotFoundError(f"config group {group!r} has no option {option!r} under {self.root}")
        data, is_global = load_yaml(path)
        self._compose_file(data, is_global, group, out, choices)

    def _compose_file(self, data: dict, is_global: bool, group: str | None, out: dict, choices: dict[str, str | None]) -> None:
        data = dict(data)
        defaults = data.pop("defaults", None) or ["_self_"]
        if "_self_" not in defaults:
            defaults = [*defaults, "_self_"]
        for entry in defaults:
            if entry == "_self_":
                if is_global or group is None:
                    deep_merge(out, data)
                else:
                    node = out
                    for k in group.split("/")[:1]:  # nested options (coop/clipseg) still land under the top group key
                        node = node.setdefault(k, {})
                    deep_merge(node, data)
                continue
            if isinstance(entry, str):  # "- other_file" relative include
                self._load_group(group or "", entry, out, choices)
                continue
            (key, option), = entry.items()
            key = key.strip()
            optional = key.startswith("optional ")
            key = key.removeprefix("optional ").removeprefix("override ").strip()
            g = key.lstrip("/")
            if key.startswith("/") or group is None:
                sub_group = g
            else:
                sub_group = f"{group}/{g}"
            option = choices.get(sub_group, option)
            if option is None or sub_group == "hydra" or sub_group.startswith("hydra/"):
                continue  # the hydra group configures Hydra's own runtime (logging plugins, sweepers): not part of the job config
            if isinstance(option, str) and option.startswith("/"):
                option = option[1:]
            choices.setdefault(sub_group, option)
            self._load_group(sub_group, option, out, choices, optional)

    def compose(self, config_name: str = "train", overrides: list[str] | None = None) -> dict:
        overrides = list(overrides or [])
        data, is_global = load_yaml(self.root / f"{config_name}.yaml")
        choices: dict[str, str | None] = {}
        value_overrides: list[tuple[str, str, bool]] = []
        for ov in overrides:
            key, _, val = ov.partition("=")
            add = key.startswith("+")
            key = key.lstrip("+")
            if "." not in key and (self.root / key).is_dir() and not add:
                choices[key] = None if val in ("null", "None", "") else val
            else:
                value_overrides.append((key, val, add))
        # experiment files may override groups chosen in the primary defaults: collect their choices first
        exp = choices.get("experiment")
        if exp:
            exp_data, _ = load_yaml(self.root / "experiment" / f"{exp}.yaml")
            for entry in exp_data.get("defaults", []) or []:
                if isinstance(entry, dict):
                    (k, v), = entry.items()
                    if k.strip().startswith("override "):
                        g = k.strip().removeprefix("override ").strip().lstrip("/")
                        choices.setdefault(g, v)
        out: dict = {}
        self._compose_file(data, True, None, out, choices)
        for key, val, add in value_overrides:
            _set_path(out, key, _parse_value(val), must_exist=not add)
        return out


# ----------------------------------------------------------------------------------------------
# interpolation
# ----------------------------------------------------------------------------------------------
_INTERP = re.compile(r"\$\{([^${}]+)\}")


def import_resolver(string: str):
    """reference ``src/utils/resolvers.py:16-46``"""
    parts = string.split(".", 1)
    if len(parts) != 2:
        raise ValueError("The string must be a module path")
    module, rest = parts
    if module == "cv2":   # OpenCV is not in the image: its enum values are what the YAML files ask for (cv2.INTER_CUBIC, cv2.BORDER_REPLICATE)
        try:
            importlib.import_module("cv2")
        except ModuleNotFoundError:
            from .data.transforms import CV2_CONSTANTS

            if rest not in CV2_CONSTANTS:
                raise KeyError(f"cv2.{rest}: cv2 is not installed and this constant is not in the built-in table") from None
            return CV2_CONSTANTS[rest]
    obj = importlib.import_module(module)
    for attr in rest.split("."):
        obj = getattr(obj, attr)
    return obj


def _lookup(cfg: dict, dotted: str) -> Any:
    node: Any = cfg
    for k in dotted.split("."):
        if isinstance(node, list):
            node = node[int(k)]
        else:
            if k not in node:
                raise KeyError(f"interpolation key {dotted!r} not found")
            node = node[k]
    return node


def _resolve_expr(expr: str, root: dict, stack: tuple[str, ...]) -> Any:
    expr = expr.strip()
    if ":" in expr and not expr.startswith("."):
        name, _, arg = expr.partition(":")
        if name == "oc.env":
            var, _, default = arg.partition(",")
            val = os.environ.get(var.strip())
            if val is None:
                if not _:
                    raise KeyError(f"environment variable {var!r} is not set")
                return _parse_value(default.strip())
            return val
        if name == "import_eval":
            return import_resolver(arg.strip())
        if name == "literal_eval":
            return eval(arg)  # noqa: S307 - the reference registers python eval for this resolver (resolvers.py:66)
        if name in ("hydra", "now"):
            return f"${{{expr}}}"  # runtime-only hydra values: left verbatim
        raise KeyError(f"unknown resolver {name!r}")
    if expr in stack:
        raise ValueError(f"interpolation cycle through {expr!r}")
    val = _lookup(root, expr)
    return _resolve_value(val, root, (*stack, expr))


def _resolve_value(val: Any, root: dict, stack: tuple[str, ...] = ()) -> Any:
    if isinstance(val, dict):
        return {k: _resolve_value(v, root, stack) for k, v in val.items()}
    if isinstance(val, list):
        return [_resolve_value(v, root, stack) for v in val]
    if not isinstance(val, str) or "${" not in val:
        return val
    m = _INTERP.fullmatch(val)
    if m:  # whole-value interpolation keeps the referenced type
        return _resolve_expr(m.group(1), root, stack)
    prev = None
    while prev != val and "${" in val:
        prev = val
        val = _INTERP.sub(lambda mm: str(_resolve_expr(mm.group(1), root, stack)), val)
    return val


def resolve(cfg: dict, node: Any = None) -> Any:
    """Resolve every ``${...}`` under ``node`` (default: the whole config) against ``cfg``."""
    return _resolve_value(cfg if node is None else node, cfg)


def select(cfg: dict, dotted: str, resolve_values: bool = True) -> Any:
    val = _lookup(cfg, dotted)
    val = _resolve_value(val, cfg) if resolve_values else val
    _check_missing(val, dotted)
    return val


def _check_missing(val: Any, path: str) -> None:
    if isinstance(val, str) and val == MISSING:
        raise MissingMandatoryValue(f"Missing mandatory value: {path}")
    if isinstance(val, dict):
        for k, v in val.items():
            _check_missing(v, f"{path}.{k}")
    if isinstance(val, list):
        for i, v in enumerate(val):
            _check_missing(v, f"{path}.{i}")


# ----------------------------------------------------------------------------------------------
# instantiate
# ----------------------------------------------------------------------------------------------
def map_target(target: str) -> str:
    for src, dst in TARGET_MAP.items():
        if target.startswith(src):
            return dst + target[len(src):]
    return target


def locate(target: str) -> Any:
    target = map_target(target)
    parts = target.split(".")
    for i in range(len(parts), 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:i]))
        except ModuleNotFoundError:
            continue
        for attr in parts[i:]:
            obj = getattr(obj, attr)
        return obj
    raise ImportError(f"cannot locate {target!r}")


def instantiate(node: Any, *args, **kwargs) -> Any:
    """``hydra.utils.instantiate`` for already-resolved config nodes (recursive, ``_partial_`` aware)."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    _check_missing(node, node["_target_"])
    fn = locate(node["_target_"])
    is_partial = bool(node.get("_partial_", False))
    pos = [instantiate(a) for a in node.get("_args_", [])]
    kw = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_", "_args_", "_recursive_", "_convert_")}
    kw.update(kwargs)
    if is_partial:
        return partial(fn, *pos, *args, **kw)
    return fn(*pos, *args, **kw)
