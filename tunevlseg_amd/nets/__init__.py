"""Drop-in counterparts of ``src.models.core_models.coop`` (reference ``__init__.py:1-8``)."""
from . import context_learner as context_learner
from .base_clipseg import BaseCLIPSeg as BaseCLIPSeg
from .coop_clipseg import COOPCLIPSeg as COOPCLIPSeg
from .hf_clipseg_wrapper import HFCLIPSegWrapper as HFCLIPSegWrapper
from .maple_clipseg import BaseMultimodalCLIPSeg as BaseMultimodalCLIPSeg
from .maple_clipseg import MapleCLIPSeg as MapleCLIPSeg
from .vpt_clipseg import VPTCLIPSeg as VPTCLIPSeg
