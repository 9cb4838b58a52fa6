"""Input side of the hot path: prompt tokenisation, pad-to-longest collation and the device-side image /
mask preparation that sits between a decoded sample and ``net(text_input, image_input)``."""
from .collate import PadToLongestCollator
from .datamodule import DeviceLoader, ImageTextDatamodule, ShardSampler
from .dataset import DeviceBatchPrep, ImageDirTextMaskDataset, ImageTextMaskDataset, ResizeTransform
from .tokenizer import ClipBpeTokenizer
from .transforms import Compose, DeviceTransform, RaggedCollator

__all__ = ["ClipBpeTokenizer", "Compose", "DeviceBatchPrep", "DeviceLoader", "DeviceTransform", "ImageDirTextMaskDataset", "ImageTextDatamodule",
           "ImageTextMaskDataset", "PadToLongestCollator", "RaggedCollator", "ResizeTransform", "ShardSampler"]
