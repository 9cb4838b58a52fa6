"""Input side of the hot path: prompt tokenisation, pad-to-longest collation and the device-side image /
mask preparation that sits between a decoded sample and ``net(text_input, image_input)``."""
from .collate import PadToLongestCollator
from .dataset import DeviceBatchPrep, ImageTextMaskDataset, ResizeTransform
from .tokenizer import ClipBpeTokenizer

__all__ = ["ClipBpeTokenizer", "DeviceBatchPrep", "ImageTextMaskDataset", "PadToLongestCollator", "ResizeTransform"]
