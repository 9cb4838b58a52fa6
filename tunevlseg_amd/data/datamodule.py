"""``ImageTextDatamodule`` of the reference (``src/data/image_text_mask_datamodule.py:7-90``) without Lightning: same constructor
keywords, ``data.batch_size`` is the GLOBAL batch (per rank ``batch_size // world_size``, ``ValueError`` when it does not divide), train
shuffles, val / test do not.  What Lightning adds silently under DDP is explicit here: every rank iterates its own shard
(:class:`ShardSampler` = ``DistributedSampler`` semantics: a seeded permutation per epoch for training, padding by wrap-around so that all
ranks run the same number of steps).  The loaders yield DEVICE batches: the ragged uint8 batch is copied once and transformed on the GPU
(``data.transforms.DeviceTransform``), or -- with a legacy host transform -- normalised by ``DeviceBatchPrep``."""
from __future__ import annotations

import math
from collections.abc import Iterator
from typing import Any

import torch
from torch.utils.data import DataLoader, Dataset, Sampler

from .. import dist as tdist
from .collate import PadToLongestCollator
from .transforms import Compose, DeviceTransform, RaggedCollator


class ShardSampler(Sampler[int]):
    def __init__(self, n: int, world: int, rank: int, shuffle: bool, seed: int = 0, drop_last: bool = False) -> None:
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.n, self.world, self.rank, self.shuffle, self.seed, self.epoch = n, world, rank, shuffle, seed, 0
        self.per_rank = n // world if drop_last else math.ceil(n / world)
        self.drop_last = drop_last

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return self.per_rank

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        total = self.per_rank * self.world
        if total > len(order):   # pad by wrapping around (DistributedSampler): every rank takes the same number of steps
            order += (order * math.ceil((total - len(order)) / max(len(order), 1)))[: total - len(order)]
        return iter(order[self.rank:total:self.world])


class DeviceLoader:
    """A DataLoader whose batches come out on the device, ready for ``ImageTextMaskModule``; ``set_epoch`` reseeds sampler and augmentation."""

    def __init__(self, loader: DataLoader, prep, sampler: ShardSampler, seed: int, rank: int) -> None:
        self.loader, self.prep, self.sampler, self.seed, self.rank = loader, prep, sampler, seed, rank

    def set_epoch(self, epoch: int) -> None:
        self.sampler.set_epoch(epoch)
        if hasattr(self.prep, "set_epoch"):
            self.prep.set_epoch(epoch, self.seed, self.rank)

    def __len__(self) -> int:
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            yield self.prep(batch)


class ImageTextDatamodule:
    def __init__(self, train_ds: Dataset | None = None, val_ds: Dataset | None = None, test_ds: Dataset | None = None, batch_size: int = 32,
                 num_workers: int = 4, pin_memory: bool = True, drop_last: bool = False, *args: Any, mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), seed: int = 0, **kwargs: Any) -> None:
        if train_ds is None and val_ds is None and test_ds is None:
            raise ValueError("Either train, validation, or test dataset should be divided.")
        self.train_ds, self.val_ds, self.test_ds = train_ds, val_ds, test_ds
        self.batch_size, self.num_workers, self.pin_memory, self.drop_last = int(batch_size), int(num_workers), bool(pin_memory), bool(drop_last)
        self.mean, self.std, self.seed = tuple(mean), tuple(std), int(seed)
        self.batch_size_per_device = self.batch_size
        self.world, self.rank, self.device = 1, 0, "cuda"

    def setup(self, stage: str | None = None, world_size: int | None = None, rank: int | None = None, device="cuda") -> None:
        r, w = tdist.env_world()[0], tdist.env_world()[1]
        self.world = int(world_size) if world_size is not None else w
        self.rank = int(rank) if rank is not None else r
        self.device = device
        self.batch_size_per_device = tdist.per_device_batch_size(self.batch_size, self.world)

    def _loader(self, ds: Dataset, shuffle: bool) -> DeviceLoader:
        if ds is None:
            raise ValueError("this split has no dataset")
        sampler = ShardSampler(len(ds), self.world, self.rank, shuffle, self.seed, self.drop_last and shuffle)
        tokens = getattr(ds, "collate_fn", None) or PadToLongestCollator(tokenizer=getattr(ds, "tokenizer", None))
        tf = getattr(ds, "transforms", None)
        if isinstance(tf, Compose):
            collate, prep = RaggedCollator(tokens), DeviceTransform(tf, self.device, self.seed)
        else:
            from .dataset import DeviceBatchPrep

            collate, prep = tokens, DeviceBatchPrep(self.mean, self.std, self.device)
        dl = DataLoader(ds, batch_size=self.batch_size_per_device, sampler=sampler, num_workers=self.num_workers, collate_fn=collate,
                        pin_memory=self.pin_memory and torch.cuda.is_available(), drop_last=self.drop_last and shuffle)
        return DeviceLoader(dl, prep, sampler, self.seed, self.rank)

    def train_dataloader(self) -> DeviceLoader:
        return self._loader(self.train_ds, True)

    def val_dataloader(self) -> DeviceLoader:
        return self._loader(self.val_ds, False)

    def test_dataloader(self) -> DeviceLoader:
        return self._loader(self.test_ds, False)
