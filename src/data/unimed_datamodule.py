"""Mirror of reference src/data/unimed_datamodule.py:16-140 (`UnimedDataModule`): same constructor arguments and loader
methods.  Lightning and webdataset are optional in this image: with `webdataset` importable and shard paths given, the
loaders read `jpg`/`txt`/`cls` tuples as the reference does (:44-46); otherwise they fall back to the seeded
SyntheticUnimed set.  `to_device_batch` turns a collated batch into what the MI355X model takes: images preprocessed on the
GPU (medmoe_amd.data), captions as id / mask tensors."""
from typing import Any, Dict, Optional

import torch
from torch.utils.data import DataLoader

from src.data.components.unimed import SyntheticUnimed, collate_fn

try:                                                     # pragma: no cover - not installed in this image
    from lightning import LightningDataModule as _Base
except Exception:                                        # noqa: BLE001
    _Base = object

try:                                                     # pragma: no cover
    import webdataset as wds
except Exception:                                        # noqa: BLE001
    wds = None


class UnimedDataModule(_Base):
    def __init__(self, data_dir: str = "data/", transformations: Any = None, batch_size: int = 64, num_workers: int = 0,
                 pin_memory: bool = False, train_data_paths: Optional[str] = None, val_data_paths: Optional[str] = None,
                 synthetic_size: int = 4096, max_len: int = 77, synthetic_vocab: int = 28996, synthetic_classes: int = 5) -> None:
        super().__init__()
        self.data_dir, self.transformations = data_dir, transformations
        self.batch_size, self.num_workers, self.pin_memory = batch_size, num_workers, pin_memory
        self.batch_size_per_device = batch_size
        self.collate_fn = collate_fn

        def make(paths, seed):
            if wds is not None and paths:                # unimed_datamodule.py:44-46
                return wds.WebDataset(paths, resampled=True, shardshuffle=True, nodesplitter=wds.split_by_node) \
                    .decode("pil").to_tuple("jpg", "txt", "cls")
            return SyntheticUnimed(synthetic_size, max_len=max_len, vocab=synthetic_vocab, n_classes=synthetic_classes, seed=seed)

        self.data_train = make(train_data_paths, 12345)
        self.data_val = make(val_data_paths, 54321)
        self.data_test = make(val_data_paths, 54321)

    def setup(self, stage: Optional[str] = None, world_size: int = 1) -> None:
        """unimed_datamodule.py:63-79: the global batch is split evenly over the devices."""
        trainer = getattr(self, "trainer", None)
        ws = trainer.world_size if trainer is not None else world_size
        if self.batch_size % ws != 0:
            raise RuntimeError(f"Batch size ({self.batch_size}) is not divisible by the number of devices ({ws}).")
        self.batch_size_per_device = self.batch_size // ws

    def _loader(self, ds) -> DataLoader:
        """Under data parallelism every rank reads its own 1/W of a map-style set (rank-strided indices, no shuffle: the shards of the
        reference are split by `wds.split_by_node`, unimed_datamodule.py:44-46, which the WebDataset branch above keeps)."""
        trainer = getattr(self, "trainer", None)
        ws = getattr(trainer, "world_size", 1) if trainer is not None else 1
        sampler = None
        if ws > 1 and hasattr(ds, "__len__") and hasattr(ds, "__getitem__"):
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, num_replicas=ws, rank=trainer.global_rank, shuffle=False, drop_last=False)
        return DataLoader(dataset=ds, batch_size=self.batch_size_per_device, num_workers=self.num_workers,
                          pin_memory=self.pin_memory, shuffle=False, sampler=sampler, collate_fn=self.collate_fn)

    def train_dataloader(self) -> DataLoader:
        return self._loader(self.data_train)

    def val_dataloader(self) -> DataLoader:
        return self._loader(self.data_val)

    def test_dataloader(self) -> DataLoader:
        return self._loader(self.data_test)

    @staticmethod
    def to_device_batch(batch: Dict[str, Any], device, size: int = 224) -> Dict[str, Any]:
        """Collated host batch -> the dict MedMoE.forward / Engine.train_step take (images resized + normalised on the GPU)."""
        from medmoe_amd.data import preprocess_images
        imgs = [torch.as_tensor(im).to(device, non_blocking=True).contiguous() for im in batch["image"]]
        image = preprocess_images(imgs, size=size)
        ids = torch.stack([torch.as_tensor(c) for c in batch["caption"]]).to(device)
        mask = (ids != 0).long()
        return {"image": image, "ids": ids, "attn_mask": mask, "token_type": torch.zeros_like(ids), "label": batch["label"].to(device),
                "caption": {"ids": ids, "attn_mask": mask}}
