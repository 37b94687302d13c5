"""Mirror of reference src/data/components/unimed.py:14-19 (`collate_fn`) plus a synthetic stand-in for the UniMed
WebDataset shards (`jpg`/`txt`/`cls` tuples, unimed_datamodule.py:44-46) - there is no dataset and no network here.

The collated batch keeps the reference keys: `image` (list of decoded images), `caption` (list), `label` (LongTensor).
With the device preprocessing of medmoe_amd.data the `image` entries are uint8 [H,W,3] tensors instead of PIL images, and
`caption` entries are token-id rows (or strings when a tokenizer is hooked up, MedMoE.set_vocabulary)."""
from typing import Any, Dict, List

import torch
from torch.utils.data import Dataset


def collate_fn(batch: List[Any]) -> Dict[str, Any]:
    """unimed.py:14-19."""
    return {
        "image": [x[0] for x in batch],
        "caption": [x[1] for x in batch],
        "label": torch.tensor([x[2] for x in batch], dtype=torch.long),
    }


class SyntheticUnimed(Dataset):
    """Seeded (image uint8 [H,W,3], token ids [T], class) tuples with the shapes of the UniMed shards: image sides vary
    (the reference resizes every PIL image to the model size), caption lengths U{8..T}."""

    def __init__(self, n: int = 1024, max_len: int = 77, vocab: int = 28996, n_classes: int = 5, seed: int = 12345,
                 min_side: int = 160, max_side: int = 320):
        self.n, self.T, self.V, self.C, self.seed = n, max_len, vocab, n_classes, seed
        self.min_side, self.max_side = min_side, max_side

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        g = torch.Generator().manual_seed(self.seed + i)
        h = int(torch.randint(self.min_side, self.max_side + 1, (1,), generator=g))
        w = int(torch.randint(self.min_side, self.max_side + 1, (1,), generator=g))
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        ln = int(torch.randint(min(8, self.T), self.T + 1, (1,), generator=g))
        ids = torch.zeros(self.T, dtype=torch.long)
        ids[:ln] = torch.randint(3, self.V, (ln,), generator=g)
        ids[0] = 1
        ids[ln - 1] = 2
        return img, ids, int(torch.randint(0, self.C, (1,), generator=g))
