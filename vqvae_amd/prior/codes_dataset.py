"""codes.npy -> training sequences: the data contract between the codebook builder and the prior
(src/data/codes_dataset.py:8-83, src/data/factory.py get_code_loaders).  Images whose grid contains -1 (nodes outside
the largest component) are dropped; a sequence is the flattened H*W grid, input = all but the last token, target = all
but the first."""
from typing import Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


class CodesDataset(Dataset):
    def __init__(self, codes_path: str, labels_path: Optional[str] = None):
        codes = np.load(codes_path)
        labels = torch.load(labels_path) if labels_path else None
        valid = ~(codes == -1).any(axis=(1, 2))
        self.codes = codes[valid]
        self.labels = labels[valid] if labels is not None else None
        n, h, w = self.codes.shape
        self.codes = self.codes.reshape(n, h * w)
        self.seq_len = h * w

    def __len__(self) -> int:
        return len(self.codes)

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, ...]:
        x = torch.from_numpy(self.codes[idx, :-1]).long()
        y = torch.from_numpy(self.codes[idx, 1:]).long()
        return (x, y, self.labels[idx]) if self.labels is not None else (x, y)


class VanillaCodesDataset(Dataset):
    """One code per image (legacy vanilla VAE): sequence [BOS, code], BOS = num_tokens - 1."""

    def __init__(self, codes_path: str, labels_path: Optional[str] = None, num_tokens: int = 512):
        codes = np.load(codes_path)
        labels = torch.load(labels_path) if labels_path else None
        self.bos_token = num_tokens - 1
        valid = codes != -1
        self.codes = codes[valid]
        self.labels = labels[valid] if labels is not None else None
        self.seq_len = 2

    def __len__(self) -> int:
        return len(self.codes)

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, ...]:
        x = torch.tensor([self.bos_token]).long()
        y = torch.tensor([self.codes[idx]]).long()
        return (x, y, self.labels[idx]) if self.labels is not None else (x, y)


def get_code_loaders(codes_path: str, labels_path: Optional[str] = None, batch_size: int = 128, num_workers: int = 4,
                     pin_memory: bool = True, persistent_workers: bool = False, vanilla_vae: bool = False,
                     num_tokens: Optional[int] = None) -> Tuple[DataLoader, DataLoader]:
    """(train loader with shuffling, validation loader over the SAME data without): the reference's loaders."""
    if vanilla_vae:
        if num_tokens is None:
            raise ValueError("`num_tokens` must be provided for VanillaCodesDataset")
        dataset = VanillaCodesDataset(codes_path, labels_path, num_tokens)
    else:
        dataset = CodesDataset(codes_path, labels_path)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=pin_memory, persistent_workers=persistent_workers)
    return DataLoader(dataset, shuffle=True, **kw), DataLoader(dataset, shuffle=False, **kw)
