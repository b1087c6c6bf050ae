"""codes.npy -> next-token training batches.

Contract kept from the reference (src/data/codes_dataset.py:8-83 and `get_code_loaders` in src/data/factory.py): class
and function names, constructor arguments, what a sample is -- `(input, target[, label])` with input = sequence minus its
last token and target = sequence minus its first -- which images survive (a spatial grid containing -1, i.e. a latent
outside the graph's largest component, drops the whole image; a vanilla code of -1 drops that image), the vanilla
sequence `[BOS, code]` with BOS = num_tokens - 1, the ValueError without `num_tokens`, and the ORDER in which a given
torch seed visits the samples (shuffled training pass, sequential validation pass over the same data).

Own design: the whole token matrix is one int64 tensor (60 000 x 16 codes = 7.7 MB) that can live in HBM, and a batch is
one `index_select` of it -- no worker processes, no per-sample Python, no collate.  `ResidentBatches` is the loader; it
draws from the torch generator exactly what torch's DataLoader + RandomSampler draw per pass (one int64 for the loader's
base seed, one for the sampler's seed, then `randperm` on a generator of that seed), so a run visits the same batches
as the reference's loaders under the same seed.
"""
from typing import Iterator, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class CodeSequences(Dataset):
    """Token matrix (n, L) + optional labels (n,); sample i = (tokens[i, :-1], tokens[i, 1:][, labels[i]])."""

    def __init__(self, tokens: torch.Tensor, labels: Optional[torch.Tensor]):
        self.tokens = tokens.to(torch.int64).contiguous()
        self.labels = labels
        self.seq_len = int(self.tokens.shape[1])

    def __len__(self) -> int:
        return int(self.tokens.shape[0])

    def __getitem__(self, i) -> Tuple[torch.Tensor, ...]:
        row = self.tokens[i]
        sample = (row[..., :-1], row[..., 1:])
        return sample if self.labels is None else sample + (self.labels[i],)

    @property
    def codes(self) -> np.ndarray:
        return self.tokens.cpu().numpy()

    def to(self, device) -> "CodeSequences":
        """Move the token matrix (and labels) to `device`; batches are then cut there."""
        self.tokens = self.tokens.to(device)
        if self.labels is not None:
            self.labels = self.labels.to(device)
        return self


def _load(codes_path: str, labels_path: Optional[str]):
    codes = torch.from_numpy(np.ascontiguousarray(np.load(codes_path)))
    labels = torch.load(labels_path) if labels_path else None
    return codes, labels


class CodesDataset(CodeSequences):
    """Spatial codes (N, H, W) -> sequences of H*W tokens in row-major grid order."""

    def __init__(self, codes_path: str, labels_path: Optional[str] = None):
        codes, labels = _load(codes_path, labels_path)
        grid = codes.reshape(codes.shape[0], -1)
        keep = ~(grid == -1).any(dim=1)
        super().__init__(grid[keep], labels[keep] if labels is not None else None)


class VanillaCodesDataset(CodeSequences):
    """One code per image (legacy vanilla VAE): the sequence is [BOS, code] with BOS = num_tokens - 1, so the input is
    [BOS] and the target [code]."""

    def __init__(self, codes_path: str, labels_path: Optional[str] = None, num_tokens: int = 512):
        codes, labels = _load(codes_path, labels_path)
        self.bos_token = num_tokens - 1
        keep = codes != -1
        kept = codes[keep].to(torch.int64)
        super().__init__(torch.stack([torch.full_like(kept, self.bos_token), kept], dim=1),
                         labels[keep] if labels is not None else None)


class ResidentBatches:
    """Batches of a CodeSequences cut on the tensor's own device.  `shuffle=True` visits the samples in the order
    DataLoader(shuffle=True) would under the same generator state; the last batch may be short (drop_last=False)."""

    def __init__(self, data: CodeSequences, batch_size: int, shuffle: bool, generator: Optional[torch.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.generator = data, int(batch_size), bool(shuffle), generator
        if self.batch_size < 1:
            raise ValueError("batch_size must be positive")

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _order(self) -> Optional[torch.Tensor]:
        # the draws torch's loader makes per pass, in its order (dataloader.py base seed; sampler.py RandomSampler)
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        if not self.shuffle:
            return None
        seed = int(torch.empty((), dtype=torch.int64).random_(generator=self.generator).item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(len(self.dataset), generator=g)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, ...]]:
        data, order = self.dataset, self._order()
        if order is not None:
            order = order.to(data.tokens.device)
        for lo in range(0, len(data), self.batch_size):
            hi = min(lo + self.batch_size, len(data))
            pick = order[lo:hi] if order is not None else slice(lo, hi)
            yield data[pick]


def get_code_loaders(codes_path: str, labels_path: Optional[str] = None, batch_size: int = 128, num_workers: int = 4,
                     pin_memory: bool = True, persistent_workers: bool = False, vanilla_vae: bool = False,
                     num_tokens: Optional[int] = None, device=None,
                     generator: Optional[torch.Generator] = None) -> Tuple[ResidentBatches, ResidentBatches]:
    """(shuffled training batches, sequential validation batches over the SAME data) -- the reference's pair.
    `num_workers`, `pin_memory`, `persistent_workers` are accepted for drop-in compatibility and unused: nothing is
    loaded per batch.  `device` places the token matrix (e.g. the rank's GPU); `generator` decouples the shuffling from
    the global torch generator (the data-parallel CLI passes one seeded identically on every rank)."""
    if vanilla_vae:
        if num_tokens is None:
            raise ValueError("`num_tokens` must be provided for VanillaCodesDataset")
        data: CodeSequences = VanillaCodesDataset(codes_path, labels_path, num_tokens)
    else:
        data = CodesDataset(codes_path, labels_path)
    if device is not None:
        data.to(device)
    return ResidentBatches(data, batch_size, True, generator), ResidentBatches(data, batch_size, False, generator)
