"""Host-side input container mirroring ``healnet.etl.MMDataset`` (reference healnet/etl/loaders.py:21-41):
a list of per-modality tensors indexed by sample, with an optional target."""
from typing import List, Optional

import torch
from torch.utils.data import Dataset


class MMDataset(Dataset):
    def __init__(self, tensors: List[torch.Tensor], target: Optional[torch.Tensor] = None):
        if len(tensors) == 0:
            raise ValueError("MMDataset needs at least one modality tensor")
        n = tensors[0].shape[0]
        for t in tensors:
            if t.shape[0] != n:
                raise ValueError("all modalities must hold the same number of samples")
        self.tensors = tensors
        self.target = target

    def __len__(self) -> int:
        return int(self.tensors[0].shape[0])

    def __getitem__(self, idx):
        sample = [t[idx] for t in self.tensors]
        return sample if self.target is None else (sample, self.target[idx])
