"""Map-style dataset over the tensors create_dataset writes (interface of /root/reference/music_gan/audio/dataset.py:14-44:
`AudioDataset(dataset_path)`, items are the float64 (2, 512, 512) tensors of `magn_phase_{idx}.pt`, in file-name order)."""
import fnmatch
import os

import torch
from torch.utils.data import Dataset

_PATTERN = "magn_phase_*.pt"


def _sample_files(folder: str):
    with os.scandir(folder) as it:
        names = [e.name for e in it if e.is_file() and fnmatch.fnmatchcase(e.name, _PATTERN)
                 and e.name[len("magn_phase_"):-len(".pt")].isdigit()]
    names.sort()  # plain string order, as the reference sorts them
    return tuple(names)


class AudioDataset(Dataset):
    def __init__(self, dataset_path: str) -> None:
        super().__init__()
        assert os.path.isdir(dataset_path)
        self._root = dataset_path
        self._names = _sample_files(dataset_path)

    def __len__(self) -> int:
        return len(self._names)

    def __getitem__(self, index: int) -> torch.Tensor:
        return torch.load(os.path.join(self._root, self._names[index]))
