"""Dataset of `magn_phase_{idx}.pt` tensors written by create_dataset (/root/reference/music_gan/audio/dataset.py:14-44)."""
import re
from os import listdir
from os.path import isdir, isfile, join

import numpy as np
import torch as th
from torch.utils.data import Dataset


class AudioDataset(Dataset):
    def __init__(self, dataset_path: str) -> None:
        super().__init__()
        assert isdir(dataset_path)
        pattern = re.compile(r"^magn_phase_\d+\.pt$")
        files = [f for f in listdir(dataset_path) if isfile(join(dataset_path, f)) and pattern.match(f)]
        self.__all_files = np.array(sorted(files))
        self.__dataset_path = dataset_path

    def __getitem__(self, index: int):
        return th.load(join(self.__dataset_path, self.__all_files[index]))

    def __len__(self):
        return len(self.__all_files)
