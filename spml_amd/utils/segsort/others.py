"""Prototype memory-bank files (host-side I/O mirror of `spml/utils/segsort/others.py`).

On disk the bank is a directory with one `.npy` per image; each file is a pickled dict
with the keys 'prototype' ([M,C] float32) and 'prototype_label' ([M] int64), as written
by `pyscripts/inference/prototype.py:207-211`."""
from pathlib import Path

import numpy as np
import torch

_KEYS = ('prototype', 'prototype_label')


def _read(path):
  record = np.load(str(path), allow_pickle=True).item()
  return tuple(np.asarray(record[k]) for k in _KEYS)


def load_memory_banks(memory_dir):
  """(prototypes [M,C] float32, labels [M] int64) of every file in `memory_dir`, in file
  name order (others.py:11-41).  CPU tensors, like the reference: the caller places them."""
  files = sorted(Path(memory_dir).glob('*.npy')) if Path(memory_dir).is_dir() else []
  assert len(files) > 0, 'No memory stored in the directory'
  protos, labels = zip(*(_read(f) for f in files))
  return (torch.from_numpy(np.concatenate(protos, 0).astype(np.float32)),
          torch.from_numpy(np.concatenate(labels, 0).astype(np.int64)))


def save_memory_bank(path, prototypes, prototype_labels):
  """One image's prototypes in that format (prototype.py:207-211)."""
  record = dict(zip(_KEYS, (prototypes.detach().cpu().numpy().astype(np.float32),
                            prototype_labels.detach().cpu().numpy().astype(np.int64))))
  np.save(str(path), record)
