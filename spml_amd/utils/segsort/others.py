"""MI355X mirror of `spml/utils/segsort/others.py` (prototype memory-bank files).

Host-side I/O only: the on-disk format is the reference's -- one `.npy` per image holding
a pickled dict `{'prototype': [M,C] float32, 'prototype_label': [M] int64}`
(written by `pyscripts/inference/prototype.py:207-211`)."""
import glob
import os

import numpy as np
import torch


def load_memory_banks(memory_dir):
  """All prototypes and labels stored in `memory_dir`, files in name order
  (others.py:11-41).  Returns a `[M,C]` float tensor and a `[M]` long tensor (CPU,
  like the reference: the caller moves them to its device)."""
  memory_paths = sorted(glob.glob(os.path.join(memory_dir, '*.npy')))
  assert len(memory_paths) > 0, 'No memory stored in the directory'
  prototypes, prototype_labels = [], []
  for memory_path in memory_paths:
    datas = np.load(memory_path, allow_pickle=True).item()
    prototypes.append(datas['prototype'])
    prototype_labels.append(datas['prototype_label'])
  prototypes = torch.FloatTensor(np.concatenate(prototypes, 0))
  prototype_labels = torch.LongTensor(np.concatenate(prototype_labels, 0))
  return prototypes, prototype_labels


def save_memory_bank(path, prototypes, prototype_labels):
  """Write one image's prototypes in the reference's format (prototype.py:207-211)."""
  np.save(path, {'prototype': prototypes.detach().cpu().numpy().astype(np.float32),
                 'prototype_label': prototype_labels.detach().cpu().numpy().astype(np.int64)})
