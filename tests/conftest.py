"""Shared pytest fixtures.  `-m gpu` tests need a real MI355X (run via gpurun);
everything else must pass on CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950) device')


def pytest_collection_modifyitems(config, items):
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU in this container')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


class Golden(dict):
  """npz fixture -> dict of torch tensors (ints as int64, floats as fp32)."""

  def __getattr__(self, k):
    return self[k]


def load_golden(name):
  z = np.load(os.path.join(GOLDEN, name + '.npz'))
  out = Golden()
  for k in z.files:
    a = z[k]
    if a.dtype.kind in 'US':              # arrays of names
      out[k] = [str(v) for v in a.reshape(-1)]
      continue
    out[k] = torch.from_numpy(np.ascontiguousarray(a)) if a.ndim > 0 else a.item()
  return out


@pytest.fixture
def golden():
  return load_golden
