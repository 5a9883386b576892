"""Seeded synthetic training batches of the shape the reference's data loader
produces (`spml/data/datasets/list_tag_dataset.py`).
There are no datasets on the GPU box; labels are spatially coherent so that the
number of segments stays representative."""
import torch


def _blocky(gen, b, s, cell, low, high, device):
  cells = -(-s // cell)
  grid = torch.randint(low, high, (b, cells, cells), generator=gen, device=device)
  idx = torch.arange(s, device=device) // cell
  return grid[:, idx][:, :, idx]


def make_batch(batch_size, size, num_classes=21, seed=235, device='cpu', supervision='scribble'):
  """-> (datas, targets):
  image [B,3,S,S] f32; semantic_label, instance_label [B,S,S] i64;
  semantic_tag [B,256] i64 (multi-hot of the classes present)."""
  gen = torch.Generator(device=device).manual_seed(seed)
  image = torch.randn(batch_size, 3, size, size, generator=gen, device=device)
  gt = _blocky(gen, batch_size, size, 171, 0, num_classes, device)
  if supervision == 'scribble':         # ~10% of the pixels keep their label (17x17 blobs)
    keep = _blocky(gen, batch_size, size, 17, 0, 10, device) == 0
  else:                                 # image-tag (CAM-like blobs, ~40%)
    keep = _blocky(gen, batch_size, size, 57, 0, 5, device) < 2
  sem = torch.where(keep, gt, torch.full_like(gt, 254))
  sem[:, size - 24:, :] = 255           # padding strip = semantic_ignore_index
  sem[:, :, size - 24:] = 255
  inst = _blocky(gen, batch_size, size, 64, 0, 200, device)
  tag = torch.zeros(batch_size, 256, dtype=torch.long, device=device)
  tag.scatter_(1, gt.reshape(batch_size, -1), 1)
  datas = {'image': image}
  targets = {'semantic_label': sem, 'instance_label': inst, 'semantic_tag': tag}
  return datas, targets
