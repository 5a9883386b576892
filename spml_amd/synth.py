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


def make_batch(batch_size, size, num_classes=21, seed=235, device='cpu', supervision='scribble', palette=None):
  """-> (datas, targets):
  image [B,3,S,S] f32; semantic_label, instance_label [B,S,S] i64;
  semantic_tag [B,256] i64 (multi-hot of the classes present).

  palette = (lo, hi): every image draws its regions from background + lo..hi object classes of its own, like a
  VOC image (`list_tag_dataset.py:75-78`: the tag set is what occurs in the label map) -- image tag sets then
  differ, and the co-occurrence term (`segsort.py:147-151,205-211`) has negatives.  None: every region draws from
  all classes (9 regions of 21 classes: nearly every pair of images shares a class; the generator of rounds 1-4,
  which the golden fixtures were made with)."""
  gen = torch.Generator(device=device).manual_seed(seed)
  image = torch.randn(batch_size, 3, size, size, generator=gen, device=device)
  if palette is None:
    gt = _blocky(gen, batch_size, size, 171, 0, num_classes, device)
  else:
    lo, hi = palette
    n_obj = torch.randint(lo, hi + 1, (batch_size,), generator=gen, device=device)          # object classes per image
    order = torch.rand(batch_size, num_classes - 1, generator=gen, device=device).argsort(1) + 1   # a shuffle of 1..C-1
    pal = torch.cat([torch.zeros(batch_size, 1, dtype=torch.long, device=device), order[:, :hi]], 1)   # [B, 1 + hi]
    cells = -(-size // 171)
    pick = (torch.rand(batch_size, cells, cells, generator=gen, device=device) *
            (n_obj + 1).view(-1, 1, 1).float()).long().clamp_(max=hi)                        # 0 .. n_obj per cell
    grid = torch.gather(pal, 1, pick.view(batch_size, -1)).view(batch_size, cells, cells)
    idx = torch.arange(size, device=device) // 171
    gt = grid[:, idx][:, :, idx]
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
