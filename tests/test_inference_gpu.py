"""N1/N2: full-resolution sliding-window embedding, k-means over the whole
image, prototypes + majority labels, memory-bank files -- the HIP path against the
oracle's restatement of pyscripts/inference/prototype.py:107-211."""
import types

import numpy as np
import pytest
import torch

from oracle import spml_oracle as O
from spml_amd import _ffi, inference
from spml_amd.models.embeddings.resnet_deeplab import ResnetDeeplab
import spml_amd.utils.segsort.others as so

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class TinyEmbedder(torch.nn.Module):
  """Stand-in for the embedding network: same interface, cheap enough for the CPU oracle
  (a smooth 3 -> C map so that k-means sees coherent regions)."""

  def __init__(self, channels, num_clusters):
    super().__init__()
    torch.manual_seed(5)
    self.conv = torch.nn.Conv2d(3, channels, 5, padding=2)
    self.label_divisor = 2048
    self.semantic_ignore_index = 255
    self.kmeans_num_clusters = num_clusters
    self.kmeans_iterations = 10

  def generate_embeddings(self, datas, targets=None, resize_as_input=False):
    return {'embedding': self.conv(datas['image'])}

  generate_clusters = ResnetDeeplab.generate_clusters


def blobs(gen, h, w):
  base = torch.randn(1, 3, h // 8 + 2, w // 8 + 2, generator=gen)
  img = torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=False)
  return img + 0.05 * torch.randn(1, 3, h, w, generator=gen)


@pytest.mark.parametrize('c,pad,valid,crop,stride,k', [
    (16, (70, 90), (60, 83), (48, 48), (32, 32), (3, 3)),
    (32, (96, 64), (96, 64), (64, 64), (40, 40), (2, 4)),
    (8, (50, 50), (41, 50), (50, 50), (33, 33), (2, 2)),          # a single crop
])
def test_full_resolution_prototypes_match_oracle(c, pad, valid, crop, stride, k, tmp_path):
  gen = torch.Generator().manual_seed(c + pad[0])
  image = blobs(gen, *pad)
  sem = torch.randint(0, 5, (valid[0] // 10 + 1, valid[1] // 10 + 1), generator=gen)
  sem = sem.repeat_interleave(10, 0).repeat_interleave(10, 1)[:valid[0], :valid[1]].contiguous()
  model = TinyEmbedder(c, list(k))
  cpu_fn = lambda crop_img: model.conv(crop_img)

  want_emb = O.full_resolution_embedding(cpu_fn, image, crop, stride)
  gmodel = TinyEmbedder(c, list(k)).to(DEV)
  gmodel.load_state_dict(model.state_dict())
  got_emb = inference.embed_full_resolution(gmodel, image.to(DEV), crop, stride)
  torch.testing.assert_close(got_emb.cpu(), want_emb, rtol=2e-5, atol=2e-6)

  w_pr, w_lab, w_map = O.full_resolution_prototypes(cpu_fn, image, sem, crop, stride, k, 2048)
  g_pr, g_lab, g_map = inference.full_resolution_prototypes(gmodel, image.to(DEV), sem, crop, stride)
  assert g_map.shape == w_map.shape == sem.shape
  assert g_pr.shape == w_pr.shape and g_lab.shape == w_lab.shape
  # 10 chaotic k-means iterations on conv outputs that differ in the last bits: statistical
  agree = (g_map.cpu() == w_map).float().mean().item()
  assert agree > 0.97, agree
  same = (g_lab.cpu() == w_lab)
  assert same.float().mean().item() > 0.85
  cos = (g_pr.cpu() * w_pr).sum(1)
  assert (cos[same] > 0.995).float().mean().item() > 0.85

  # memory-bank file in the reference's format, read back by the loader
  inference.save_image_memory(str(tmp_path / 'img0.npy'), g_pr, g_lab)
  p2, l2 = so.load_memory_banks(str(tmp_path))
  assert torch.equal(p2, g_pr.cpu()) and torch.equal(l2, g_lab.cpu())


def test_window_accumulate_kernel_exact_cases():
  """Overlapping windows, zero vectors (eps branch) and bounds checking."""
  gen = torch.Generator().manual_seed(3)
  acc = torch.zeros(6, 20, 24, device=DEV)
  cnt = torch.zeros(20, 24, device=DEV)
  want_acc = torch.zeros(6, 20, 24)
  want_cnt = torch.zeros(20, 24)
  for (sh, sw) in [(0, 0), (5, 8), (8, 12), (0, 12)]:
    patch = torch.randn(6, 12, 12, generator=gen)
    patch[:, 3, 4] = 0.0
    patch[:, 5, 5] *= 1e-15
    _ffi.window_accumulate(patch.to(DEV), acc, cnt, sh, sw)
    n = O.normalize_embedding(patch.permute(1, 2, 0).contiguous()).permute(2, 0, 1)
    want_acc[:, sh:sh + 12, sw:sw + 12] += n
    want_cnt[sh:sh + 12, sw:sw + 12] += 1
  torch.testing.assert_close(acc.cpu(), want_acc, rtol=1e-6, atol=1e-6)
  assert torch.equal(cnt.cpu(), want_cnt)
  with pytest.raises(_ffi.SpmlHipError):
    _ffi.window_accumulate(torch.zeros(6, 12, 12, device=DEV), acc, cnt, 10, 0)
  with pytest.raises(_ffi.SpmlHipError):
    _ffi.window_accumulate(torch.zeros(6, 12, 12), acc, cnt, 0, 0)      # CPU tensor: no fallback


@pytest.mark.parametrize('c,h,w,views', [(64, 13, 11, 2), (20, 9, 16, 1), (32, 32, 32, 3)])
def test_affinity_random_walk_matches_oracle(c, h, w, views):
  """N3: exp(5 cos - 5) affinity, mean over views, 20th power, column normalisation
  (fused kernel) and the 6-step walk against pseudo_camrw_crf.py:143-164 restated."""
  gen = torch.Generator().manual_seed(c + h)
  base = torch.randn(1, c, h // 3 + 2, w // 3 + 2, generator=gen)
  embs = []
  for v in range(views):
    e = torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=False)
    embs.append(e + 0.2 * torch.randn(1, c, h, w, generator=gen))
  cam = torch.rand(21, h, w, generator=gen)
  want, want_t = O.affinity_random_walk(embs, cam, return_transition=True)

  stacked = torch.stack([(e / torch.norm(e, dim=1)).reshape(c, -1) for e in embs], 0)
  got_t = _ffi.affinity_transition(stacked.to(DEV).contiguous())
  torch.testing.assert_close(got_t.cpu(), want_t, rtol=2e-4, atol=1e-9)
  torch.testing.assert_close(got_t.sum(0).cpu(), torch.ones(h * w), rtol=1e-5, atol=1e-5)
  got = inference.affinity_random_walk([e.to(DEV) for e in embs], cam.to(DEV))
  torch.testing.assert_close(got.cpu(), want, rtol=1e-3, atol=1e-6)
  assert torch.equal(got.argmax(0).cpu(), want.argmax(0)) or \
      (got.argmax(0).cpu() != want.argmax(0)).float().mean().item() < 5e-3


# ---------------------------------------------------------------------------
# Against fixtures exec'd from the reference scripts' own lines (tools/gen_golden.py):
# prototype.py:134-205 (n2_window.npz) and pseudo_camrw_crf.py:139-164 (n3_randomwalk.npz).
@pytest.mark.parametrize('ci', [0, 1])
def test_full_resolution_pass_matches_reference_lines(ci):
  from conftest import load_golden
  from test_oracle_golden import n2_case
  g = load_golden('n2_window')
  t, conv, crop, stride, k = n2_case(g, ci)
  model = TinyEmbedder(conv.out_channels, list(k)).to(DEV)
  model.conv.load_state_dict({k_: v.to(DEV) for k_, v in conv.state_dict().items()})
  image, sem = g[t + 'image'].to(DEV), g[t + 'sem']
  emb = inference.embed_full_resolution(model, image, crop, stride)
  torch.testing.assert_close(emb.cpu(), g[t + 'embedding'], rtol=1e-4, atol=2e-6)
  protos, labels, cmap = inference.full_resolution_prototypes(model, image, sem, crop, stride)
  # 10 k-means iterations on a GPU convolution's output (last-bit differences): near ties
  # may flip, so the cluster map is compared statistically, the prototypes where it agrees
  want_map = g[t + 'cluster_index'].view(cmap.shape)
  agree = (cmap.cpu() == want_map).float().mean().item()
  assert agree > 0.97, agree
  assert protos.shape == g[t + 'prototypes'].shape
  same = labels.cpu() == g[t + 'prototype_labels']
  assert same.float().mean().item() > 0.85
  # exact chain given the reference's own clustering: prototypes + majority labels
  cl_emb = O.normalize_embedding(g[t + 'embedding'].permute(0, 2, 3, 1).contiguous())
  h, w = sem.shape
  cl_emb = cl_emb[0, :h, :w].reshape(h * w, -1)
  import spml_amd.utils.segsort.common as sc
  pr = sc.calculate_prototypes_from_labels(cl_emb.to(DEV), g[t + 'cluster_index'].to(DEV))
  torch.testing.assert_close(pr.cpu(), g[t + 'prototypes'], rtol=0, atol=1e-5)
  _, lab = sc.find_majority_label_index(sem.to(DEV), g[t + 'cluster_index'].to(DEV))
  assert torch.equal(lab.cpu(), g[t + 'prototype_labels'])


@pytest.mark.parametrize('ci', [0, 1])
def test_affinity_random_walk_matches_reference_lines(ci):
  from conftest import load_golden
  from test_oracle_golden import n3_views
  g = load_golden('n3_randomwalk')
  embs8 = n3_views(g, ci)
  c = embs8[0].shape[1]
  stacked = torch.stack([(e / torch.norm(e, dim=1)).reshape(c, -1) for e in embs8], 0)
  got_t = _ffi.affinity_transition(stacked.to(DEV).contiguous())
  torch.testing.assert_close(got_t.cpu(), g['c%d_trans' % ci], rtol=1e-4, atol=1e-9)
  got = inference.affinity_random_walk([e.to(DEV) for e in embs8], g['c%d_cam8' % ci].to(DEV),
                                       walk_steps=int(g.walk_steps))
  torch.testing.assert_close(got.cpu(), g['c%d_cam_rw' % ci], rtol=1e-4, atol=1e-6)
