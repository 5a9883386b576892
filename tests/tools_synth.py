"""Seeded synthetic inputs shared by tests and bench."""
import torch


def coherent_rows(gen, n, d, side, noise=0.3, blobs=7):
  """n images of side*side unit-norm rows with spatially coherent structure."""
  dirs = torch.randn(blobs, d, generator=gen)
  cy = torch.rand(n, blobs, generator=gen)
  cx = torch.rand(n, blobs, generator=gen)
  yy = torch.linspace(0, 1, side).view(1, 1, side, 1)
  xx = torch.linspace(0, 1, side).view(1, 1, 1, side)
  wgt = torch.exp(-((yy - cy.view(n, blobs, 1, 1)) ** 2 + (xx - cx.view(n, blobs, 1, 1)) ** 2) / 0.03)
  e = torch.einsum('nbhw,bd->nhwd', wgt, dirs) + noise * torch.randn(n, side, side, d, generator=gen)
  e = e.reshape(n, side * side, d)
  return e / e.norm(dim=-1, keepdim=True).clamp(min=1e-12)
