"""Accuracy of the NLL kernels against an fp64 evaluation of the oracle at a chosen prototype count (default
M = 100 003, D = 64, tag-set predicate, image-major codes):  python tools/probe_nll_accuracy.py [M]
Prints the errors of num / den, of dEmbedding on the v3 / v2 / round-2 kernels and of dPrototypes."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import spml_oracle as O
from spml_amd import _ffi
DEV='cuda:0'
gen = torch.Generator().manual_seed(100003)
p, m, d, kappa = 1500, int(sys.argv[1]) if len(sys.argv) > 1 else 100003, 64, 12.0
protos = O.normalize_embedding(torch.randn(m, d, generator=gen))
own = torch.randint(0, m, (p,), generator=gen)
emb = O.normalize_embedding(protos[own] + 0.8 * torch.randn(p, d, generator=gen))
n_run = (m + 999) // 1000
run_tags = torch.zeros(n_run, 20, dtype=torch.long)
run_tags.scatter_(1, torch.stack([torch.randperm(20, generator=gen)[:2] for _ in range(n_run)]), 1)
p_tags = run_tags.repeat_interleave(1000, dim=0)[:m]
tags = p_tags[own]
wgt = torch.rand(p, generator=gen) / p
def oracle(dtype):
  pr = protos.to(dtype).requires_grad_(True)
  nlls, des, nums, dens = [], [], [], []
  for lo in range(0, p, 500):
    e = emb[lo:lo + 500].to(dtype).requires_grad_(True)
    part = O.set_segsort_nll(e, tags[lo:lo + 500], own[lo:lo + 500], pr, p_tags, kappa).view(-1)
    (part * wgt[lo:lo + 500].to(dtype)).sum().backward()
    nlls.append(part.detach()); des.append(e.grad)
    with torch.no_grad():
      sim = ((e @ pr.t()) * kappa).exp()
      own_s = sim.gather(1, own[lo:lo + 500].view(-1, 1)).view(-1)
      same = (tags[lo:lo + 500].to(dtype) @ p_tags.to(dtype).t()) > 0
      nums.append((sim * same).sum(1) - own_s); dens.append(sim.sum(1) - own_s)
  return torch.cat(nlls), torch.cat(des), pr.grad, torch.cat(nums), torch.cat(dens)
n64, de64, dp64, num64, den64 = oracle(torch.float64)
n32, de32, dp32, _, _ = oracle(torch.float32)
w = lambda t: torch.tensor(0)
def tags_to_mask(t):
  return (t.long() * (2 ** torch.arange(t.shape[1], dtype=torch.long)).view(1, -1)).sum(1)
px_code, pr_code = tags_to_mask(tags).to(DEV), tags_to_mask(p_tags).to(DEV)
scale = de64.abs().max().item()
print('scale', scale, 'oracle32 vs 64: max %.2e mean %.2e (rel to scale)' % ((de32.double() - de64).abs().max().item() / scale, (de32.double() - de64).abs().mean().item() / scale))
for mode in (1 | 4, 1):
  nll, stats = _ffi.segsort_nll_fwd(emb.to(DEV), own.to(DEV), px_code, protos.to(DEV), pr_code, kappa, mode)
  st = stats.cpu().double()
  print('mode', mode, 'num rel err max %.2e, den rel err max %.2e; nll abs err max %.2e' % (
      ((st[:, 0] - num64).abs() / num64.abs()).max().item(), ((st[:, 1] - den64).abs() / den64.abs()).max().item(),
      (nll.cpu().double() - n64).abs().max().item()))
  for env in ({}, {'SPML_NLL_DE3': '0'}, {'SPML_NLL_BWD2': '0'}):
    for k in ('SPML_NLL_DE3', 'SPML_NLL_BWD2'): os.environ.pop(k, None)
    os.environ.update(env)
    de, dp = _ffi.segsort_nll_bwd(emb.to(DEV), own.to(DEV), px_code, protos.to(DEV), pr_code, kappa, mode, stats, wgt.to(DEV), m_grad=m // 3)
    err = (de.cpu().double() - de64).abs()
    i = err.argmax().item()
    print('  ', env, 'dE err max %.2e mean %.2e (rel to scale) at pixel %d ch %d; row scale %.2e; dp err max %.2e' % (
        err.max().item() / scale, err.mean().item() / scale, i // d, i % d, de64[i // d].abs().max().item() / scale,
        (dp.cpu().double()[:m // 3] - dp64[:m // 3]).abs().max().item() / dp64[:m // 3].abs().max().item()), flush=True)
  # exact stats from fp64 fed to the backward
  st64 = stats.clone(); st64[:, 0] = num64.float().to(DEV); st64[:, 1] = den64.float().to(DEV)
  for k in ('SPML_NLL_DE3', 'SPML_NLL_BWD2'): os.environ.pop(k, None)
  de, dp = _ffi.segsort_nll_bwd(emb.to(DEV), own.to(DEV), px_code, protos.to(DEV), pr_code, kappa, mode, st64, wgt.to(DEV), m_grad=m // 3)
  err = (de.cpu().double() - de64).abs()
  print('   with fp64 num/den: dE err max %.2e mean %.2e' % (err.max().item() / scale, err.mean().item() / scale))
