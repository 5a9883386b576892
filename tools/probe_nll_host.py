import time, torch, sys
sys.path.insert(0, '/root/repo')
import spml_amd
from spml_amd.utils.segsort.loss import SegSortLoss
import spml_amd.utils.segsort.common as sc
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(1)
P, D, M = 16900, 66, 150
e = torch.nn.functional.normalize(torch.randn(P, D, device=dev, generator=g), dim=1).requires_grad_(True)
c = torch.randint(0, M, (P,), device=dev, generator=g)
p_lab = torch.randint(0, 40, (M,), device=dev, generator=g)
lab = p_lab[c]
loss = SegSortLoss(concentration=16)
def once():
  pr = sc.calculate_prototypes_from_labels(e, c, M)
  l = loss(e, lab, c, pr, p_lab, codes32=True)
  l.backward()
for _ in range(5): once()
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = 64
for _ in range(N): once()
t1 = time.perf_counter()
b.record(); torch.cuda.synchronize()
print('host us per fwd+bwd: %.1f   gpu us: %.1f' % ((t1 - t0) / N * 1e6, a.elapsed_time(b) / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): once()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
