"""SGD whose `step(lr)` scales a global learning rate by per-group multipliers
(mirror of `lib/nn/optimizer.py:18-104` of the reference: the train loop passes
the scheduled lr every step and parameter groups carry `lr` multipliers 1/2/10/20).
Update rule (optimizer.py:67-104), momentum > 0:
    g   = grad + weight_decay * p
    buf = momentum * buf + (group_lr * lr) * g
    p  -= buf
implemented with torch._foreach ops (one fused launch per group)."""
import torch
from torch.optim.optimizer import Optimizer, required


class SGD(Optimizer):

  def __init__(self, params, lr=required, momentum=0, dampening=0, weight_decay=0,
               nesterov=False):
    if nesterov and (momentum <= 0 or dampening != 0):
      raise ValueError('Nesterov momentum requires a momentum and zero dampening')
    assert dampening == 0, 'not implemented'
    defaults = dict(lr=lr, momentum=momentum, dampening=dampening,
                    weight_decay=weight_decay, nesterov=nesterov)
    super().__init__(params, defaults)

  @torch.no_grad()
  def step(self, lr, closure=None):
    loss = None
    if closure is not None:
      with torch.enable_grad():
        loss = closure()
    for group in self.param_groups:
      params = [p for p in group['params'] if p.grad is not None]
      if not params:
        continue
      grads = [p.grad for p in params]
      wd, mom, scale = group['weight_decay'], group['momentum'], group['lr'] * lr
      if wd != 0:
        grads = torch._foreach_add(grads, params, alpha=wd)
      if mom != 0:
        bufs = []
        for p in params:
          st = self.state[p]
          if 'momentum_buffer' not in st:
            st['momentum_buffer'] = torch.zeros_like(p)
          bufs.append(st['momentum_buffer'])
        torch._foreach_mul_(bufs, mom)
        torch._foreach_add_(bufs, grads, alpha=scale)
        if group['nesterov']:
          upd = torch._foreach_add(grads, bufs, alpha=mom)
        else:
          upd = bufs
        torch._foreach_sub_(params, upd)
      else:
        torch._foreach_sub_(params, grads)
    return loss
