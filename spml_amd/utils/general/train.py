"""Learning-rate schedules of `spml/utils/general/train.py` (pure math)."""


def lr_poly(base_lr, curr_iter, max_iter, warmup_iter=0, power=0.9):
  """Polynomial decay with linear warm-up from base_lr/10 (general/train.py:8-28)."""
  decayed = base_lr * ((1 - float(curr_iter) / max_iter) ** power)
  if curr_iter < warmup_iter:
    alpha = curr_iter / warmup_iter
    return min(base_lr * (0.1 * (1 - alpha) + alpha), decayed)
  return decayed


def get_step_index(curr_iter, decay_iters):
  """Number of decay milestones already passed (general/train.py:31-37)."""
  for idx, milestone in enumerate(decay_iters):
    if curr_iter < milestone:
      return idx
  return len(decay_iters)


def lr_step(base_lr, curr_iter, decay_iters, warmup_iter=0):
  """Step decay (x0.1 per milestone) with linear warm-up (general/train.py:40-57)."""
  if curr_iter < warmup_iter:
    alpha = curr_iter / warmup_iter
    return base_lr * (0.1 * (1 - alpha) + alpha)
  return base_lr * (0.1 ** get_step_index(curr_iter, decay_iters))
