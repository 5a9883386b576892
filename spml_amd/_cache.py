"""Small bounded cache for shape-keyed device constants (grid ids, pooling / interpolation matrices).

The reference's functions are pure (re-entrant, no module-level mutable state); the constants they
rebuild on every call (linspace grids, `unique` of a grid) are worth keeping per shape, but not without bound:
variable crop sizes or multi-scale evaluation would pin one entry per shape for the life of the process.  At most
`maxsize` entries, least recently used out first, guarded by a lock (one Python thread per GPU is allowed)."""
import collections
import threading


class BoundedCache(object):

  def __init__(self, maxsize=8):
    self._maxsize = int(maxsize)
    self._items = collections.OrderedDict()
    self._lock = threading.Lock()

  def get_or_make(self, key, make):
    with self._lock:
      hit = self._items.get(key)
      if hit is not None:
        self._items.move_to_end(key)
        return hit
    value = make()                      # (outside the lock: may launch kernels / synchronise)
    with self._lock:
      self._items[key] = value
      self._items.move_to_end(key)
      while len(self._items) > self._maxsize:
        self._items.popitem(last=False)
    return value

  def __len__(self):
    return len(self._items)
