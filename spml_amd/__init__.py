"""spml_amd: MI355X-native hot path of SPML (pixel-to-segment contrastive
learning) -- hand-written gfx950 HIP kernels behind the reference's own
`spml.utils.segsort` / `spml.models` Python API.

`spml_amd.install_as_spml()` registers this package under the name `spml` so
that code written against the reference (`import spml.utils.segsort.common`)
resolves to the MI355X implementation unchanged."""
import os
import sys

__version__ = '0.2.0'

_HERE = os.path.dirname(os.path.abspath(__file__))


def use_tuned_miopen_db(force=False):
  """Point MIOpen at the solver-search results shipped with this package.

  The ROCm image has no gfx950 find-db, so PyTorch's immediate mode picks im2col + GEMM /
  layout-transposing fallbacks for the fp32 convolutions of ResNet-101 (290 ms per training
  step at batch 16, 513x513).  `tools/miopen_tune.py` ran MIOpen's search once for every
  conv shape of the training step (forward, backward-data, backward-weights; 18 minutes on
  one MI355X) with MIOPEN_USER_DB_PATH inside the repository: `spml_amd/miopen_db/` holds
  the resulting user find-db / perf-db (text) and the compiled kernels of the winners
  (`cache/`).  With them a fresh process reaches 248 ms per step without searching.
  Environment variables already set by the user win unless `force`.  Must run before the
  first convolution (MIOpen reads the variables when it initialises); importing spml_amd
  does it."""
  db = os.path.join(_HERE, 'miopen_db')
  if not os.path.isdir(db):
    return False
  # MIOpen opens the user db and the kernel cache read-write: every process works on a private
  # copy (1.7 MB) in a fresh mkdtemp() directory, so the tracked files are never modified, a
  # read-only install works, and the ranks of a node do not contend for one sqlite file.
  # `tools/miopen_tune.py` sets MIOPEN_USER_DB_PATH itself (user-set variables win).
  if 'MIOPEN_USER_DB_PATH' not in os.environ or force:
    try:
      import atexit
      import shutil
      import tempfile
      private = tempfile.mkdtemp(prefix='spml_miopen_db_')
      shutil.copytree(db, private, dirs_exist_ok=True)
      atexit.register(shutil.rmtree, private, True)
      db = private
    except Exception:                    # no writable temp dir: fall back to the shipped copy
      pass
  for key, val in (('MIOPEN_USER_DB_PATH', db),
                   ('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(db, 'cache'))):
    if force or key not in os.environ:
      os.environ[key] = val
  return True


use_tuned_miopen_db()


def install_as_spml():
  """Alias spml_amd (and its sub-packages) as `spml` in sys.modules."""
  import importlib
  pkg = sys.modules[__name__]
  sys.modules.setdefault('spml', pkg)
  for sub in ('utils', 'utils.general', 'utils.general.common', 'utils.general.train',
              'utils.segsort', 'utils.segsort.common', 'utils.segsort.loss',
              'utils.segsort.eval', 'models', 'models.utils', 'config',
              'config.default', 'config.parse_args',
              'models.backbones', 'models.backbones.resnet', 'models.heads',
              'models.heads.spp', 'models.embeddings', 'models.embeddings.base_model',
              'models.embeddings.local_model', 'models.embeddings.resnet_deeplab',
              'models.embeddings.resnet_pspnet', 'models.embeddings.resnet_pspnet_densepose',
              'models.predictions', 'models.predictions.segsort',
              'models.predictions.segsort_softmax', 'models.predictions.segsort_softmax_densepose',
              'models.predictions.softmax_classifier', 'utils.segsort.others'):
    try:
      mod = importlib.import_module(__name__ + '.' + sub)
    except ImportError:
      continue
    sys.modules.setdefault('spml.' + sub, mod)
  return pkg
