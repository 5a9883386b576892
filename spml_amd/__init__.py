"""spml_amd: MI355X-native hot path of SPML (pixel-to-segment contrastive
learning) -- hand-written gfx950 HIP kernels behind the reference's own
`spml.utils.segsort` / `spml.models` Python API.

`spml_amd.install_as_spml()` registers this package under the name `spml` so
that code written against the reference (`import spml.utils.segsort.common`)
resolves to the MI355X implementation unchanged."""
import sys

__version__ = '0.1.0'


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
