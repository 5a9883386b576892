"""Global experiment configuration (mirror of `spml/config/default.py`): same
keys and defaults, attribute access without easydict, `yaml.safe_load`."""
import numpy as np
import yaml


class ConfigNode(dict):
  """dict with attribute access (the slice of easydict the reference uses)."""

  def __getattr__(self, key):
    try:
      return self[key]
    except KeyError:
      raise AttributeError(key)

  def __setattr__(self, key, value):
    self[key] = value

  @classmethod
  def wrap(cls, obj):
    if isinstance(obj, dict):
      return cls({k: cls.wrap(v) for k, v in obj.items()})
    return obj


def _defaults():
  c = ConfigNode()
  c.embedding_model = ''
  c.prediction_model = ''
  c.gpus = ''
  c.num_threads = 4
  c.network = ConfigNode(
      pixel_means=np.array((0.485, 0.456, 0.406)), pixel_stds=np.array((0.229, 0.224, 0.225)),
      pretrained='', use_syncbn=False, backbone_types='', prediction_types='',
      aspp_feature_dim=512, pspp_feature_dim=512, embedding_dim=128, label_divisor=255,
      kmeans_iterations=10, kmeans_num_clusters=[5, 5])
  c.dataset = ConfigNode(data_dir='', train_data_list='', test_data_list='', color_map_path='',
                         num_classes=0, semantic_ignore_index=255)
  c.train = ConfigNode(
      lr_policy='step', random_mirror=True, random_scale=True, random_crop=True, shuffle=True,
      resume=False, begin_iteration=0, max_iteration=0, warmup_iteration=0, decay_iterations=[0],
      snapshot_step=0, tensorboard_step=0, base_lr=1e-3, weight_decay=5e-3, momentum=0.9,
      batch_size=0, crop_size=[0, 0], memory_bank_size=0,
      sem_ann_loss_types='none', sem_occ_loss_types='none', img_sim_loss_types='none',
      feat_aff_loss_types='none', sem_ann_concentration=0, sem_occ_concentration=0,
      img_sim_concentration=0, feat_aff_concentration=0, sem_ann_loss_weight=0.0,
      sem_occ_loss_weight=0.0, img_sim_loss_weight=0.0, feat_aff_loss_weight=0.0)
  c.test = ConfigNode(scales=[0], image_size=0, crop_size=[0, 0], stride=[0, 0])
  return c


config = _defaults()


def update_config_from_dict(exp_config, target=None):
  """One-level-deep merge (default.py:88-103); base_lr / weight_decay -> float."""
  target = config if target is None else target
  for k, v in exp_config.items():
    if k in target and isinstance(v, dict):
      if k == 'train':
        for key in ('base_lr', 'weight_decay'):
          if key in v:
            v[key] = float(v[key])
      for vk, vv in v.items():
        target[k][vk] = vv
    else:
      target[k] = ConfigNode.wrap(v)
  return target


def update_config(config_file):
  with open(config_file) as f:
    update_config_from_dict(yaml.safe_load(f))


def make_config(**sections):
  """A fresh config (not the global one) with the given section overrides."""
  return update_config_from_dict(sections, _defaults())
