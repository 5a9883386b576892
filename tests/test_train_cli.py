"""The training entry point (pyscripts/train/train.py) keeps the reference's command line,
YAML keys and snapshot files."""
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

YAML = """
gpus: "0"
num_threads: 4
dataset:
  num_classes: 21
  semantic_ignore_index: 255
  dataset: VOC2012
  data_dir: ""
  train_data_list: ""
  test_data_list: ""
  color_map_path: "misc/colormapvoc.mat"
network:
  pretrained: ""
  embedding_dim: 32
  label_divisor: 2048
  use_syncbn: false
  kmeans_iterations: 3
  kmeans_num_clusters:
    - 4
    - 4
  backbone_types: panoptic_deeplab_50
  prediction_types: segsort
train:
  resume: false
  lr_policy: poly
  begin_iteration: 0
  snapshot_step: 2
  tensorboard_step: 100
  max_iteration: 2
  random_mirror: true
  random_scale: true
  random_crop: true
  warmup_iteration: 0
  base_lr: 3e-3
  weight_decay: 5e-4
  momentum: 0.9
  batch_size: 2
  crop_size:
    - 97
    - 97
  memory_bank_size: 2
  sem_ann_concentration: 6
  sem_occ_concentration: 12
  img_sim_concentration: 16
  feat_aff_concentration: 0
  sem_ann_loss_types: segsort
  sem_occ_loss_types: segsort
  img_sim_loss_types: segsort
  feat_aff_loss_types: none
  sem_ann_loss_weight: 1.0
  sem_occ_loss_weight: 0.5
  img_sim_loss_weight: 0.1
  feat_aff_loss_weight: 0.0
test:
  scales:
    - 1
  image_size: 97
  crop_size:
    - 97
    - 97
  stride:
    - 97
    - 97
"""


def load_cli():
  spec = importlib.util.spec_from_file_location('spml_train_cli', os.path.join(ROOT, 'pyscripts', 'train', 'train.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_cli_parses_reference_arguments_and_config(tmp_path):
  from spml_amd.config.default import config
  from spml_amd.config.parse_args import parse_args
  cfg = tmp_path / 'config_emb.yaml'
  cfg.write_text(YAML)
  args = parse_args('x', ['--snapshot_dir', str(tmp_path / 's'), '--cfg_path', str(cfg), '--data_dir', 'd',
                          '--data_list', 'synthetic', '--kmeans_num_clusters', '4,4'])
  assert args.snapshot_dir.endswith('s') and args.data_list == 'synthetic'
  assert config.network.backbone_types == 'panoptic_deeplab_50'
  assert config.train.base_lr == 3e-3 and isinstance(config.train.weight_decay, float)
  assert config.train.crop_size == [97, 97] and config.train.memory_bank_size == 2
  if not torch.cuda.is_available():
    with pytest.raises(SystemExit):          # no CPU fallback
      load_cli().main(['--snapshot_dir', str(tmp_path / 's'), '--cfg_path', str(cfg)])


@pytest.mark.gpu
def test_cli_trains_and_writes_reference_snapshot_files(tmp_path):
  cfg = tmp_path / 'config_emb.yaml'
  cfg.write_text(YAML)
  snap = tmp_path / 'stage1'
  load_cli().main(['--snapshot_dir', str(snap), '--cfg_path', str(cfg), '--data_list', 'synthetic'])
  model = torch.load(str(snap / 'model-1.pth'), map_location='cpu')
  assert sorted(model.keys()) == ['embedding_model', 'prediction_model']
  assert any(k.startswith('resnet_backbone.') for k in model['embedding_model'])
  opt = torch.load(str(snap / 'model-1.state.pth'), map_location='cpu')
  assert 'state' in opt and 'param_groups' in opt


@pytest.mark.gpu
def test_cli_resume_restores_memory_bank_iteration_and_generators(tmp_path):
  """`model-{iter}.state.pth` keeps the reference's optimizer state dict and, next to it, the memory
  bank, the iteration counter and the generator states: a run resumed after iteration 1
  ends where the uninterrupted run ends."""
  cfg_a = tmp_path / 'a.yaml'
  cfg_a.write_text(YAML.replace('max_iteration: 2', 'max_iteration: 3').replace('snapshot_step: 2', 'snapshot_step: 1'))
  snap = tmp_path / 'run'
  load_cli().main(['--snapshot_dir', str(snap), '--cfg_path', str(cfg_a), '--data_list', 'synthetic'])
  want = torch.load(str(snap / 'model-2.pth'), map_location='cpu')
  st1 = torch.load(str(snap / 'model-1.state.pth'), map_location='cpu', weights_only=False)
  assert st1['spml_iteration'] == 2 and 'spml_rng' in st1
  banks = st1['spml_memory_banks']
  assert len(banks['memory_prototype']) == 2 and 'memory_prototype_semantic_tag' in banks
  os.rename(str(snap / 'model-2.pth'), str(snap / 'uninterrupted-2.pth'))
  cfg_b = tmp_path / 'b.yaml'
  cfg_b.write_text(YAML.replace('max_iteration: 2', 'max_iteration: 3').replace('snapshot_step: 2', 'snapshot_step: 1')
                   .replace('resume: false', 'resume: true').replace('begin_iteration: 0', 'begin_iteration: 1'))
  load_cli().main(['--snapshot_dir', str(snap), '--cfg_path', str(cfg_b), '--data_list', 'synthetic'])
  got = torch.load(str(snap / 'model-2.pth'), map_location='cpu')
  for part in ('embedding_model', 'prediction_model'):
    for k, v in want[part].items():
      if v.is_floating_point():
        # (fp32 atomics in the prototype sums make two runs differ in the last bits; three SGD steps on)
        torch.testing.assert_close(got[part][k], v, rtol=1e-3, atol=1e-5, msg=lambda m: '%s.%s: %s' % (part, k, m))


@pytest.mark.gpu
def test_densepose_entry_point_binds_the_densepose_modules(tmp_path):
  spec = importlib.util.spec_from_file_location(
      'spml_train_densepose_cli', os.path.join(ROOT, 'pyscripts', 'train', 'train_densepose.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  cfg = tmp_path / 'config_emb.yaml'          # (no recipe hint in the file name)
  cfg.write_text(YAML.replace('panoptic_deeplab_50', 'panoptic_pspnet_50').replace('num_classes: 21', 'num_classes: 15')
                 .replace('sem_occ_loss_types: segsort', 'sem_occ_loss_types: none')
                 .replace('memory_bank_size: 2', 'memory_bank_size: 0'))
  snap = tmp_path / 'dp'
  mod.main(['--snapshot_dir', str(snap), '--cfg_path', str(cfg), '--data_list', 'synthetic'])
  model = torch.load(str(snap / 'model-1.pth'), map_location='cpu')
  assert any(k.startswith('pspp.') for k in model['embedding_model'])
  assert any(k.startswith('lfn.') for k in model['embedding_model'])          # colour-smoothing kernel of the local features


@pytest.mark.gpu
def test_classifier_entry_point_trains_on_a_stage1_snapshot(tmp_path):
  """Stage 2 (pyscripts/train/train_classifier.py): stage 1's snapshot as `network.pretrained`, the reference's
  two snapshot files out; the embedding network in them is stage 1's, bit for bit (frozen)."""
  stage1 = tmp_path / 'config_emb.yaml'
  stage1.write_text(YAML.replace('panoptic_deeplab_50', 'panoptic_deeplab_101'))
  snap1 = tmp_path / 'stage1'
  load_cli().main(['--snapshot_dir', str(snap1), '--cfg_path', str(stage1), '--data_list', 'synthetic'])
  spec = importlib.util.spec_from_file_location(
      'spml_train_classifier_cli', os.path.join(ROOT, 'pyscripts', 'train', 'train_classifier.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  cfg = tmp_path / 'config_classifier.yaml'
  text = (YAML.replace('panoptic_deeplab_50', 'panoptic_deeplab_101').replace('prediction_types: segsort', 'prediction_types: softmax_classifier')
          .replace('kmeans_iterations: 3', 'kmeans_iterations: 0'))
  cfg.write_text(text)                                            # (pretrained: "" -> refused like the reference)
  with pytest.raises(ValueError, match='Pre-trained model is required'):
    mod.main(['--snapshot_dir', str(tmp_path / 'x'), '--cfg_path', str(cfg), '--data_list', 'synthetic'])
  cfg.write_text(text.replace('pretrained: ""', 'pretrained: "%s"' % str(snap1 / 'model-1.pth')))
  snap2 = tmp_path / 'softmax_classifier_stage1'
  mod.main(['--snapshot_dir', str(snap2), '--cfg_path', str(cfg), '--data_list', 'synthetic'])
  one = torch.load(str(snap1 / 'model-1.pth'), map_location='cpu')
  two = torch.load(str(snap2 / 'model-1.pth'), map_location='cpu')
  assert sorted(two.keys()) == ['embedding_model', 'prediction_model']
  for k, v in one['embedding_model'].items():
    assert torch.equal(two['embedding_model'][k], v), k
  assert any(k.startswith('semantic_classifier.') for k in two['prediction_model'])
  opt = torch.load(str(snap2 / 'model-1.state.pth'), map_location='cpu')
  assert 'state' in opt and 'param_groups' in opt
