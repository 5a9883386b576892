"""Stage 2: `spml_amd.train.ClassifierTrainer` -- the hot loop of
pyscripts/train/train_classifier.py:33-185 -- against two steps of the reference's own loop body
(tests/golden/h02_classifier_step.npz, tools/gen_golden.py).  The stage has no HIP-only operator (frozen
backbone forward, 3x3 conv + BN + ReLU + 1x1 conv, bilinear up-sampling, cross-entropy), so the same
class also runs on CPU tensors: the host logic is pinned here without a GPU, the GPU run in
tests/test_stage2_gpu.py."""
import pytest
import torch

from conftest import load_golden
from tools_synth import check_h02_step, h02_batch, h02_config, h02_models, parameter_checksums


def test_classifier_trainer_matches_reference_steps_on_cpu():
  from spml_amd.train import ClassifierTrainer
  g = load_golden('h02_classifier_step')
  cfg = h02_config()
  emb, pred = h02_models(cfg)
  tr = ClassifierTrainer(cfg, 'cpu', models=(emb, pred))
  tr.curr_iter = g.iter0
  before = parameter_checksums(emb)[1]
  for it in range(2):
    datas, targets = h02_batch(g, it)
    out = tr.step(datas, targets)
    assert abs(out['lr'] - g['s%d_lr' % it]) < 1e-12
    check_h02_step(g, it, out, tr.prediction_model, 2e-6)
  assert torch.equal(parameter_checksums(emb)[1], before)          # frozen: weight decay included
  assert not tr.embedding_model.training and tr.prediction_model.training
  assert tr.curr_iter == g.iter0 + 2
  state = tr.state_dict()
  assert sorted(state) == ['embedding_model', 'iteration', 'optimizer', 'prediction_model']


def test_classifier_trainer_refuses_what_the_reference_refuses():
  from spml_amd.train import ClassifierTrainer
  cfg = h02_config()
  cfg.network.prediction_types = 'segsort'
  with pytest.raises(ValueError, match='Not support segsort'):            # train_classifier.py:86-87
    ClassifierTrainer(cfg, 'cpu')
  cfg = h02_config()
  cfg.network.backbone_types = 'panoptic_deeplab_50'
  with pytest.raises(ValueError, match='Not support panoptic_deeplab_50'):   # :81-82
    ClassifierTrainer(cfg, 'cpu')
  cfg = h02_config()
  tr = ClassifierTrainer(cfg, 'cpu', models=h02_models(cfg))
  with pytest.raises(ValueError, match='Pre-trained model is required'):    # :103-104
    tr.load_pretrained({'prediction_model': {}})
  # stage 1's snapshot: only its embedding model is read
  emb2, _ = h02_models(cfg)
  with torch.no_grad():
    for p in emb2.parameters():
      p.add_(0.5)
  tr.load_pretrained({'embedding_model': emb2.state_dict(), 'prediction_model': {'junk': torch.zeros(1)}})
  assert torch.equal(parameter_checksums(tr.embedding_model)[1], parameter_checksums(emb2)[1])
