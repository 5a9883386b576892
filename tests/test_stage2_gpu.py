"""Stage 2 and the full-resolution kNN inference on the GPU (SURVEY 8f rows N4 and N1 o N2) against fixtures
exec'd from the reference's own script lines (tools/gen_golden.py): train_classifier.py:139-169
(h02_classifier_step.npz) and inference.py:162-227 (n5_inference.npz)."""
import pytest
import torch

from conftest import load_golden
from oracle import spml_oracle as O
from spml_amd import inference
from tools_synth import check_h02_step, h02_batch, h02_config, h02_models, parameter_checksums

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('channels_last', [False, True])
def test_classifier_trainer_matches_reference_steps(channels_last):
  """Two ClassifierTrainer steps on the GPU (NCHW / library convolutions, and NHWC where the frozen network's
  res4 / res5 units run on the matrix-core kernels in eval mode): loss, accuracy, classifier parameters and BN
  running statistics after each SGD step at north_star's 1e-4; the embedding network does not move."""
  from spml_amd.train import ClassifierTrainer
  from step_helpers import to_gpu
  g = load_golden('h02_classifier_step')
  cfg = h02_config()
  emb, pred = h02_models(cfg)
  tr = ClassifierTrainer(cfg, DEV, channels_last=channels_last, models=(emb, pred))
  tr.curr_iter = g.iter0
  before = parameter_checksums(tr.embedding_model)[1].clone()
  for it in range(2):
    datas, targets = h02_batch(g, it)
    out = tr.step(*to_gpu(datas, targets, channels_last))
    assert abs(out['lr'] - g['s%d_lr' % it]) < 1e-12
    check_h02_step(g, it, out, tr.prediction_model, 1e-4)
  assert torch.equal(parameter_checksums(tr.embedding_model)[1], before)


@pytest.mark.parametrize('ci', [0, 1])
def test_full_resolution_knn_inference_matches_reference_lines(ci):
  """`inference.predict_full_resolution` (sliding window -> k-means at full resolution -> Segsort.predictions
  against a memory bank -> label map) against inference.py:162-227.  End to end the cluster map is bounded
  by k-means near ties on a GPU convolution's output (compared statistically, like n2_window); GIVEN the
  reference's own clustering the retrieval + vote + scatter chain is compared label by label."""
  from spml_amd.models.predictions.segsort import segsort
  from spml_amd.train import voc12_scribble_config
  from test_inference_gpu import TinyEmbedder
  from test_oracle_golden import n5_case
  g = load_golden('n5_inference')
  t, conv, valid, crop, stride, k = n5_case(g, ci)
  model = TinyEmbedder(conv.out_channels, list(k)).to(DEV)
  model.conv.load_state_dict({k_: v.to(DEV) for k_, v in conv.state_dict().items()})
  predictor = segsort(voc12_scribble_config()).to(DEV).eval()
  bank, bank_lab = inference.drop_ignored_memory(g[t + 'bank'].to(DEV), g[t + 'bank_lab'].to(DEV))
  assert bank.shape == g[t + 'bank'].shape                       # (no ignore-class prototype in this bank)
  out = inference.predict_full_resolution(model, predictor, g[t + 'image'].to(DEV), valid, crop, stride,
                                          bank, bank_lab)
  want = g[t + 'semantic_prediction'].long()
  assert out['semantic_prediction'].shape == want.shape
  agree_c = (out['cluster_index'].cpu() == g[t + 'cluster_index'].long()).float().mean().item()
  agree_p = (out['semantic_prediction'].cpu() == want).float().mean().item()
  assert agree_c > 0.97 and agree_p > 0.97, (agree_c, agree_p)
  # the chain behind the clustering, on the reference's own segment ids
  emb = inference.embed_full_resolution(model, g[t + 'image'].to(DEV), crop, stride)
  cl_emb = O.normalize_embedding(emb.cpu().permute(0, 2, 3, 1).contiguous())[0, :valid[0], :valid[1]]
  pred, topk = predictor.predictions(
      {'cluster_embedding': cl_emb.reshape(valid[0] * valid[1], -1).to(DEV), 'cluster_index': g[t + 'cluster_index'].long().to(DEV)},
      {'semantic_memory_prototype': bank, 'semantic_memory_prototype_label': bank_lab})
  assert (pred.cpu().view(valid) == want).float().mean().item() >= 0.995
  # (a retrieval list may swap two equally similar bank entries; the labels it holds are the reference's)
  same_rows = (topk.cpu()[::7].sort(1).values == g[t + 'semantic_topk'].long().sort(1).values).all(1).float().mean().item()
  assert same_rows >= 0.99, same_rows


def test_drop_ignored_memory():
  protos = torch.randn(7, 4, device=DEV)
  labels = torch.tensor([3, 255, 0, 255, 1, 2, 255], device=DEV)
  p, l = inference.drop_ignored_memory(protos, labels)
  assert torch.equal(l.cpu(), torch.tensor([3, 0, 1, 2])) and torch.equal(p, protos[[0, 2, 4, 5]])
