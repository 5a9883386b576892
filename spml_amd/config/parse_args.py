"""CLI of the reference's entry points (mirror of `spml/config/parse_args.py`)."""
import argparse

from spml_amd.config.default import config, update_config


def build_parser(description=''):
  p = argparse.ArgumentParser(description=description)
  p.add_argument('--snapshot_dir', required=True, type=str, help='/path/to/snapshot/dir.')
  p.add_argument('--save_dir', type=str, help='/path/to/save/dir.')
  p.add_argument('--cfg_path', required=True, type=str, help='/path/to/specific/config/file.')
  p.add_argument('--semantic_memory_dir', type=str, default=None)
  p.add_argument('--cam_dir', type=str, default=None)
  p.add_argument('--data_dir', type=str, default=None)
  p.add_argument('--data_list', type=str, default=None)
  p.add_argument('--kmeans_num_clusters', type=str, help='H,W')
  p.add_argument('--label_divisor', type=int)
  # not in the reference's CLI (it has one script per recipe and real data lists): which recipe the
  # training entry point binds, and the kind of supervision the synthetic batches imitate
  p.add_argument('--recipe', type=str, default=None, choices=['voc', 'densepose'])
  p.add_argument('--supervision', type=str, default='scribble', choices=['scribble', 'tag'])
  for name, default in (('crf_iter_max', 10), ('crf_pos_xy_std', 1), ('crf_pos_w', 3),
                        ('crf_bi_xy_std', 67), ('crf_bi_w', 4), ('crf_bi_rgb_std', 3)):
    p.add_argument('--' + name, type=int, default=default)
  return p


def parse_args(description='', argv=None):
  """parse_known_args -> update_config(cfg_path) -> parse_args (parse_args.py:46-53)."""
  parser = build_parser(description)
  args, _ = parser.parse_known_args(argv)
  update_config(args.cfg_path)
  return parser.parse_args(argv)
