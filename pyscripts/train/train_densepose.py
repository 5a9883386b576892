#!/usr/bin/env python3
"""DensePose training entry point (`pyscripts/train/train_densepose.py` of twke18/SPML): the same
loop as train.py with the DensePose embedding network (colour + location local features) and
predictor (tags propagated from the nearest labelled segment) bound -- the reference's script
differs from train.py only in those imports (train_densepose.py:27-29) and in its dataset class.

  python3 pyscripts/train/train_densepose.py --snapshot_dir S --cfg_path C.yaml [--data_list synthetic]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train as _train  # noqa: E402


def main(argv=None):
  _train.main(argv, default_recipe='densepose',
              description='Training for pixel-wise embeddings for DensePose.')


if __name__ == '__main__':
  main()
