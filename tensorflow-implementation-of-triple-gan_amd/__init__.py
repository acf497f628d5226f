"""MI355X-native Triple-GAN training step (hot path of
Wenyuan-Vincent-Li/Tensorflow-Implementation-of-Triple-GAN) — see DESIGN.md.

The directory is laid out like the reference's repository root (Model/, Training/,
config.py) so that `from Model import nn`, `from Training.Train_goodGAN import Train`
read exactly as they do there (the reference appends its root to sys.path,
Training/Train_goodGAN.py:8-12); importing this package does the same.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
