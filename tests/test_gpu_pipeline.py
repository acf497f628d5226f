"""Input pipeline on the GPU: the device tail (tg_u8_affine_f32 / tg_onehot_i32_f32) is bit-identical to the reference's
float32 expression evaluated by NumPy, and Train.train runs end to end from TFRecord files in the reference's layout."""
import os

import numpy as np
import pytest

from oracle import tfrecord as O
import gpu_common as G

pytestmark = pytest.mark.gpu


def _files(tmp_path, Dataset, cfg, n_lab, n_unl, n_test):
    d = tmp_path / 'Tfrecord'
    d.mkdir()
    rng = np.random.default_rng(0)
    proto = rng.integers(0, 256, (10, 32, 32, 3))
    Dataset.TRAIN_SIZE = n_lab + n_unl
    tr = Dataset(str(tmp_path), cfg, n_lab, 'train', True)
    te = Dataset(str(tmp_path), cfg, n_lab, 'test', False)
    for name, n in zip(tr.get_filenames() + te.get_filenames(), (n_lab, n_unl, n_test)):
        lab = rng.integers(0, 10, n)
        img = np.clip(proto[lab] + rng.normal(0, 30, (n, 32, 32, 3)), 0, 255).astype(np.uint8)
        O.write_tfrecord(name, img, lab)


def test_device_tail_is_bit_identical_to_the_host_expression(tmp_path):
    import torch
    from tg import lib
    lib.load()
    from Input_Pipeline.cifar10Dataset import cifar10Dataset
    from Input_Pipeline.mnistDataset import mnistDataset
    cfg = G.make_config(dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6))
    G.fresh_trainer(cfg)                                            # creates the Context the tail launches on
    u8 = np.arange(256, dtype=np.uint8).repeat(12).reshape(4, 16, 16, 3)
    lab = np.array([3, 0, 9, 3], np.int32)
    for Dataset in (cifar10Dataset, mnistDataset):
        ds = Dataset('/nonexistent', cfg, 10, 'train')
        xd, yd = ds._to_device(u8, lab)
        xh, yh = ds._to_host(u8, lab)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(xd.numpy(), xh)
        np.testing.assert_array_equal(yd.numpy(), yh)
    assert xh.min() == 0.0 and xh.max() == 1.0                       # MNIST range; CIFAR range checked above through equality


def test_train_from_tfrecords_end_to_end(tmp_path):
    import torch
    from tg import runtime
    from Training.Train_goodGAN import Train
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10
    from Input_Pipeline.cifar10Dataset import cifar10Dataset
    sizes = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)
    cfg = G.make_config(sizes, DATA_DIR=str(tmp_path), NUM_LABEL=40, TRAIN_SIZE=8 * 6, EPOCHS=2, SAMPLE_DIR=None, USE_HIP_GRAPH=True, REPEAT=-1)
    train_size = cifar10Dataset.TRAIN_SIZE
    try:
        _files(tmp_path, cifar10Dataset, cfg, 40, 120, 24)
        runtime.set_context(None)
        torch.cuda.empty_cache()
        tr = Train(cfg, None, None)
        hist = tr.train(cifar10Dataset, Good_GAN_cifar10, None)
    finally:
        cifar10Dataset.TRAIN_SIZE = train_size
    assert len(hist) == 2
    for rec in hist:
        assert all(np.isfinite(rec[k]) for k in ('d_loss', 'g_loss', 'c_loss')) and 0.0 <= rec['val_accuracy'] <= 1.0
    assert tr.iteration == 2 * 6
