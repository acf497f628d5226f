"""The reference's experiment entry points (_main_training_mnist / _svhn / _cifar10, Training/Train_goodGAN.py:472-725)
run end to end on the GPU for a shortened epoch: D/G/C (or pre-training) iterations through hipGraphs, validation with
train=False, sample grid PNG."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["mnist", "cifar10"])
def test_entry_point_runs_one_short_epoch(which, tmp_path, monkeypatch):
    from tg import runtime
    from Training import Train_goodGAN as TG
    runtime.set_context(None)

    class Flags(object):      # _customize_config(tmp_config, FLAGS) takes an argparse-like object (:707-720)
        train_size = 300 + (100 if which == "mnist" else 4000)
        sample_dir = str(tmp_path / "samples")
        seed = 1
    fn = {"mnist": TG._main_training_mnist, "cifar10": TG._main_training_cifar10}[which]
    monkeypatch.setattr(TG, "_root_dir", lambda: str(tmp_path))
    hist = fn(Flags(), epochs=1)
    assert len(hist) == 1 and np.isfinite([hist[0]['d_loss'], hist[0]['g_loss'], hist[0]['c_loss']]).all()
    assert 0.0 <= hist[0]['val_accuracy'] <= 1.0 and hist[0]['images_per_sec'] > 0
    pngs = [f for _, _, fs in os.walk(str(tmp_path)) for f in fs if f.endswith('.png')]
    assert pngs == ['train_01.png']


def test_training_learns_the_synthetic_task(tmp_path, monkeypatch):
    """End-to-end sanity beyond parity: 150 D+G+C iterations of the CIFAR-10 experiment on the synthetic class-prototype data
     take the classifier from chance to > 90 % validation accuracy, with finite adversarial losses."""
    from tg import runtime
    from Training import Train_goodGAN as TG
    runtime.set_context(None)

    class Flags(object):
        train_size = 4000 + 100 * 150
        sample_dir = None
        seed = 1
        summary = False
    monkeypatch.setattr(TG, "_root_dir", lambda: str(tmp_path))
    hist = TG._main_training_cifar10(Flags(), epochs=1)
    assert hist[0]['val_accuracy'] > 0.9, hist
    assert hist[0]['c_loss'] < 1.0 and np.isfinite([hist[0]['d_loss'], hist[0]['g_loss']]).all()
