"""Checkpoint / resume on the GPU: a run that is saved, torn down and restored continues bit-identically to
the uninterrupted run (weights, Adam slots, running statistics, EMA shadows, Philox stream); Train.train writes the
reference's artefacts (Run_ directory with model_<epoch>.ckpt.*, event files, sample grid) and resumes from them."""
import os

import numpy as np
import pytest

import gpu_common as G

pytestmark = pytest.mark.gpu
SIZES = dict(B_G=8, L_C=4, U_C=4, L_D=2, U_D=6)


def _state(tr):
    out = {}
    for k, st in tr.cx.stores.items():
        for buf in ('p', 'm', 'v', 's', 'step'):
            out[k + '/' + buf] = getattr(st, buf).detach().cpu().numpy().copy()
        if st.ema is not None:
            out[k + '/ema'] = st.ema.detach().cpu().numpy().copy()
    return out


def test_resume_continues_bit_identically(tmp_path):
    import torch
    from oracle import step_cifar10 as S
    from Training.Saver import Saver
    feeds = [S.synth_batch(40 + i, dict(S.SIZES, **SIZES)) for i in range(4)]

    def run(tr, its):
        for i in its:
            tr.feed(feeds[i])
            tr.sample_latent()
            tr.train_iteration()
        torch.cuda.synchronize()

    a = G.fresh_trainer(G.make_config(SIZES, SEED=9, USE_HIP_GRAPH=True))
    a.set_hyper(lambda_1=0.3, lambda_2=0.5)
    run(a, [0, 1])
    saver = Saver(str(tmp_path))
    saver.set_save_path(comments='resume test')
    saver.save(a, 'model_0002.ckpt')
    run(a, [2, 3])
    want, want_losses = _state(a), a.losses()

    b = G.fresh_trainer(G.make_config(SIZES, SEED=1234, USE_HIP_GRAPH=True))           # different seed: everything must come from the file
    b.set_hyper(lambda_1=0.3, lambda_2=0.5)
    assert Saver(str(tmp_path)).restore(b) == 2
    run(b, [2, 3])
    got = _state(b)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert b.losses() == want_losses


def test_train_writes_and_resumes_from_the_reference_artefacts(tmp_path):
    import torch
    from tg import runtime
    from Training.Train_goodGAN import Train
    from Model.Good_GAN_cifar10 import Good_GAN_cifar10
    from Input_Pipeline.syntheticDataset import syntheticDataset
    from oracle import tfrecord as O
    save, log, smp = str(tmp_path / 'Weight'), str(tmp_path / 'Log'), str(tmp_path / 'Samples')
    os.makedirs(save)
    kw = dict(TRAIN_SIZE=8 * 3, EPOCHS=2, SAMPLE_DIR=smp, SAMPLE_SIZE=16, USE_HIP_GRAPH=True, SUMMARY=True, SAVE_PER_EPOCH=1, NUM_LABEL=40)
    sample_y = np.eye(10, dtype=np.float32)[np.arange(16) % 10]

    def fresh(**over):
        runtime.set_context(None)
        torch.cuda.empty_cache()
        return Train(G.make_config(SIZES, **dict(kw, **over)), log, save, comments='e2e')

    tr = fresh()
    hist = tr.train(syntheticDataset, Good_GAN_cifar10, sample_y)
    assert [h['epoch'] for h in hist] == [1, 2]
    runs = [d for d in os.listdir(save) if d.startswith('Run_')]
    assert len(runs) == 1
    assert sorted(os.listdir(os.path.join(save, runs[0]))) == ['Comments.txt', 'model_0001.ckpt.npz', 'model_0002.ckpt.npz']
    assert sorted(os.listdir(smp)) == ['train_01.png', 'train_02.png']
    for kind, tags in (('train', {'g_loss', 'd_loss', 'c_loss'}), ('val', {'val_accuracy'})):
        rd = os.path.join(log, kind)
        rdir = os.path.join(rd, os.listdir(rd)[0])
        ev = [f for f in os.listdir(rdir) if f.startswith('events.out.tfevents.')]
        assert len(ev) == 1 and len(O.read_tfrecord(os.path.join(rdir, ev[0]))) == 3     # file_version + one event per epoch
        assert set(open(os.path.join(rdir, 'history.csv')).readline().strip().split(',')[1:]) == tags
    # resume: RESTORE picks the latest run / epoch, numbering continues (Train_goodGAN.py:140-147)
    tr2 = fresh(RESTORE=True, EPOCHS=1)
    hist2 = tr2.train(syntheticDataset, Good_GAN_cifar10, sample_y)
    assert [h['epoch'] for h in hist2] == [3]
    assert 'model_0003.ckpt.npz' in os.listdir(os.path.join(save, runs[0]))
    assert int(tr2.cx.stores['classifier'].step.item()) == 3 * 3                       # 3 iterations per epoch, 3 epochs


@pytest.mark.parametrize("data", ['cifar10', 'mnist', 'svhn'])
def test_checkpoint_holds_exactly_the_variables_tensorflow_would_save(tmp_path, data):
    """the key set of a checkpoint equals the names tf.train.Saver() would write for the reference's graph — derived
    independently of the package's own variable tables from the reference's scopes (oracle/tf_names.py) — with two documented
    substitutions: TensorFlow's six beta1_power / beta2_power scalars are replaced by one step count per optimiser (tg/adam_step/<net>),
    and tg/rng_state, tg/epoch have no TensorFlow counterpart."""
    from oracle import tf_names
    from Training.Saver import Saver
    if data == 'cifar10':
        tr = G.fresh_trainer(G.make_config(SIZES))
    else:
        from Model.Good_GAN import Good_GAN
        tr = G.fresh_trainer(G.make_config_goodgan(data, SIZES), None, Good_GAN)
    saver = Saver(str(tmp_path))
    saver.set_save_path(comments='names')
    keys = set(np.load(saver.save(tr, 'model_0001.ckpt')).files)
    extras = {k for k in keys if k.startswith('tg/')}
    assert extras == {'tg/adam_step/good_generator', 'tg/adam_step/discriminator', 'tg/adam_step/classifier', 'tg/rng_state', 'tg/epoch'}
    want = set(tf_names.saver_variables(data)) - set(tf_names.NON_SLOT)
    assert keys - extras == want, (sorted((keys - extras) - want)[:5], sorted(want - (keys - extras))[:5])
    for name in ('classifier/conv1_1/V', 'classifier/NiN1/NiN1/V', 'good_generator/gg_h0_lin/gg_h0_lin/kernel') if data == 'cifar10' else ():
        assert {name, name + '/Adam_optimizer', name + '/Adam_optimizer_1'} <= keys               # the names SURVEY §8f spells out
    assert 'classifier/NiN1/NiN1/V/ExponentialMovingAverage' in keys or data != 'cifar10'


def test_restored_checkpoint_reproduces_the_golden_evaluation(tmp_path):
    """save -> load into a differently initialised process state -> the restored model's evaluation logits, accuracy and sampler output are
    the committed golden vectors (outputs of the oracle, tests/golden/cifar10_small_k10.npz), not merely what the saving run computed."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_golden as M
    from oracle import step_cifar10 as S
    from tg.runtime import InjectedRNG
    from Training.Saver import Saver
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cifar10_small_k10.npz'))
    a = G.fresh_trainer(G.make_config(M.SIZES), S.init_params(0))
    saver = Saver(str(tmp_path))
    saver.set_save_path(comments='golden')
    saver.save(a, 'model_0007.ckpt')
    b = G.fresh_trainer(G.make_config(M.SIZES, SEED=4321), S.init_params(77))                     # other weights everywhere
    cx = b.cx
    xt, yt, noise = M.test_split()
    z, y = M.sample_latents()

    def logits_and_acc():
        cx.rng = InjectedRNG({'val/C/noise': noise}, cx.device)
        with cx.phase_scope('val', record=False):
            with cx.rng_scoped('val/C'):
                lg, _ = b.model.classifier(b.model.zca().apply(cx.from_numpy(xt)), False)
        cx.rng = InjectedRNG({'val/C/noise': noise}, cx.device)
        return lg.numpy(), b.evaluate([(xt, yt)])

    before, _ = logits_and_acc()
    assert G.rel_err(before, g['logits_init']) > 1e-2                                            # b really starts elsewhere
    assert Saver(str(tmp_path)).restore(b) == 7
    after, acc = logits_and_acc()
    assert G.rel_err(after, g['logits_init']) < 2e-4
    assert abs(acc - float(g['acc_init'])) <= 1.0 / M.N_TEST + 1e-9
    assert G.rel_err(b.sample(z, y), g['sample_init']) < 2e-4
