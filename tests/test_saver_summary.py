"""Checkpoint / resume and scalar summaries: Training/Saver.py keeps the reference's directory protocol
(Run_<timestamp>/model_<epoch:04d>.ckpt.*, _findfilename) and round-trips every variable with its Adam slots, EMA shadow, step
counts and RNG state bit-exactly; Training/Summary.py writes TensorBoard event files (TFRecord-framed Event protos) that the
oracle's reader decodes.  Host only (ParamStore on the CPU device)."""
import os
import struct
import sys
import types

import numpy as np
import pytest
import torch

from oracle import tfrecord as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-implementation-of-triple-gan_amd'))


def fake_session(seed):
    from tg.runtime import ParamStore
    rng = np.random.default_rng(seed)
    specs = {'classifier': [('classifier/conv1_1/V', (3, 3, 3, 8), True), ('classifier/conv1_1/b', (8,), True),
                            ('classifier/conv1_1/meanOnlyBatchNormalization/pop_mean', (8,), False)],
             'good_generator': [('good_generator/gg_h0_lin/gg_h0_lin/kernel', (11, 5), True), ('good_generator/gg_bn0/moving_mean', (5,), False)]}
    stores = {}
    for net, sp in specs.items():
        st = ParamStore(net, sp, torch.device('cpu'))
        for buf in (st.p, st.m, st.v, st.s):
            buf.copy_(torch.from_numpy(rng.standard_normal(buf.numel()).astype(np.float32)))
        st.step.fill_(int(rng.integers(1, 1000)))
        stores[net] = st
    stores['classifier'].enable_ema()
    stores['classifier'].ema.mul_(0.5)
    cx = types.SimpleNamespace(stores=stores, rng=types.SimpleNamespace(state=torch.tensor([7, int(rng.integers(1, 99))], dtype=torch.int64)))
    return types.SimpleNamespace(cx=cx)


def test_save_restore_roundtrip_and_directory_protocol(tmp_path):
    from Training.Saver import Saver, ADAM_M, ADAM_V, EMA
    a = fake_session(1)
    saver = Saver(str(tmp_path))
    saver.set_save_path(comments='run A')
    run = os.path.basename(saver.save_dir)
    assert run.startswith('Run_') and len(run) == len('Run_2019-04-30_12_00_00')
    assert open(os.path.join(saver.save_dir, 'Comments.txt')).read() == 'run A'
    saver.save(a, 'model_0003.ckpt')
    a.cx.stores['classifier'].p.add_(1.0)
    path = saver.save(a, 'model_0012.ckpt')
    assert sorted(os.listdir(saver.save_dir)) == ['Comments.txt', 'model_0003.ckpt.npz', 'model_0012.ckpt.npz']
    z = np.load(path)
    for key in ('classifier/conv1_1/V', 'classifier/conv1_1/V' + ADAM_M, 'classifier/conv1_1/V' + ADAM_V, 'classifier/conv1_1/V' + EMA,
                'classifier/conv1_1/meanOnlyBatchNormalization/pop_mean', 'good_generator/gg_bn0/moving_mean', 'tg/adam_step/classifier',
                'tg/rng_state', 'tg/epoch'):
        assert key in z.files, key
    assert z['classifier/conv1_1/V'].shape == (3, 3, 3, 8) and int(z['tg/epoch']) == 12
    assert 'good_generator/gg_h0_lin/gg_h0_lin/kernel' + EMA not in z.files           # only classifier variables have shadows
    # restore into a differently initialised session: latest run, latest epoch
    b = fake_session(2)
    s2 = Saver(str(tmp_path))
    assert s2.restore(b, dir_names=None, epoch=None) == 12
    for net in a.cx.stores:
        sa, sb = a.cx.stores[net], b.cx.stores[net]
        assert torch.equal(sa.step, sb.step)
        for nm, _, trainable in sa.specs:                   # per variable (the 32-float alignment padding between them is not state)
            for which in (('value', 'm', 'v') + (('ema',) if sa.ema is not None else ())) if trainable else ('value',):
                np.testing.assert_array_equal(sa.get(nm, which), sb.get(nm, which), err_msg='%s %s' % (nm, which))
    assert torch.equal(a.cx.rng.state, b.cx.rng.state)
    # a named run and an explicit epoch (Train_goodGAN.py:142: saver.restore(sess, dir_names=config.RUN, epoch=config.RESTORE_EPOCH))
    c = fake_session(3)
    assert Saver(str(tmp_path)).restore(c, dir_names=run, epoch=3) == 3
    assert np.abs(c.cx.stores['classifier'].get('classifier/conv1_1/V') + 1.0 - a.cx.stores['classifier'].get('classifier/conv1_1/V')).max() < 1e-6   # epoch 3 was written before the +1
    os.makedirs(tmp_path / 'nothing_here_yet')
    with pytest.raises(ValueError, match='Cannot find ckpt file'):
        Saver(str(tmp_path / 'nothing_here_yet')).restore(c)
    # a checkpoint that lacks variables is an error, not a silent partial restore
    from Training.Saver import load_state_dict
    d = dict(np.load(path))
    del d['classifier/conv1_1/b' + ADAM_V]
    with pytest.raises(KeyError, match='lacks 1 variables'):
        load_state_dict(c.cx.stores, d)


def test_eastern_timestamp_offset():
    from datetime import datetime, timezone
    from Training.Saver import _eastern_now
    off = (_eastern_now().replace(tzinfo=None) - datetime.now(timezone.utc).replace(tzinfo=None)).total_seconds() / 3600
    assert round(off) in (-4, -5)


def _decode_event(payload):
    ev = {}
    for f, wt, v in O._fields(payload):
        if f == 1:
            ev['wall_time'] = struct.unpack('<d', v)[0]
        elif f == 2:
            ev['step'] = v
        elif f == 3:
            ev['file_version'] = v.decode()
        elif f == 5:
            vals = {}
            for f2, _, val in O._fields(v):
                tag = x = None
                for f3, wt3, vv in O._fields(val):
                    if f3 == 1:
                        tag = vv.decode()
                    elif f3 == 2:
                        x = struct.unpack('<f', vv)[0]
                vals[tag] = x
            ev['scalars'] = vals
    return ev


def test_summary_writes_tensorboard_event_files(tmp_path):
    from Training.Summary import Summary
    s = Summary(str(tmp_path), None, log_type='train', log_comments='hello')
    assert s.log_dir.startswith(os.path.join(str(tmp_path), 'train', 'Run_'))
    assert open(os.path.join(s.log_dir, 'Comments.txt')).read() == 'hello'
    tags = s.add_summary({'scalar': {'g_loss': None, 'd_loss': None}})
    assert tags == ['g_loss', 'd_loss']
    s.write(dict(g_loss=0.5, d_loss=1.25, ignored=9.0), 1)
    s.write(dict(g_loss=0.25, d_loss=1.0), 2)
    files = [f for f in os.listdir(s.log_dir) if f.startswith('events.out.tfevents.')]
    assert len(files) == 1
    events = [_decode_event(p) for p in O.read_tfrecord(os.path.join(s.log_dir, files[0]))]      # CRCs verified by the reader
    assert events[0]['file_version'] == 'brain.Event:2' and 'scalars' not in events[0]
    assert [e['step'] for e in events[1:]] == [1, 2]
    assert events[1]['scalars'] == {'g_loss': 0.5, 'd_loss': 1.25} and events[2]['scalars'] == {'g_loss': 0.25, 'd_loss': 1.0}
    assert open(os.path.join(s.log_dir, 'history.csv')).read().splitlines() == ['step,g_loss,d_loss', '1,0.5,1.25', '2,0.25,1']


def test_replica_resume_keeps_each_ranks_seed(tmp_path):
    """Data-parallel resume (round-1 advisor finding): rank 0 writes the checkpoint, its tg/rng_state holds rank 0's Philox seed.  A
    replica restoring it takes the step counter only — with rank 0's seed every replica would draw identical latents, dropout
    masks and noise.  A single-process resume still restores the full state (bit-identical continuation, tests/test_gpu_resume.py)."""
    from Training.Saver import Saver
    writer = fake_session(1)
    writer.cx.rng.state.copy_(torch.tensor([11, 42], dtype=torch.int64))            # rank 0: seed 11, 42 iterations done
    saver = Saver(str(tmp_path))
    saver.set_save_path(comments='dp')
    saver.save(writer, 'model_0001.ckpt')
    ranks = []
    for rank in range(2):
        s = fake_session(5 + rank)
        s.world, s.rank = 2, rank
        s.cx.rng.state.copy_(torch.tensor([11 + 7919 * rank, 0], dtype=torch.int64))   # Train.__init__: SEED + 7919 * rank
        assert Saver(str(tmp_path)).restore(s) == 1
        ranks.append(s)
    assert ranks[0].cx.rng.state.tolist() == [11, 42] and ranks[1].cx.rng.state.tolist() == [11 + 7919, 42]
    for net in writer.cx.stores:                                                    # weights and slots are the writer's on both ranks
        for r in ranks:
            for nm in writer.cx.stores[net].names(True):
                for which in ('value', 'm', 'v'):
                    np.testing.assert_array_equal(r.cx.stores[net].get(nm, which), writer.cx.stores[net].get(nm, which))
    solo = fake_session(9)
    solo.world = 1
    Saver(str(tmp_path)).restore(solo)
    assert solo.cx.rng.state.tolist() == [11, 42]
