"""Checkpoint / resume — counterpart of the reference's Training/Saver.py (same class, method names and directory
layout: <save_dir>/Run_<YYYY-mm-dd_HH_MM_SS US/Eastern>/model_<epoch:04d>.ckpt.*, Comments.txt; :14-70).

tf.train.Saver() stores every global variable of the graph; the same set is stored here, one array per variable under its
TensorFlow name, in ONE file `model_<epoch>.ckpt.npz` (so `_findfilename`'s name.suffix.ext split (:54) keeps working):

    <variable>                               trainable values, pop_mean / moving statistics     (tf.GraphKeys.GLOBAL_VARIABLES)
    <variable>/Adam_optimizer                Adam first-moment slot  m                          (train_base.py:91-97) [UNVERIFIED-TF slot naming]
    <variable>/Adam_optimizer_1              Adam second-moment slot v
    <variable>/ExponentialMovingAverage      EMA shadow of the classifier variables             (Train_goodGAN.py:101-103)
    tg/adam_step/<network>                   step count t of that network's optimiser (TF keeps beta1_power = beta1^t, beta2_power)
    tg/rng_state                             Philox (seed, step) of the writing rank — TF's graph-level seeds have no equivalent.  On
                                             restore every rank takes the STEP and keeps its own seed (replicas draw different
                                             z / y / masks / noise: config.SEED + 7919 * rank, Train.__init__)
    tg/epoch                                 epoch the file was written after

Save gathers from the flat device buffers; restore scatters back — the MFMA-side filter layouts are rebuilt from the values every
step, so nothing else needs to be kept.  With several replicas rank 0 writes (weights are identical; rank-local running
statistics are averaged first: Train.sync_running_state)."""
import os
from datetime import datetime, timedelta, timezone as _tz

import numpy as np

ADAM_M, ADAM_V, EMA = '/Adam_optimizer', '/Adam_optimizer_1', '/ExponentialMovingAverage'


def _eastern_now():
    """datetime.now(timezone('US/Eastern')) of the reference (:21) without pytz: EST/EDT by the US rule (second Sunday of March
    to first Sunday of November)."""
    utc = datetime.now(_tz.utc)
    y = utc.year
    march = datetime(y, 3, 8, 7, tzinfo=_tz.utc)           # 02:00 EST = 07:00 UTC, on the second Sunday
    march += timedelta(days=(6 - march.weekday()) % 7)
    nov = datetime(y, 11, 1, 6, tzinfo=_tz.utc)            # 02:00 EDT = 06:00 UTC, on the first Sunday
    nov += timedelta(days=(6 - nov.weekday()) % 7)
    return utc + timedelta(hours=-4 if march <= utc < nov else -5)


def state_dict(stores, rng=None, epoch=0):
    """every variable of the three networks with its optimiser slots, as host arrays keyed by TF variable names."""
    out = {}
    for net, st in stores.items():
        for nm, _shape, trainable in st.specs:
            out[nm] = st.get(nm)
            if trainable:
                out[nm + ADAM_M] = st.get(nm, 'm')
                out[nm + ADAM_V] = st.get(nm, 'v')
                if st.ema is not None:
                    out[nm + EMA] = st.get(nm, 'ema')
        out['tg/adam_step/' + net] = st.step.detach().cpu().numpy().astype(np.int64)
    if rng is not None and hasattr(rng, 'state'):
        out['tg/rng_state'] = rng.state.detach().cpu().numpy()
    out['tg/epoch'] = np.asarray(epoch, np.int64)
    return out


def load_state_dict(stores, d, rng=None, strict=True, keep_seed=False):
    """keep_seed: restore only the RNG's step counter (data-parallel resume: the file holds rank 0's seed)."""
    import torch
    missing = []
    for net, st in stores.items():
        for nm, _shape, trainable in st.specs:
            keys = [(nm, None)] + ([(nm + ADAM_M, 'm'), (nm + ADAM_V, 'v')] if trainable else [])
            if trainable and st.ema is not None:
                keys.append((nm + EMA, 'ema'))
            for key, which in keys:
                if key not in d:
                    missing.append(key)
                    continue
                a = np.ascontiguousarray(d[key], np.float32).reshape(-1)
                dst = st.value(nm) if which is None else st._slice(getattr(st, which), nm)
                assert a.size == dst.numel(), (key, a.size, dst.numel())
                dst.copy_(torch.from_numpy(a))
        k = 'tg/adam_step/' + net
        if k in d:
            st.step.copy_(torch.from_numpy(np.asarray(d[k]).astype(np.int32).reshape(1)))
        else:
            missing.append(k)
    if rng is not None and hasattr(rng, 'state') and 'tg/rng_state' in d:
        saved = torch.from_numpy(np.asarray(d['tg/rng_state']).astype(np.int64))
        if keep_seed:
            rng.state[1:2].copy_(saved[1:2])
        else:
            rng.state.copy_(saved)
    if strict and missing:
        raise KeyError("checkpoint lacks %d variables, e.g. %s" % (len(missing), missing[:3]))
    return missing


class Saver(object):
    def __init__(self, save_dir, **kwargs):
        self.save_dir = save_dir

    def set_save_path(self, **kwargs):                                              # :19-28
        self.save_dir = os.path.join(self.save_dir, 'Run_' + _eastern_now().strftime("%Y-%m-%d_%H_%M_%S"))
        os.makedirs(self.save_dir, exist_ok=True)
        if 'comments' in kwargs:
            self.comments = kwargs.get('comments')
            self._write_comments()

    def save(self, sess, save_name):
        """sess: the Train object (its Context holds what a tf.Session holds).  :30-32"""
        path = os.path.join(self.save_dir, save_name)
        epoch = int(os.path.basename(save_name).split('.')[0].split('_')[-1]) if '_' in save_name else 0
        d = state_dict(sess.cx.stores, sess.cx.rng, epoch)
        tmp = path + '.tmp.npz'
        np.savez(tmp, **d)
        os.replace(tmp, path + '.npz')                                              # readers never see a partial file
        return path + '.npz'

    def restore(self, sess, dir_names=None, epoch=None):                            # :34-37
        self.save_dir, filename, start_epoch = self._findfilename(dir_names, epoch)
        with np.load(filename + '.npz') as z:
            # a replica keeps its own seed: with rank 0's every replica would draw the same latents, masks and noise
            load_state_dict(sess.cx.stores, z, sess.cx.rng, keep_seed=getattr(sess, 'world', 1) > 1)
        return start_epoch

    def _findfilename(self, dir_names=None, epoch=None):                            # :39-66
        if dir_names is None:
            dir_names = sorted(f for f in next(os.walk(self.save_dir))[1] if f.startswith('Run'))
            if not dir_names:
                raise ValueError('Cannot find ckpt file!')
            save_dir = os.path.join(self.save_dir, dir_names[-1])
        else:
            save_dir = os.path.join(self.save_dir, dir_names)
        checkpoints = sorted(f for f in next(os.walk(save_dir))[2] if f.startswith("model") and '.tmp.' not in f)
        if not checkpoints:
            raise ValueError('Cannot find ckpt file!')
        name, suffix, _ = checkpoints[-1].split('.')
        if epoch is None:
            start_epoch = name.split('_')[1]
            checkpoints = os.path.join(save_dir, name + '.' + suffix)
        else:
            start_epoch = epoch
            checkpoints = os.path.join(save_dir, name.split('_')[0] + '_' + str(epoch).zfill(4) + '.' + suffix)
        return save_dir, checkpoints, int(start_epoch)

    def _write_comments(self):                                                      # :68-70
        with open(os.path.join(self.save_dir, 'Comments.txt'), 'w') as txt_file:
            txt_file.write(self.comments)
