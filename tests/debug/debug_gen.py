import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from oracle import step_cifar10 as S, nets_cifar10 as N, tf_ops as T
import gpu_common as G
from tg import ops
from Model.Good_GAN_cifar10 import _dense_view

P = S.init_params(0)
tr = G.fresh_trainer(G.make_config({}), P)
batch = S.synth_batch(100)
cx, m = tr.cx, tr.model
tr.feed(batch)
z, y = batch['z_g'], batch['y_g']
# oracle
zy = np.concatenate([z, y], 1)
h = T.relu(zy @ P['good_generator/gg_h0_lin/gg_h0_lin/kernel'] + P['good_generator/gg_h0_lin/gg_h0_lin/bias'])
r0 = h
b0, cache = T.batch_norm_train(h, P['good_generator/gg_bn0/gamma'], P['good_generator/gg_bn0/beta'], 1e-5)
b0_64, _ = T.batch_norm_train(h.astype(np.float64), P['good_generator/gg_bn0/gamma'].astype(np.float64), P['good_generator/gg_bn0/beta'].astype(np.float64), 1e-5)
x0 = T.conv_cond_concat(b0.reshape(-1, 4, 4, 512), y)
d0 = T.relu(T.conv2d_transpose(x0, P['good_generator/gg_dconv0/gg_dconv0/kernel']) + P['good_generator/gg_dconv0/gg_dconv0/bias'])
with cx.phase_scope('dbg', record=False):
    with cx.variable_scope('good_generator'):
        zyh = ops.cond_concat(tr.z_g_ph, tr.y_g_ph.t, 10)
        h0 = m._linear_fc(zyh, 8192, 'gg_h0_lin', activation=m._relu)
        print('lin+relu err', G.rel_err(h0.numpy(), r0), 'scale', np.abs(r0).max())
        hb = m._batch_norm_contrib(_dense_view(h0), 'gg_bn0', train=True)
        got = hb.numpy()
        print('bn0 err vs f32 oracle', G.rel_err(got, b0), 'vs f64 oracle', G.rel_err(got, b0_64), 'oracle32 vs 64', G.rel_err(b0, b0_64), 'scale', np.abs(b0).max())
        e = np.abs(got - b0_64); j = np.unravel_index(e.argmax(), e.shape)
        print('worst feature', j, 'var', cache[3][j[1]], 'mean', cache[2][j[1]], 'got', got[j], 'ref', b0_64[j])
        hr = ops.reshape(hb, 100, 4, 4, 512)
        hc = m._conv_cond_concat(hr, tr.y_g_ph)
        hd = m._deconv2d(hc, 256, name='gg_dconv0', activation=m._relu)
        print('dconv0 err', G.rel_err(hd.numpy(), d0), 'scale', np.abs(d0).max())
        b1, _ = T.batch_norm_train(d0, P['good_generator/gg_bn1/gamma'], P['good_generator/gg_bn1/beta'], 1e-5)
        hb1 = m._batch_norm_contrib(hd, 'gg_bn1', train=True)
        print('bn1 err', G.rel_err(hb1.numpy(), b1), 'scale', np.abs(b1).max())
        x1 = T.conv_cond_concat(b1, y)
        d1 = T.relu(T.conv2d_transpose(x1, P['good_generator/gg_dconv1/gg_dconv1/kernel']) + P['good_generator/gg_dconv1/gg_dconv1/bias'])
        hc1 = m._conv_cond_concat(hb1, tr.y_g_ph)
        hd1 = m._deconv2d(hc1, 128, name='gg_dconv1', activation=m._relu)
        print('dconv1 err', G.rel_err(hd1.numpy(), d1), 'scale', np.abs(d1).max())
        b2, _ = T.batch_norm_train(d1, P['good_generator/gg_bn2/gamma'], P['good_generator/gg_bn2/beta'], 1e-5)
        hb2 = m._batch_norm_contrib(hd1, 'gg_bn2', train=True)
        print('bn2 err', G.rel_err(hb2.numpy(), b2), 'scale', np.abs(b2).max())
        x2 = T.conv_cond_concat(b2, y)
        pre = T.conv2d_transpose(x2, P['good_generator/gg_dconv2/gg_dconv2/kernel']) + P['good_generator/gg_dconv2/gg_dconv2/bias']
        hc2 = m._conv_cond_concat(hb2, tr.y_g_ph)
        hd2 = m._deconv2d(hc2, 3, name='gg_dconv2', activation=None, narrow=True)
        print('dconv2 pre-tanh err', G.rel_err(hd2.numpy(), pre), 'scale', np.abs(pre).max())
        hd3 = m._deconv2d(hc2, 3, name='gg_dconv2', activation=m._tanh, narrow=True)
        print('dconv2 tanh err', G.rel_err(hd3.numpy(), np.tanh(pre)), 'abs', np.abs(hd3.numpy() - np.tanh(pre)).max())
        e = np.abs(hd3.numpy() - np.tanh(pre)); j = np.unravel_index(e.argmax(), e.shape)
        print('worst', j, 'pre', pre[j], 'hip pre', hd2.numpy()[j], 'tanh ref', np.tanh(pre[j]), 'hip', hd3.numpy()[j])
P64 = {k: v.astype(np.float64) for k, v in P.items()}
o64, _ = N.generator_fwd(P64, z.astype(np.float64), y.astype(np.float64))
o32, _ = N.generator_fwd(P, z, y)
with cx.phase_scope('dbg2', record=False):
    oh = m.good_generator(tr.z_g_ph, tr.y_g_ph).numpy()
print('G: hip vs f64', np.abs(oh - o64).max(), ' oracle32 vs f64', np.abs(o32 - o64).max(), ' hip vs oracle32', np.abs(oh - o32).max())
