import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from oracle import step_cifar10 as S, nets_cifar10 as N, tf_ops as T
import gpu_common as G
from tg.runtime import InjectedRNG, Act
from tg import ops
from tg.batching import concat_acts

P = S.init_params(0)
zca = G.zca()
tr = G.fresh_trainer(G.make_config({}), P)
batch = S.synth_batch(100); rnd = S.synth_rnd(200)
cx, m, c = tr.cx, tr.model, tr.config
cx.rng = InjectedRNG(G.injected_arrays(rnd), cx.device)
tr.feed(batch)
# oracle pieces
Gimg, _ = N.generator_fwd(P, batch['z_g'], batch['y_g'])
pops = {}
c_unl, _, _ = N.classifier_fwd(P, N.zca_apply(batch['x_u_c'], *zca), True, rnd['D']['C_unl'], pops)
c_unl_d, _, _ = N.classifier_fwd(P, N.zca_apply(batch['x_u_d'], *zca), True, rnd['D']['C_unl_d'], pops)
with cx.phase_scope('D', train_nets=('discriminator',)):
    Gh = m.good_generator(tr.z_g_ph, tr.y_g_ph)
    xz = m.zca().apply(concat_acts([tr.x_u_c_ph, tr.x_u_d_ph]))
    with cx.rng_scoped('D/C'):
        cl, _ = m.classifier(xz, True, segments=[50, 80])
    print('G err', G.rel_err(Gh.numpy(), Gimg))
    ref = np.concatenate([c_unl, c_unl_d]); got = cl.numpy()
    print('C logits err', G.rel_err(got, ref), 'abs max', np.abs(got-ref).max(), 'logit scale', np.abs(ref).max())
    srt = np.sort(ref, axis=1); gap = srt[:, -1] - srt[:, -2]
    print('min top2 gap', gap.min(), 'argmax agree', (got.argmax(1) == ref.argmax(1)).mean(), 'n disagree', (got.argmax(1) != ref.argmax(1)).sum())
