import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import nets_goodgan as N, step_goodgan as S, tf_ops as T
import gpu_common as G
from test_oracle_goodgan import scrambled
from test_gpu_goodgan import trainer, f64
from tg.runtime import InjectedRNG
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
T.MFMA_BF16 = prec == 'bf16'
data = 'svhn'
P = {k: v.astype(np.float32).astype(np.float64) for k, v in scrambled(data, 3).items()}
tr = trainer(data, P, prec=prec)
cx, m = tr.cx, tr.model
n = 5
sizes = dict(B_G=n, L_C=n, U_C=n, L_D=1, U_D=n - 1)
b = f64(S.synth_batch(data, 5, sizes))
rnd = S.synth_rnd(data, 6, sizes)
rng = np.random.default_rng(7)
r = rnd['G']['D_fake']
img = b['x_l_c']
logits, dc, _ = N.seq_fwd(P, N.discriminator_layers(data), img, b['y_l_c'], f64(r), True)
dl = rng.standard_normal(logits.shape).astype(np.float32)
gref, dimg = N.seq_bwd(P, N.discriminator_layers(data), dc, dl.astype(np.float64), b['y_l_c'], f64(r))
cx.rng = InjectedRNG({'Td/D/' + k: v for k, v in r.items()}, cx.device)
with cx.phase_scope('Td', train_nets=('discriminator',)):
    ia = cx.from_numpy(img)
    ia.requires_grad = True
    with cx.rng_scoped('Td/D'):
        _, lg = m.discriminator(ia, cx.from_numpy(b['y_l_c']))
    lg.grad = cx.from_numpy(dl, ld=32)
    cx.backward()
print('logits rel', G.rel_err(lg.numpy(), logits))
st = cx.stores['discriminator']
for k, ref in gref.items():
    d = st.get(k, 'grad') - ref
    print('%-40s L2 %.2e max %.2e' % (k, np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30), np.abs(d).max() / max(np.abs(ref).max(), 1e-30)))
print('dimg', G.rel_err(ia.grad.numpy().reshape(dimg.shape), dimg))
