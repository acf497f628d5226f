import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorflow-implementation-of-triple-gan_amd"))
from oracle import tf_ops as T
from tg import lib, geom
lib.load()
n, h, w, cin, cout = 2, 8, 8, 74, 3
rng = np.random.default_rng(1)
x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
wt = (rng.standard_normal((5, 5, cout, cin)) * 0.1).astype(np.float32)
bias = rng.standard_normal(cout).astype(np.float32)
ci_p, co_p = 96, 32
pre = T.conv2d_transpose(x, wt) + bias
w_pad = np.zeros((25, co_p, ci_p), np.float32); w_pad[:, :cout, :cin] = wt.reshape(25, cout, cin)
xp = np.zeros((n, h, w, ci_p), np.float32); xp[..., :cin] = x
bp = np.zeros(co_p, np.float32); bp[:cout] = bias
xd, wd, bd = torch.from_numpy(xp).cuda(), torch.from_numpy(w_pad).cuda(), torch.from_numpy(bp).cuda()
yd = torch.full((n, 2 * h, 2 * w, cout), 7.0, device='cuda')
for d in geom.deconv_fwd(n, h, w, ci_p, co_p, ld_out=cout, n_store=cout, act=None):
    lib.call("tg_igemm_f32", d, lib.ptr(xd), lib.ptr(wd), lib.ptr(bd), lib.ptr(yd), None, 0, lib.cur_stream())
y = yd.cpu().numpy()
e = np.abs(y - pre)
print('max err', e.max(), 'unwritten', (y == 7.0).sum())
idx = np.argwhere(e > 1e-3)
print(len(idx), idx[:10])
for i in idx[:5]:
    print(tuple(i), y[tuple(i)], pre[tuple(i)])
print('channel-wise max err', e.max(axis=(0,1,2)))
