"""Diagnostic (GPU): error of the fp32 filter gradient against a float64 einsum, csrc/wgrad3x3.hip (default routing) vs the generic per-tap
kernel (tg_conv3x3_policy 2), at the long-horizon run's launch sizes (40 images) and the bench line's (250) — evidence for the settling
window of tests/test_gpu_long_horizon.py.  Round 2, one MI355X: 3.7e-4 vs 4.6e-4 (32x32, 128 -> 128, 40 images, scale 786), 2.3e-3 vs
2.8e-3 (250 images, scale 1 924)."""
import sys, ctypes as C
sys.path.insert(0, '/root/repo/tensorflow-implementation-of-triple-gan_amd'); sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tg import geom, lib
lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(13)
for (hw, cin, cout, n) in ((32, 128, 128, 40), (16, 128, 256, 40), (16, 256, 256, 40), (32,128,128,250)):
    x = torch.from_numpy(rng.standard_normal((n, hw, hw, cin)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((n, hw, hw, cout)).astype(np.float32)).cuda()
    d = geom.conv_wgrad(n, hw, hw, cin, cout, 3, 1, 'SAME')
    ref = None
    # float64 reference of tap (0,0) and (-1,-1) via torch
    X = x.double(); DY = dy.double()
    ref_c = torch.einsum('nhwc,nhwk->ck', X, DY).cpu().numpy()                       # centre tap
    ref_tl = torch.einsum('nhwc,nhwk->ck', X[:, :-1, :-1], DY[:, 1:, 1:]).cpu().numpy()  # tap (dy=-1, dx=-1)
    for policy in (0, 2):
        was = lib.call('tg_conv3x3_policy', policy)
        ns = geom.wgrad_splits(d, False)
        slab = torch.zeros((ns, 9, cin, cout), device='cuda')
        lib.call('tg_wgrad_f32', d, lib.ptr(x), lib.ptr(dy), lib.ptr(slab), ns, st)
        lib.call('tg_conv3x3_policy', was)
        dw = slab.cpu().numpy().astype(np.float64).sum(0)
        print(hw, cin, cout, n, 'policy', policy, 'ns', ns, 'centre err %.3e' % np.abs(dw[4] - ref_c).max(), 'corner err %.3e' % np.abs(dw[0] - ref_tl).max(), 'scale %.1f' % np.abs(ref_c).max())
