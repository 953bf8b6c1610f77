import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from gennet_amd import ops
from oracle import keras_ref as K
rng = np.random.RandomState(0)
B, L, Cin, Cout, k, s, padding = 2, 64, 16, 128, 5, 1, 'same'
x = rng.randn(B, L, Cin).astype(np.float32); w = (rng.randn(k, Cin, Cout)/9).astype(np.float32)
Lout, pl = ops.conv_geometry(L, k, s, padding)
y = ops.conv1d_fwd(torch.tensor(x).cuda(), torch.tensor(w).cuda(), None, s, pl, Lout).cpu().numpy()
ref = K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), None, s, padding)
err = np.abs(y - ref)
print('max err', err.max(), 'scale', np.abs(ref).max())
bad = err > 1e-3
print('bad count', bad.sum(), 'of', bad.size)
print('bad per batch', bad.sum(axis=(1,2)))
print('bad rows (t) b0', np.where(bad[0].any(axis=1))[0][:40])
print('bad cols count b0', bad[0].any(axis=0).sum())
# test with x zero except one element to see mapping
for (tt, cc) in ((10, 0), (10, 5), (10, 9)):
    x2 = np.zeros_like(x); x2[0, tt, cc] = 1.0
    y2 = ops.conv1d_fwd(torch.tensor(x2).cuda(), torch.tensor(w).cuda(), None, s, pl, Lout).cpu().numpy()
    r2 = K.conv1d_fwd(x2.astype(np.float64), w.astype(np.float64), None, s, padding)
    nz = np.where(np.abs(y2[0]).sum(axis=1) > 0)[0]; nzr = np.where(np.abs(r2[0]).sum(axis=1) > 0)[0]
    print('impulse', tt, cc, 'rows got', nz, 'expected', nzr, 'maxerr', np.abs(y2-r2).max())
