"""Development check + A/B timing of the stride-2 transform-domain kernels (csrc/conv_wino_s2.hip) against the fp64 oracle and the direct kernels.
python tests/tools/wino_s2_check.py [--bench] [--wgrad]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from gennet_amd import ops
from oracle import keras_ref as K

dev = torch.device('cuda:0')


def check(B, L, Cin, Cout, padding, seed=0):
    rng = np.random.RandomState(seed + L)
    x = rng.randn(B, L, Cin).astype(np.float32)
    lim = np.sqrt(6.0 / (5 * (Cin + Cout)))
    w = rng.uniform(-lim, lim, (5, Cin, Cout)).astype(np.float32)
    b = (rng.randn(Cout) * 0.1).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    ref = np.maximum(K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 2, padding), 0)
    dy = rng.randn(B, Lout, Cout).astype(np.float32)
    dx_ref, _, _ = K.conv1d_bwd(x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64), 2, padding)
    xd, wd, bd, dyd = (torch.from_numpy(v).to(dev) for v in (x, w, b, dy))
    wt = ops.conv1d_transpose_w(wd)
    res = {}
    for math in ('wino', 'fp32'):
        with ops.conv_math(math):
            ops.prof_enable(True); ops.prof_reset()
            y = ops.conv1d_fwd(xd, wd, bd, 2, pl, Lout, 'relu').cpu().numpy()
            dx = ops.conv1d_dgrad(dyd, wt, L, 2, pl).cpu().numpy()
            n7 = ops.prof_collect(7)['launches']
            ops.prof_enable(False)
        res[math] = (y, dx, n7)
    s, sd = np.sqrt(np.mean(ref ** 2)), np.sqrt(np.mean(dx_ref ** 2))
    e = {m: (np.abs(res[m][0] - ref).max() / s, np.sqrt(np.mean((res[m][0] - ref) ** 2)) / s, np.abs(res[m][1] - dx_ref).max() / sd,
             np.sqrt(np.mean((res[m][1] - dx_ref) ** 2)) / sd) for m in res}
    print('B %d L %d %d->%d %s pl %d: wino launches %d | fwd max %.2e rms %.2e (direct %.2e %.2e) | dgrad max %.2e rms %.2e (direct %.2e %.2e)'
          % (B, L, Cin, Cout, padding, pl, res['wino'][2], e['wino'][0], e['wino'][1], e['fp32'][0], e['fp32'][1], e['wino'][2], e['wino'][3], e['fp32'][2], e['fp32'][3]), flush=True)
    assert res['wino'][2] >= 1 and e['wino'][0] < 2e-5 and e['wino'][2] < 2e-5, 'stride-2 transform-domain result wrong'


def bench(B, L, Cin, Cout, padding, reps=5):
    x = torch.randn(B, L, Cin, device=dev); w = torch.randn(5, Cin, Cout, device=dev) * 0.02; b = torch.zeros(Cout, device=dev)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    dy = torch.randn(B, Lout, Cout, device=dev); wt = ops.conv1d_transpose_w(w)
    for what, fn in (('fwd', lambda: ops.conv1d_fwd(x, w, b, 2, pl, Lout, 'relu')), ('dgrad', lambda: ops.conv1d_dgrad(dy, wt, L, 2, pl))):
        out = {}
        for math in ('fp32', 'wino'):
            with ops.conv_math(math):
                fn(); fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                out[math] = e0.elapsed_time(e1) / reps
        fl = 2.0 * B * Lout * 5 * Cin * Cout
        print('%-5s B %d L %d %d->%d: direct %.3f ms (%.1f) | transform domain %.3f ms (%.1f algorithmic TFLOP/s)  x%.3f'
              % (what, B, L, Cin, Cout, out['fp32'], fl / out['fp32'] / 1e9, out['wino'], fl / out['wino'] / 1e9, out['fp32'] / out['wino']), flush=True)


def check_wgrad(B, L, Cin, Cout, padding, seed=0):
    rng = np.random.RandomState(seed + L)
    x = rng.randn(B, L, Cin).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    dy = rng.randn(B, Lout, Cout).astype(np.float32)
    w0 = np.zeros((5, Cin, Cout))
    _, dw_ref, db_ref = K.conv1d_bwd(x.astype(np.float64), w0, dy.astype(np.float64), 2, padding)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    res = {}
    for math in ('wino', 'fp32'):
        with ops.conv_math(math):
            ops.prof_enable(True); ops.prof_reset()
            dw, db = ops.conv1d_wgrad(xd, dyd, 5, 2, pl)
            n8 = ops.prof_collect(8)['launches']
            ops.prof_enable(False)
        res[math] = (dw.cpu().numpy(), db.cpu().numpy(), n8)
    s = np.sqrt(np.mean(dw_ref ** 2))
    e = {m: (np.abs(res[m][0] - dw_ref).max() / s, np.sqrt(np.mean((res[m][0] - dw_ref) ** 2)) / s) for m in res}
    eb = np.abs(res['wino'][1] - db_ref).max() / np.abs(db_ref).max()
    print('wgrad B %d L %d %d->%d %s pl %d: launches %d | max %.2e rms %.2e (direct %.2e %.2e) | db %.1e'
          % (B, L, Cin, Cout, padding, pl, res['wino'][2], e['wino'][0], e['wino'][1], e['fp32'][0], e['fp32'][1], eb), flush=True)
    assert res['wino'][2] == 1 and e['wino'][0] < 5e-5 and eb < 1e-5, 'stride-2 transform-domain weight gradient wrong'


def bench_wgrad(B, L, Cin, Cout, padding, reps=5):
    x = torch.randn(B, L, Cin, device=dev)
    Lout, pl = ops.conv_geometry(L, 5, 2, padding)
    dy = torch.randn(B, Lout, Cout, device=dev)
    out = {}
    for math in ('fp32', 'wino'):
        with ops.conv_math(math):
            fn = lambda: ops.conv1d_wgrad(x, dy, 5, 2, pl)
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            out[math] = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * Lout * 5 * Cin * Cout
    print('wgrad B %d L %d %d->%d: direct %.3f ms (%.1f) | transform domain %.3f ms (%.1f algorithmic TFLOP/s)  x%.3f'
          % (B, L, Cin, Cout, out['fp32'], fl / out['fp32'] / 1e9, out['wino'], fl / out['wino'] / 1e9, out['fp32'] / out['wino']), flush=True)


if __name__ == '__main__' and '--wgrad' in sys.argv:
    for args in [(2, 64, 64, 64, 'same'), (3, 133, 64, 128, 'valid'), (2, 150, 128, 256, 'same'), (2, 151, 64, 64, 'same'), (1, 300, 256, 128, 'valid'), (7, 1024, 128, 64, 'same'),
                 (5, 37, 64, 64, 'same'), (2, 6, 64, 64, 'valid'), (40, 1018, 64, 128, 'valid')]:
        check_wgrad(*args)
    if '--bench' in sys.argv:
        bench_wgrad(1024, 1024, 512, 1024, 'same'); bench_wgrad(512, 1024, 512, 1024, 'same'); bench_wgrad(256, 1018, 512, 1024, 'valid'); bench_wgrad(256, 2040, 256, 512, 'valid')
        bench_wgrad(256, 4084, 128, 256, 'valid'); bench_wgrad(256, 8192, 64, 128, 'valid')
    sys.exit(0)

if __name__ == '__main__':
    for args in [(2, 64, 64, 64, 'same'), (3, 133, 64, 128, 'valid'), (2, 150, 128, 256, 'same'), (2, 151, 64, 64, 'same'), (1, 300, 256, 128, 'valid'), (2, 1024, 512, 1024, 'same'),
                 (5, 37, 32, 64, 'same'), (2, 6, 64, 64, 'valid')]:
        check(*args)
    if '--bench' in sys.argv:
        bench(1024, 1024, 512, 1024, 'same'); bench(512, 1024, 512, 1024, 'same'); bench(256, 1018, 512, 1024, 'valid'); bench(256, 2040, 256, 512, 'valid'); bench(256, 253, 256, 512, 'valid')
