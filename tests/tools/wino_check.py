"""Development check + A/B timing of the transform-domain conv kernel (csrc/conv_wino.hip) against the fp64 oracle and the direct kernel.
python tests/tools/wino_check.py [--bench]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from gennet_amd import ops
from oracle import keras_ref as K

dev = torch.device('cuda:0')


def check(B, L, Cin, Cout, padding, act='linear', seed=0):
    rng = np.random.RandomState(seed)
    x = rng.randn(B, L, Cin).astype(np.float32)
    lim = np.sqrt(6.0 / (5 * (Cin + Cout)))
    w = rng.uniform(-lim, lim, (5, Cin, Cout)).astype(np.float32)
    b = rng.randn(Cout).astype(np.float32) * 0.1
    Lout, pl = ops.conv_geometry(L, 5, 1, padding)
    ref = K.conv1d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 1, padding)
    if act == 'tanh':
        ref = np.tanh(ref)
    xd, wd, bd = (torch.from_numpy(v).to(dev) for v in (x, w, b))
    yw = ops.conv1d_fwd_wino(xd, wd, bd, pl, Lout, act).cpu().numpy()
    yd = ops.conv1d_fwd(xd, wd, bd, 1, pl, Lout, act).cpu().numpy()
    s = np.sqrt(np.mean(ref ** 2))
    ew, ed = np.abs(yw - ref).max() / s, np.abs(yd - ref).max() / s
    rw, rd = np.sqrt(np.mean((yw - ref) ** 2)) / s, np.sqrt(np.mean((yd - ref) ** 2)) / s
    print('B %d L %d %d->%d %s %s: wino max %.2e rms %.2e | direct max %.2e rms %.2e' % (B, L, Cin, Cout, padding, act, ew, rw, ed, rd), flush=True)
    assert ew < 2e-5, 'wino result wrong'


def bench(B, L, Cin, Cout, reps=5):
    x = torch.randn(B, L, Cin, device=dev)
    lim = np.sqrt(6.0 / (5 * (Cin + Cout)))
    w = (torch.rand(5, Cin, Cout, device=dev) * 2 - 1) * lim
    b = torch.zeros(Cout, device=dev)
    out = {}
    for name, fn in (('direct', lambda: ops.conv1d_fwd(x, w, b, 1, 2, L)), ('wino', lambda: ops.conv1d_fwd_wino(x, w, b, 2, L))):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = ms
        print('%-7s B %d L %d %d->%d: %.3f ms  %.1f TFLOP/s algorithmic' % (name, B, L, Cin, Cout, ms, 2.0 * B * L * 5 * Cin * Cout / ms / 1e9), flush=True)
    print('   speed-up %.3f' % (out['direct'] / out['wino']), flush=True)


if __name__ == '__main__' and '--wgrad' not in sys.argv:
    if '--abl' in sys.argv:        # timing only (ablation build: results are wrong)
        bench(512, 2048, 512, 1024)
        sys.exit(0)
    for args in [(2, 64, 64, 64, 'same'), (3, 130, 128, 256, 'same'), (2, 257, 64, 128, 'valid'), (1, 2048, 512, 1024, 'same'), (2, 2044, 128, 256, 'valid', 'tanh'),
                 (5, 37, 8, 64, 'same'), (1, 1, 16, 64, 'same')]:
        check(*args)
    if '--bench' in sys.argv:
        bench(512, 2048, 512, 1024)
        bench(512, 2048, 256, 512)
        bench(512, 2048, 128, 256)
        bench(256, 2048, 64, 128)


def check_wgrad(B, L, Cin, Cout, padding, seed=0):
    rng = np.random.RandomState(seed)
    x = (np.tanh(rng.randn(B, L, Cin)) * (rng.rand(B, L, Cin) > 0.2) / 0.8).astype(np.float32)
    Lout, pl = ops.conv_geometry(L, 5, 1, padding)
    dy = rng.randn(B, Lout, Cout).astype(np.float32)
    _, dw_ref, db_ref = K.conv1d_bwd(x.astype(np.float64), np.zeros((5, Cin, Cout)), dy.astype(np.float64), 1, padding)
    xd, dyd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev)
    with ops.conv_math('wino'):
        dww, dbw = ops.conv1d_wgrad(xd, dyd, 5, 1, pl)
    with ops.conv_math('fp32'):
        dwd, dbd = ops.conv1d_wgrad(xd, dyd, 5, 1, pl)
    s = np.sqrt(np.mean(dw_ref ** 2))
    ew, ed = np.abs(dww.cpu().numpy() - dw_ref).max() / s, np.abs(dwd.cpu().numpy() - dw_ref).max() / s
    rw, rd = np.sqrt(np.mean((dww.cpu().numpy() - dw_ref) ** 2)) / s, np.sqrt(np.mean((dwd.cpu().numpy() - dw_ref) ** 2)) / s
    eb = np.abs(dbw.cpu().numpy() - db_ref).max() / np.abs(db_ref).max()
    print('wgrad B %d L %d %d->%d %s: wino max %.2e rms %.2e | direct max %.2e rms %.2e | db %.1e' % (B, L, Cin, Cout, padding, ew, rw, ed, rd, eb), flush=True)
    assert ew < 5e-5 and eb < 1e-5, 'wino wgrad wrong'


def bench_wgrad(B, L, Cin, Cout, reps=5):
    x = torch.randn(B, L, Cin, device=dev); dy = torch.randn(B, L, Cout, device=dev)
    out = {}
    for name in ('fp32', 'wino'):
        with ops.conv_math(name):
            fn = lambda: ops.conv1d_wgrad(x, dy, 5, 1, 2)
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / reps
        print('wgrad %-5s B %d L %d %d->%d: %.3f ms  %.1f TFLOP/s algorithmic' % (name, B, L, Cin, Cout, out[name], 2.0 * B * L * 5 * Cin * Cout / out[name] / 1e9), flush=True)
    print('   speed-up %.3f' % (out['fp32'] / out['wino']), flush=True)


if __name__ == '__main__' and '--wgrad' in sys.argv:
    if '--abl' in sys.argv:        # timing only (ablation build: results are wrong)
        bench_wgrad(512, 2048, 512, 1024)
        sys.exit(0)
    for args in [(2, 64, 64, 64, 'same'), (3, 130, 128, 256, 'same'), (2, 257, 64, 128, 'valid'), (5, 37, 64, 64, 'same'), (1, 1, 64, 64, 'same'), (4, 2048, 128, 256, 'same'),
                 (2, 2044, 128, 256, 'valid'), (16, 600, 256, 512, 'same')]:
        check_wgrad(*args)
    if '--bench' in sys.argv:
        bench_wgrad(512, 2048, 512, 1024); bench_wgrad(512, 2048, 256, 512); bench_wgrad(512, 2048, 128, 256); bench_wgrad(256, 2048, 64, 128)
