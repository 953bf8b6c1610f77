#!/usr/bin/env python
"""Template-synthesis throughput (SURVEY 8d metric iii): templates/s of the batched HIP synthesiser (FD chirp + whitening ->
irFFT x2 -> argmax-align -> crop) and of coloured-noise generation, next to the numpy oracle chain timed on one host core
(the reference's loop is serial, gw_template_maker.py:676).

  python tests/tools/synth_bench.py [--fs 2048] [--nb 8192] [--reps 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fs', type=int, default=2048)
    ap.add_argument('--nb', type=int, default=8192)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--cpu-templates', type=int, default=200)
    a = ap.parse_args()
    from gennet_amd import templates as T
    from oracle import synth_ref as S
    fs, Tobs = a.fs, 4
    N = fs * Tobs; Nf = N // 2 + 1
    psd = S.analytic_psd(Nf, 1.0 / Tobs)
    np.random.seed(1)
    pars = [T.gen_par(fs, Tobs, mdist='hunt_constrain', beta=[0.45, 0.55]) for _ in range(a.nb)]
    m1 = np.array([p.m1 for p in pars]); m2 = np.array([p.m2 for p in pars]); idx = np.array([p.idx for p in pars])
    syn = T.Synth(fs, Tobs, psd)
    syn.templates(m1[:64], m2[:64], idx[:64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ts, _ = syn.templates(m1, m2, idx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    alg_bytes = a.nb * (2 * Nf * 16 + Nf * 8 + fs * 8)
    print('templates (fused kernel)   : fs=%d nb=%d  %.3f ms  %.0f templates/s  (%.1f GB/s algorithmic)' % (fs, a.nb, dt * 1e3, a.nb / dt, alg_bytes / dt / 1e9))
    syn.templates(m1[:64], m2[:64], idx[:64], fused=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ts_u, _ = syn.templates(m1, m2, idx, fused=False)
    torch.cuda.synchronize()
    du = (time.perf_counter() - t0) / a.reps
    print('templates (separate kernels): fs=%d nb=%d  %.3f ms  %.0f templates/s  -> fused / separate = %.2fx; max |diff| / max = %.2e'
          % (fs, a.nb, du * 1e3, a.nb / du, du / dt, float((ts - ts_u).abs().max() / ts_u.abs().max())))
    T.gen_noise_device(fs, Tobs, psd, 64, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(a.reps):
        x = T.gen_noise_device(fs, Tobs, psd, a.nb, seed=3, offset=r * a.nb * Nf)
    torch.cuda.synchronize()
    dn = (time.perf_counter() - t0) / a.reps
    print('noise     : fs=%d nb=%d  %.3f ms  %.0f series/s' % (fs, a.nb, dn * 1e3, a.nb / dn))
    n = a.cpu_templates
    t0 = time.perf_counter()
    for p in pars[:n]:
        S.gen_bbh(fs, Tobs, psd, p, Fp=syn.Fp, Fc=syn.Fc)
    dc = (time.perf_counter() - t0) / n
    print('cpu oracle: %d templates, 1 core: %.1f templates/s  -> GPU/CPU = %.0fx' % (n, 1.0 / dc, (a.nb / dt) * dc))


if __name__ == '__main__':
    main()
