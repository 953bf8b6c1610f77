"""Soak version of tests/test_wino_gpu.py::test_repeated_launches_are_bit_identical: many repeats of forward, data gradient and weight gradient of the
transform-domain kernels, bit for bit.  python tests/tools/determinism_soak.py [repeats]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gennet_amd import ops
from gennet_amd.engine import device

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = device()
total = 0
for (Cin, Cout, L, stride, B) in [(64, 128, 2048, 1, 256), (128, 256, 2044, 1, 256), (256, 512, 2048, 1, 128), (512, 1024, 2048, 1, 64), (64, 128, 1024, 2, 256), (256, 512, 2040, 2, 128),
                                 (512, 1024, 1018, 2, 128)]:
    x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 3, 0, dev); w = ops.fill_normal((5, Cin, Cout), 0.0, 0.05, 4, 0, dev)
    Lout, pl = ops.conv_geometry(L, 5, stride, 'valid')
    dy = ops.fill_normal((B, Lout, Cout), 0.0, 1.0, 5, 0, dev)
    wt = ops.conv1d_transpose_w(w)
    first, bad = None, [0, 0, 0]
    for rep in range(reps):
        got = (ops.conv1d_fwd(x, w, None, stride, pl, Lout, 'relu'), ops.conv1d_dgrad(dy, wt, L, stride, pl), ops.conv1d_wgrad(x, dy, 5, stride, pl)[0])
        if first is None:
            first = got
        else:
            for k, (a, b) in enumerate(zip(got, first)):
                bad[k] += int((a != b).sum().item())
    total += sum(bad)
    print('%4d -> %4d stride %d, B %3d, %d repeats: mismatching elements forward %d, data gradient %d, weight gradient %d' % (Cin, Cout, stride, B, reps, bad[0], bad[1], bad[2]), flush=True)
print('TOTAL', total)
sys.exit(1 if total else 0)
