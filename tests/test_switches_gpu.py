"""Every environment switch the shipped library reads, each exercised once: the switches are read once per process, so the
switched leg runs in a child process and its results come back through a file.

Library (csrc/): GN_CONV_NOPIPE, GN_CONV_NODMA (fallback kernel families of the direct convolution), GN_CONV_NOMERGE, GN_CONV_NONARROW,
GN_CONV_NOPATCH (members / block order of the pipelined family: same arithmetic, bit-identical results), GN_WGRAD_NOPIPE (register-staged weight
gradient), GN_BF16X3_MIN_GFLOP (size gate of the opt-in split; the suite runs it at 0, here at its default), GN_BF16X3_NO_MERGE (tests/test_bf16x3_gpu.py).
Planner (layers.py): GN_NO_UPFOLD, GN_NO_LAZYGRAD (tests/test_nets_gpu.py), GN_NO_CONVSTATS, GN_NO_DROPGEN (here).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# B, L, Cin, Cout, stride, padding: a 512-block patch-ordered launch, a stride-2 forward, small launches that take the narrow-wave tiles and the
# merged stride-2 data gradient, ragged rows
_SHAPES = [(16, 1024, 64, 512, 1, 'same'), (8, 600, 128, 256, 2, 'valid'), (2, 300, 256, 128, 2, 'same'), (3, 133, 64, 128, 1, 'valid'), (8, 1024, 512, 1024, 2, 'valid')]


def _direct_cases():
    """Forward, data gradient and weight gradient of every shape on the DIRECT kernels (conv math 'fp32'), inputs from the device generator; the last
    shape once more on small integers (an exact fmaf chain whatever its order)."""
    from gennet_amd import ops
    dev = torch.device('cuda')
    out = []
    with ops.conv_math('fp32'):
        for k, (B, L, Cin, Cout, s, padding) in enumerate(_SHAPES):
            Lout, pl = ops.conv_geometry(L, 5, s, padding)
            x = ops.fill_normal((B, L, Cin), 0.0, 1.0, 11, 0, dev)
            w = ops.fill_normal((5, Cin, Cout), 0.0, 0.05, 12, 0, dev)
            dy = ops.fill_normal((B, Lout, Cout), 0.0, 1.0, 13, 0, dev)
            out.append(ops.conv1d_fwd(x, w, None, s, pl, Lout, 'relu').cpu().numpy())
            out.append(ops.conv1d_dgrad(dy, ops.conv1d_transpose_w(w), L, s, pl).cpu().numpy())
            dw, db = ops.conv1d_wgrad(x, dy, 5, s, pl)
            out += [dw.cpu().numpy(), db.cpu().numpy()]
        B, L, Cin, Cout, s, padding = _SHAPES[1]
        Lout, pl = ops.conv_geometry(L, 5, s, padding)
        rng = np.random.RandomState(0)
        xi = torch.tensor(rng.randint(-3, 4, (B, L, Cin)), dtype=torch.float32, device=dev)
        wi = torch.tensor(rng.randint(-2, 3, (5, Cin, Cout)), dtype=torch.float32, device=dev)
        dyi = torch.tensor(rng.randint(-2, 3, (B, Lout, Cout)), dtype=torch.float32, device=dev)
        out.append(ops.conv1d_fwd(xi, wi, None, s, pl, Lout).cpu().numpy())
        out.append(ops.conv1d_dgrad(dyi, ops.conv1d_transpose_w(wi), L, s, pl).cpu().numpy())
        out.append(ops.conv1d_wgrad(xi, dyi, 5, s, pl)[0].cpu().numpy())
    return out


def _child(code, env_extra, out):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    return np.load(out) if out.endswith('.npz') else None


@pytest.fixture(scope='module')
def default_results():
    return _direct_cases()


@pytest.mark.parametrize("switch,same_arithmetic", [('GN_CONV_NOMERGE', True), ('GN_CONV_NONARROW', True), ('GN_CONV_NOPATCH', True),
                                                    ('GN_CONV_NOPIPE', False), ('GN_CONV_NODMA', False), ('GN_WGRAD_NOPIPE', False)])
def test_direct_kernel_switch(tmp_path, default_results, switch, same_arithmetic):
    """same_arithmetic: the switch changes which member of the pipelined family runs or in what order the blocks start -- every output is the same fmaf
    chain, so the results are bit-identical.  The fallback FAMILIES pair the channels of a chunk differently (another exact chain): equal to fp32
    rounding on real data, bit-identical on small integers."""
    out = str(tmp_path / 'r.npz')
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; import test_switches_gpu as T; "
            "np.savez(%r, *T._direct_cases())") % (ROOT, os.path.join(ROOT, 'tests'), out)
    got = _child(code, {switch: '1'}, out)
    assert len(got.files) == len(default_results)
    for k, ref in enumerate(default_results):
        a = got['arr_%d' % k]
        if same_arithmetic or k >= len(default_results) - 3:
            assert np.array_equal(a, ref), (switch, k)
        else:
            assert np.abs(a - ref).max() <= 3e-6 * np.abs(ref).max(), (switch, k, float(np.abs(a - ref).max() / np.abs(ref).max()))


def test_opt_in_split_at_its_default_size_gate(tmp_path):
    """the suite forces GN_BF16X3_MIN_GFLOP=0 (tests/conftest.py), so the shipped 50-GFLOP gate of the opt-in bf16 split ran nowhere.  A child
    with the default gate: a 2.7-GFLOP launch stays on the direct kernel, a 54-GFLOP launch of the same layer takes the split kernel; both agree with
    the direct kernels to the split's bound (tests/test_bf16x3_gpu.py: 4e-6 of the largest entry)."""
    out = str(tmp_path / 'gate.npz')
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from gennet_amd import ops
dev = torch.device('cuda')
res = []
for B in (2, 40):
    x = ops.fill_normal((B, 2048, 256), 0.0, 1.0, 21, 0, dev); w = ops.fill_normal((5, 256, 256), 0.0, 0.05, 22, 0, dev)
    ops.set_conv_math('bf16x3', workspace_gb=1.0)
    ops.prof_enable(True); ops.prof_reset()
    y = ops.conv1d_fwd(x, w, None, 1, 2, 2048)
    n_split = ops.prof_collect(2)['launches']
    ops.prof_enable(False)
    ops.set_conv_math('fp32')
    yd = ops.conv1d_fwd(x, w, None, 1, 2, 2048)
    res += [np.array([n_split, float((y - yd).abs().max() / yd.abs().max())])]
np.savez(%r, *res)
""" % (ROOT, out)
    env = {'GENNET_CONV_MATH': 'fp32'}
    e = dict(os.environ); e.pop('GN_BF16X3_MIN_GFLOP', None); e.update(env)
    r = subprocess.run([sys.executable, '-c', code], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    got = np.load(out)
    small, large = got['arr_0'], got['arr_1']
    assert small[0] == 0 and small[1] == 0.0                   # below the gate: the direct kernel, bit for bit
    assert large[0] == 1 and 0.0 < large[1] <= 4e-6


def test_opt_in_split_refuses_a_workspace_that_is_too_small():
    """under the opt-in math a launch whose split operands do not fit the workspace used to run silently on the other kernel; which kernel
    a launch takes now depends on its shape alone, and the shortfall is an error that names the size."""
    from gennet_amd import _lib, ops
    dev = torch.device('cuda')
    x = ops.fill_normal((8, 2048, 256), 0.0, 1.0, 1, 0, dev); w = ops.fill_normal((5, 256, 256), 0.0, 0.05, 2, 0, dev)
    ops.set_conv_math('bf16x3', workspace_gb=0.001)
    try:
        with pytest.raises(_lib.GennetHipError, match='workspace'):
            ops.conv1d_fwd(x, w, None, 1, 2, 2048)
        with pytest.raises(_lib.GennetHipError, match='workspace'):
            ops.conv1d_wgrad(x, ops.fill_normal((8, 2048, 256), 0.0, 1.0, 3, 0, dev), 5, 1, 2)
    finally:
        ops.set_conv_math()


@pytest.mark.parametrize("name", ['_NO_CONVSTATS', '_NO_DROPGEN'])
def test_planner_switch_trains_the_same_step(monkeypatch, name):
    """GN_NO_CONVSTATS (BatchNorm statistics from a separate pass instead of the conv epilogue: the same fp64 sums in another order) and GN_NO_DROPGEN
    (keep-mask from the separate Philox kernel instead of inside the BN apply pass: the same stream) against the default plan: one generator update."""
    from gennet_amd import bbh, engine, layers

    def run(flag):
        monkeypatch.setattr(layers, name, flag)
        engine.set_init_seed(5); engine.set_device_seed(7)
        rng = np.random.RandomState(1)
        nets = bbh.build_and_compile(rng.randn(64, 1).astype(np.float32), 64, do_pe=False)
        z = rng.uniform(-1, 1, (4, 100)).astype(np.float32)
        out = nets.signal_discriminator_on_generator.train_on_batch(z, np.ones(4))
        return out, [p.data.cpu().numpy() for l in nets.generator.layers for p in l.params]

    (l0, w0), (l1, w1) = run(False), run(True)
    assert abs(l0[0] - l1[0]) <= 1e-6 * abs(l0[0]) and l0[1] == l1[1]
    for a, b in zip(w0, w1):
        assert np.abs(a - b).max() <= 1e-4 * np.abs(a).max() + 0.02 * 9e-5
