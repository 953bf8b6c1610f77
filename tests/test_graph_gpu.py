"""hipGraph capture of the two loop bodies (bbh.GraphedPEStep / GraphedGANStep; engine.StepGraph): a replayed graph must return, step after
step, EXACTLY what the un-captured loop returns -- same host index stream, same Philox positions (dropout masks, latents, noise), same Adam
and BatchNormalization step counts -- and leave bit-identical weights and moving statistics behind."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_pix, seed):
    from gennet_amd import bbh, engine
    engine.set_init_seed(seed); engine.set_device_seed(100 + seed)
    random.seed(seed); np.random.seed(seed)
    rng = np.random.RandomState(seed)
    event = rng.randn(n_pix, 1).astype(np.float32)
    nets = bbh.build_and_compile(event, n_pix)
    bank = bbh.DeviceBank(rng.randn(64, n_pix).astype(np.float32), np.stack([rng.uniform(20, 35, 64), rng.uniform(0.5, 1, 64)], 1))
    return nets, bank, engine.to_device(event.reshape(-1))


def _weights(model):
    return [w.copy() for w in model.get_weights()]


@pytest.mark.parametrize("n_pix,B", [(256, 8), (1024, 8)])
def test_graphed_gan_step_is_bit_identical_to_the_eager_loop(n_pix, B):
    from gennet_amd import bbh
    steps = 6
    nets, bank, ev = _setup(n_pix, 3)
    eager = [bbh.gan_train_step(nets, bank, ev, B) for _ in range(steps)]
    w_eager = _weights(nets.generator) + _weights(nets.signal_discriminator)
    nets, bank, ev = _setup(n_pix, 3)
    step = bbh.GraphedGANStep(nets, bank, ev, B)
    graphed = [step() for _ in range(steps)]
    assert step.sg is not None and step.calls == steps                       # steps 2.. really were graph replays
    assert graphed == eager, (graphed, eager)
    w_graph = _weights(nets.generator) + _weights(nets.signal_discriminator)
    assert all(np.array_equal(a, b) for a, b in zip(w_eager, w_graph))        # incl. BatchNormalization moving statistics
    assert nets.signal_discriminator.optimizer.iterations == steps and nets.signal_discriminator_on_generator.optimizer.iterations == steps
    # without the loss read-out (no per-step synchronisation) the state still advances identically
    nets2, bank2, ev2 = _setup(n_pix, 3)
    step2 = bbh.GraphedGANStep(nets2, bank2, ev2, B)
    for _ in range(steps - 1):
        step2(want_losses=False)
    assert step2() == eager[-1]


def test_graphed_pe_step_is_bit_identical_to_the_eager_loop():
    from gennet_amd import bbh
    n_pix, B, steps = 256, 8, 7
    nets, bank, _ = _setup(n_pix, 5)
    eager = [bbh.pe_train_step(nets.signal_pe, bank, B) for _ in range(steps)]
    w_eager = _weights(nets.signal_pe)
    nets, bank, _ = _setup(n_pix, 5)
    step = bbh.GraphedPEStep(nets.signal_pe, bank, B)
    graphed = [step() for _ in range(steps)]
    assert graphed == eager, (graphed, eager)
    assert all(np.array_equal(a, b) for a, b in zip(w_eager, _weights(nets.signal_pe)))
    # the eager loop can take over from a graphed one (host counters and stream positions are where they should be)
    nets, bank, _ = _setup(n_pix, 5)
    step = bbh.GraphedPEStep(nets.signal_pe, bank, B)
    mixed = [step() for _ in range(4)] + [bbh.pe_train_step(nets.signal_pe, bank, B) for _ in range(steps - 4)]
    assert mixed == eager


def test_two_graphs_share_the_device_random_stream():
    """A CNN graph and a GAN graph replayed alternately (what a script with both loops interleaved would do) equal the eager interleaving."""
    from gennet_amd import bbh
    n_pix, B = 128, 4
    nets, bank, ev = _setup(n_pix, 7)
    eager = []
    for _ in range(4):
        eager.append(bbh.pe_train_step(nets.signal_pe, bank, B)); eager.append(bbh.gan_train_step(nets, bank, ev, B))
    nets, bank, ev = _setup(n_pix, 7)
    pe, gan = bbh.GraphedPEStep(nets.signal_pe, bank, B), bbh.GraphedGANStep(nets, bank, ev, B)
    got = []
    for _ in range(4):
        got.append(pe()); got.append(gan())
    assert got == eager


def test_capture_while_the_launch_profiler_is_on():
    """bench.py switches the HIP-event launch profiler on around its timed region; a graph captured inside that region (--warmup < 2) must not
    record those events: StepGraph.capture suspends the profiler, and the replayed losses still equal the eager ones."""
    from gennet_amd import bbh, ops
    n_pix, B = 128, 4
    nets, bank, _ = _setup(n_pix, 11)
    eager = [bbh.pe_train_step(nets.signal_pe, bank, B) for _ in range(4)]
    nets, bank, _ = _setup(n_pix, 11)
    step = bbh.GraphedPEStep(nets.signal_pe, bank, B)
    ops.prof_enable(True); ops.prof_reset()
    try:
        got = [step() for _ in range(4)]
        assert ops.prof_enabled()
    finally:
        ops.prof_enable(False)
    assert got == eager
    assert sum(ops.prof_collect(k)['launches'] for k in (0, 5, 7)) > 0      # the eager first call was profiled (direct + transform-domain conv launches); the capture was not (it would have failed)


def test_a_captured_graph_keeps_its_scratch_buffer_alive():
    """a captured graph holds the RAW address of ops.workspace()'s scratch; a later, larger eager request replaces that buffer.  The
    graph must keep the one it captured alive (StepGraph.scratch), so replays after the replacement still equal the eager loop bit for bit --
    here the old buffer is dropped from ops._ws, torch's cache emptied and the freed range deliberately overwritten by a new allocation."""
    import gc
    from gennet_amd import bbh, engine, ops
    n_pix, B, steps = 256, 8, 6
    nets, bank, _ = _setup(n_pix, 13)
    eager = [bbh.pe_train_step(nets.signal_pe, bank, B) for _ in range(steps)]
    nets, bank, _ = _setup(n_pix, 13)
    step = bbh.GraphedPEStep(nets.signal_pe, bank, B)
    got = [step() for _ in range(3)]                                     # eager, capture + replay, replay
    assert step.sg is not None and len(step.sg.scratch) >= 1
    key = (engine.device().type, engine.device().index)
    old = ops._ws[key]
    assert any(b is old for b in step.sg.scratch)
    old_ptr, old_bytes = old.data_ptr(), old.numel()
    big = ops.workspace(old_bytes * 3, engine.device())                  # what a later, larger eager layer would ask for
    assert ops._ws[key] is big and big.data_ptr() != old_ptr
    del old
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    junk = [torch.full((old_bytes // 4,), float('nan'), device=engine.device()) for _ in range(3)]      # would land on a freed buffer
    assert all(j.data_ptr() != old_ptr for j in junk)                    # ... but the graph still owns it
    got += [step() for _ in range(steps - 3)]
    assert got == eager, (got, eager)
