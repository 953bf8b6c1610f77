"""Device-side randomness: no two consumers may share Philox counters ("no test would notice two consumers sharing a counter
range").  The host allocator (engine.PhiloxStream.take) hands every consumer the next ceil(n / 4) counters of ONE (seed) stream, so disjointness
holds by construction PROVIDED every kernel uses exactly the counters offset .. offset + ceil(n / 4) - 1 for its n values.  That is what is tested,
kernel by kernel, through the concatenation property
        draw(n1 + n2 values at offset)  ==  draw(n1 at offset)  ++  draw(n2 at offset + n1 / 4)          (n1 a multiple of 4)
bit for bit -- a kernel that stepped its counter per value, per row or per block would break it -- and, for the loops, by recording every take()
of one CNN step and one GAN iteration: the ranges tile the stream without gap or overlap, each is exactly as long as its consumer's tensor needs,
and no two draws of the iteration coincide in value.
"""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize("n1,n2", [(4096, 1000), (4, 3), (1 << 20, 12345)])
def test_every_generator_kernel_consumes_exactly_a_quarter_counter_per_value(n1, n2):
    from gennet_amd import ops
    seed, off = 77, 123456789
    for draw in (lambda n, o: ops.fill_uniform((n,), -1.0, 1.0, seed, o, _dev()),
                 lambda n, o: ops.fill_normal((n,), 0.5, 2.0, seed, o, _dev()),
                 lambda n, o: ops.dropout_mask((n,), 0.4, seed, o, _dev())):
        whole = draw(n1 + n2, off)
        a, b = draw(n1, off), draw(n2, off + n1 // 4)
        assert torch.equal(whole, torch.cat([a, b]))
        assert not torch.equal(draw(n2, off + n1 // 4 + 1), b)          # ... and the next counter really is a different draw
    # the fused BatchNormalization apply + dropout kernel draws dropout_mask's stream for the same (seed, offset), whatever the row length
    rows, C = 96, 64
    x = ops.fill_normal((rows, C), 0.0, 1.0, 5, 0, _dev())
    scale = torch.ones(C, device=_dev()); shift = torch.zeros(C, device=_dev())
    _, m = ops.bn_apply_dropgen(x, scale, shift, 'tanh', 0.0, 0.2, seed, off)
    assert torch.equal(m.reshape(-1), ops.dropout_mask((rows * C,), 0.2, seed, off, _dev()))
    _, m2 = ops.bn_apply_dropgen(x[32:].contiguous(), scale, shift, 'tanh', 0.0, 0.2, seed, off + 32 * C // 4)
    assert torch.equal(m2, m[32:])


def test_the_draws_of_one_cnn_step_and_one_gan_iteration_tile_the_stream():
    from gennet_amd import bbh, engine
    n_pix, B = 128, 8
    engine.set_init_seed(3); engine.set_device_seed(42)
    random.seed(3); np.random.seed(3)
    rng = np.random.RandomState(3)
    event = rng.randn(n_pix, 1).astype(np.float32)
    nets = bbh.build_and_compile(event, n_pix)
    bank = bbh.DeviceBank(rng.randn(64, n_pix).astype(np.float32), np.stack([rng.uniform(20, 35, 64), rng.uniform(0.5, 1, 64)], 1))
    ev = engine.to_device(event.reshape(-1))
    stream = engine.device_rng()
    taken = []
    plain = stream.take

    def recording_take(n_values):
        seed, off = plain(n_values)
        taken.append((off, int(n_values)))
        return seed, off
    plain_rows = stream.take_rows

    def recording_take_rows(row_len, blocks, global_rows):          # the loop bodies draw row-tiled (one block = the whole batch in a single process)
        seed, offs = plain_rows(row_len, blocks, global_rows)
        assert len(blocks) == 1 and blocks[0] == (0, global_rows) and offs[0] == stream.offset - (global_rows * row_len + 3) // 4
        taken.append((offs[0], int(global_rows * row_len)))
        return seed, offs
    stream.take = recording_take
    stream.take_rows = recording_take_rows
    try:
        bbh.pe_train_step(nets.signal_pe, bank, B)
        bbh.gan_train_step(nets, bank, ev, B)
    finally:
        del stream.take
        del stream.take_rows
    # consumers of one iteration: CNN noise rows; z, noise column, D's two dropout layers, z, G's six dropout layers + D's two inside the combined model
    assert len(taken) == 1 + 2 + 2 + 1 + 6 + 2, taken
    pos = 0
    for off, n in taken:
        assert off == pos, (taken, 'gap or overlap in the counter stream')
        pos += (n + 3) // 4
    assert stream.offset == pos
    sizes = [n for _, n in taken]
    assert sizes[0] == int(B / 8) * n_pix and sizes[1] == B * 100 and sizes[2] == B * n_pix
    assert sizes[3] == 2 * B * (n_pix // 2) * 2 * 256 and sizes[4] == 2 * B * (n_pix // 4) * 2 * 512           # D's dropout masks on 2B rows
    assert sizes[5] == B * 100
    assert sizes[6:12] == [B * 256 * (n_pix // 2), B * (n_pix // 2) * 64, B * n_pix * 128, B * n_pix * 256, B * n_pix * 512, B * n_pix * 1024]
    assert sizes[12:] == [B * (n_pix // 2) * 2 * 256, B * (n_pix // 4) * 2 * 512]
    # two latent draws of the same iteration come from different counters: position by position they differ (one chance coincidence allowed)
    from gennet_amd import ops
    z1 = ops.fill_uniform((B * 100,), -1.0, 1.0, stream.seed, taken[1][0], _dev())
    z2 = ops.fill_uniform((B * 100,), -1.0, 1.0, stream.seed, taken[5][0], _dev())
    assert int((z1 == z2).sum()) <= 1
