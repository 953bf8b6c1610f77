"""Diagnostic: per-parameter gradient errors of the full-size PE step / GAN G step against the fp64 oracle (prints every tensor)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import test_nets_gpu as T
from oracle import nets_ref as N
from oracle import keras_ref as K


def pe(n_pix, B):
    from gennet_amd import bbh
    from gennet_amd.engine import Adam
    rng = np.random.RandomState(n_pix)
    ref = N.PENet(n_pix, rng)
    T.round_stack(ref.mc); T.round_stack(ref.q)
    ref.mc.params[-1][...] = 25.0; ref.q.params[-1][...] = 0.6
    model = bbh.signal_pe_model(n_pix)
    n_mc = len([s for s in ref.mc.spec if s[0] in ('dense', 'conv1d')])
    wp = [l for l in model.layers if l.weights]
    T.load_stack_into_layers(ref.mc, wp[:n_mc]); T.load_stack_into_layers(ref.q, wp[n_mc:])
    model.compile(loss='mean_squared_error', optimizer=Adam(lr=9e-5, beta_1=0.5), metrics=['accuracy'])
    x = T.f32(rng.randn(B, n_pix, 1)); y_mc = T.f32(rng.uniform(20, 35, B)); y_q = T.f32(rng.uniform(0.5, 1, B))
    cap = {}
    out = model.train_on_batch(x, [y_mc, y_q], capture=cap)
    dec = (T.decisions_for(ref.mc, wp[:n_mc], cap), T.decisions_for(ref.q, wp[n_mc:], cap)) if os.environ.get('INJECT', '1') == '1' else (None, None)
    out_ref = ref.train_on_batch(x, y_mc, y_q, decisions=dec)
    print('decision stats (in band, flipped, outside, layer elements):', ref.mc.decision_stats, ref.q.decision_stats)
    print('PE', n_pix, B, out[:3], out_ref[:3])
    names = [p_.name for l in wp for p_ in l.params]
    grads = [p_.grad.cpu().numpy() for l in wp for p_ in l.params]
    for nm, gq, gr in zip(names, grads, ref.last_grads):
        d = np.abs(gq - gr)
        print('  %-28s shape %-22s rel %.2e  max|ref| %.3e  argmax err %s' % (nm, gq.shape, T.rel(gq, gr), np.abs(gr).max(), np.unravel_index(d.argmax(), d.shape)))


def gan(n_pix, B):
    from gennet_amd import bbh
    rng = np.random.RandomState(3)
    ref, nets, event = T._build_gan(n_pix, rng)
    G, D, DG = nets.generator, nets.signal_discriminator, nets.signal_discriminator_on_generator
    z2 = T.f32(rng.uniform(-1, 1, (B, 100)))
    g_masks = T.stack_masks(ref.G, z2, rng)
    probe = K.mylayer_fwd(ref.G.forward(z2, False), ref.event)
    d_masks2 = T.stack_masks(ref.D, probe, rng)
    names = dict(T.masks_by_name(ref.G, g_masks, G.layers)); names.update(T.masks_by_name(ref.D, d_masks2, D.layers))
    cap = {}
    out = DG.train_on_batch(z2, [1] * B, dropout_masks=names, capture=cap)
    out_ref = ref.g_train_on_batch(z2, [1] * B, g_masks, d_masks2, T.decisions_for(ref.D, D.layers, cap) if os.environ.get('INJECT', '1') == '1' else None)
    print('decision stats (in band, flipped, outside, layer elements):', ref.D.decision_stats)
    print('GAN G step', n_pix, B, out, out_ref)
    pn = [p.name for l in G.layers for p in l.params]
    ggr = [p.grad.cpu().numpy() for l in G.layers for p in l.params]
    for nm, gq, gr in zip(pn, ggr, ref.last_g_grads):
        d = np.abs(gq - gr)
        print('  %-36s shape %-18s rel %.2e  max|ref| %.3e  argmax err %s' % (nm, gq.shape, T.rel(gq, gr), np.abs(gr).max(), np.unravel_index(d.argmax(), d.shape)))


if __name__ == '__main__':
    for a in sys.argv[1:]:
        kind, n_pix, B = a.split(':')
        (pe if kind == 'pe' else gan)(int(n_pix), int(B))
