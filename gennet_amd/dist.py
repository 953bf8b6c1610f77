"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for CPU tests).

The reference has no distributed code at all; this is the batch-sharded scheme of SURVEY section 8e:
rank r trains on rows [r*B/N, (r+1)*B/N) of the global batch and three kinds of sums cross the ranks:
  1. the flat gradient buffer of the compiled model, once per train_on_batch (one all-reduce: PE 19.7 MB, D 15.2 MB,
     G 122 MB fp32 at n_pix = 2048) -- losses are already normalised by the GLOBAL batch size, so SUM is the reduction;
  2. BatchNorm statistics (sum x, sum x^2) and their backward counterparts, fp64, in the generator only (SyncBN);
  3. the loss / accuracy scalars.
xGMI is point-to-point (7 links per GPU): a ring all-reduce of S bytes moves 2*(N-1)/N*S per GPU over one link, ~1.4 ms
for the 122 MB generator gradient, against a >100 ms step -- so one large flat all-reduce per step is the right shape and
bucketing/overlap would buy nothing here.
"""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
    def __init__(self, group=None, always=False):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised; call gennet_amd.dist.init() first')
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.always = always               # issue the collectives even in a one-rank group (init(allow_single=True))
        # accounting of the exchange step: calls and bytes always; with `timing` every all-reduce is bracketed by HIP events on the launch
        # stream (the collective itself runs on RCCL's stream, which the launch stream waits for), read by collect()
        self.calls = 0
        self.bytes = 0
        self.timing = False
        self._events = []

    def reset_counters(self):
        self.calls, self.bytes, self._events = 0, 0, []

    def all_reduce_sum(self, t):
        if self.world_size > 1 or self.always:
            self.calls += 1
            self.bytes += t.numel() * t.element_size()
            timed = self.timing and t.is_cuda
            if timed:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            if timed:
                e1.record()
                self._events.append((e0, e1))
        return t

    def collect(self):
        """{'calls', 'bytes', 'ms'}: all-reduces since reset_counters(); ms = summed event-bracketed time on the launch stream (None unless
        `timing` was on) -- the exchange is not overlapped with compute, so this is its whole cost to the step."""
        ms = None
        if self._events:
            torch.cuda.synchronize()
            ms = float(sum(a.elapsed_time(b) for a, b in self._events))
        return {'calls': self.calls, 'bytes': self.bytes, 'ms': ms}

    def broadcast(self, t, src=0):
        if self.world_size > 1 or self.always:
            dist.broadcast(t, src, group=self.group)
        return t

    def sync_model(self, model):
        """Make every rank start from rank 0's weights (parameters and BN moving statistics)."""
        for p in model.weights:
            self.broadcast(p.data)


def expected_collectives(nets, cnn_steps=2):
    """What ONE bench step (cnn_steps x CNN train_on_batch + one GAN iteration, bbh.pe_train_step / gan_train_step) must put through
    all_reduce_sum, from the models' own shapes -- so that the first real N > 1 run validates itself against DataParallel's counters:
      * per train_on_batch of a compiled model: its trainable parameters, each padded to the flat buffer's 256-byte granule, fp32 (SUM of
        the gradients; n_pix 2048: CNN 19.7 MB, D 15.2 MB, G 122 MB) and one (n_outputs, 2) fp32 block of loss / accuracy sums;
      * per BatchNormalization of the generator inside the combined model: (sum x, sum x^2) forward and (sum g, sum g xhat) backward, 2 C fp64
        each (SyncBN: the first one reduces over the batch axis only, bbhMahoGANy.py:235).
    Returns {'calls', 'bytes', 'parts': {...}}."""
    from .engine import ParamGroup, segments
    from .layers import BatchNormalization
    gran = ParamGroup.ALIGN

    def grad_bytes(model):
        return 4 * sum(-(-p.size // gran) * gran for p in model._train_params)
    parts, calls = {}, 0
    for name, model, n in (('cnn', nets.signal_pe, cnn_steps), ('discriminator', nets.signal_discriminator, 1),
                           ('generator_through_frozen_discriminator', nets.signal_discriminator_on_generator, 1)):
        if model is None or n == 0:
            continue
        parts[name + '_gradients'] = n * grad_bytes(model)
        parts[name + '_loss_scalars'] = n * len(model.output_ids) * 2 * 4
        bound = all(getattr(p, 'group', None) is not None for p in model._train_params)       # flat buffers exist after the first device step
        calls += n * ((len(segments(model._train_params)) if bound else 1) + 1)
    bns = [l for l in nets.generator.layers if isinstance(l, BatchNormalization)]
    parts['syncbn_sums'] = sum(2 * (2 * int(l.gamma.size) * 8) for l in bns)
    calls += 2 * len(bns)
    return {'calls': calls, 'bytes': int(sum(parts.values())), 'parts': parts}


def init(backend=None, allow_single=False):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns a DataParallel, or None when WORLD_SIZE is 1 or unset (allow_single=True initialises a one-rank group anyway: the
    collectives then really go through the backend, which is how the RCCL path is exercised on a one-GPU box)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not (allow_single and 'MASTER_PORT' in os.environ):
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            raise RuntimeError('WORLD_SIZE=%d but MASTER_PORT is unset: start the ranks with torch.distributed.run (bench.py --gpus N does)' % world)
        if backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(backend=backend, rank=int(os.environ.get('RANK', '0')), world_size=world)
    return DataParallel(always=allow_single and world <= 1)
