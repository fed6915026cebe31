"""Data-parallel G+D step: one process per GPU, minibatch sharded over ranks (whole sequences per
rank), parameters / optimizer state replicated.  The reference has no multi-device code at all
(SURVEY 5); this is new, built on ``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" in the CPU tests).

Exchange steps per global step:
  1. global valid-frame count Tv = sum over ranks of mask.sum()  -> loss normaliser on every rank; kept on
     the device, its all-reduce overlaps the generator forward
     (losses divide by the GLOBAL Tv: train.py:258,269-270,286,308; seqloss.py:43);
  2. D step: local forward+backward -> all-reduce(sum) of D's flat gradient + the additive loss /
     count sums (one coalesced launch) -> identical clip-norm + optimizer step on every rank;
  3. G step: same with G's flat gradient (the D->G "leak" gradient stays local: it is a
     per-frame upstream gradient, already normalised by the global Tv).
Gradients are single flat buffers (<= 19 MB), so each exchange is ONE all-reduce per network.

The class only orchestrates; compute goes through a backend object with the split-phase methods
of ``StepEngine`` (``*_begin`` / ``*_end``), which lets the CPU tests drive it with the oracle.
"""
import torch
import torch.distributed as dist


class DataParallelStep(object):
    def __init__(self, backend, process_group=None, always_reduce=False, reduce_via_host=False):
        self.backend = backend
        self.pg = process_group
        self.via_host = reduce_via_host    # device tensors through a CPU-only process group (gloo): tests on one GPU
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.always_reduce = always_reduce and dist.is_initialized()   # exercise the collectives with one rank
        self._pending = None
        # the backend's dropout streams are keyed by GLOBAL sequence (round-robin dealing: local sequence b of rank r is
        # sequence r + world * b of the minibatch), so a world-k run draws the masks of the one-process run
        if self.world > 1 and hasattr(backend, "set_shard"):
            backend.set_shard(dist.get_rank(process_group), self.world)
        self._coalesce = dist.is_initialized() and dist.get_backend(process_group) == "nccl" and \
            hasattr(dist, "_coalescing_manager")

    def _allreduce(self, *tensors):
        """Sum over ranks, in place.  Several tensors (the flat gradient + the few loss sums of the same step)
        go out as ONE coalesced launch where the backend supports it (RCCL: one group call, one latency)."""
        if not (self.world > 1 or self.always_reduce):
            return
        if self.via_host:
            for t in tensors:
                h = t.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
                t.copy_(h)
            return
        if len(tensors) > 1 and self._coalesce:
            try:
                with dist._coalescing_manager(group=self.pg, device=tensors[0].device, async_ops=False):
                    for t in tensors:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
                return
            except Exception:          # backend without coalescing support (gloo): fall back for good
                self._coalesce = False
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def broadcast_parameters(self, *flat_buffers):
        """Rank 0's parameters / optimizer state become everyone's (call once after construction)."""
        if self.world > 1:
            for t in flat_buffers:
                dist.broadcast(t, src=0, group=self.pg)

    def global_valid_frames(self, mask):
        tv = mask.sum().reshape(1).double()
        self._allreduce(tv)
        return float(tv.item())

    def step(self, batch, adv_w=1.0, mse_w=0.0, mge_w=1.0, update_d=True, update_g=True, phase="train",
             tv_global=None, lazy_g=False):
        """One global step on this rank's shard.  ``batch`` is the backend's opaque local batch.
        Returns (d_result or None, g_result or None), identical on all ranks.  ``lazy_g=True`` returns the
        generator scalars as a ``LazyResult`` (no host wait at the end of the step)."""
        be = self.backend
        tv_work = None
        if tv_global is None:
            if getattr(be, "device_normalizer", False) and be.mask_of(batch).is_cuda:
                # the global valid-frame count stays on the device: its all-reduce overlaps the generator
                # forward and nobody synchronises with the host for it
                tv_global = be.mask_of(batch).sum().reshape(1).double()
                if self.via_host:
                    self._allreduce(tv_global)
                elif self.world > 1 or self.always_reduce:
                    tv_work = dist.all_reduce(tv_global, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                tv_global = self.global_valid_frames(be.mask_of(batch))
        be.zero_grad()
        be.apply_generator(batch)
        if self._pending is not None:    # last step's lazy generator scalars: fetched now that this step's forward is queued
            self._pending.get()
            self._pending = None
        if tv_work is not None:
            tv_work.wait()               # stream-level dependency, not a host sync
        be.set_loss_normalizer(tv_global)
        d_res = g_res = None
        train = phase == "train"
        defer = getattr(be, "deferred_results", False)
        if update_d:
            be.update_discriminator_begin(batch, phase)
            if train:
                self._allreduce(be.flat_grads("D"), be.scalar_sums("D"))
            else:
                self._allreduce(be.scalar_sums("D"))
            d_res = be.update_discriminator_end(batch, phase, defer=True) if defer else be.update_discriminator_end(batch, phase)
        if update_g:
            be.update_generator_begin(batch, adv_w, mse_w, mge_w, phase)
        if update_d and defer:
            d_res = be.update_discriminator_result()     # the generator step is already queued behind it
        if update_g:
            if train:
                self._allreduce(be.flat_grads("G"), be.scalar_sums("G"))
            else:
                self._allreduce(be.scalar_sums("G"))
            if defer and lazy_g:
                be.update_generator_end(batch, adv_w, mse_w, mge_w, phase, defer=True)
                g_res = LazyResult(be.update_generator_result)
                self._pending = g_res
            else:
                g_res = be.update_generator_end(batch, adv_w, mse_w, mge_w, phase)
        return d_res, g_res


class LazyResult(object):
    """Generator-step scalars that are fetched (one event wait) on first use: ``tuple(r)``, ``r[i]``, ``r.get()``.
    ``DataParallelStep.step(..., lazy_g=True)`` hands these out so that the host can enqueue the next step while the
    GPU is still finishing this one; the next ``step`` call resolves a still-pending one before it touches the engine."""

    def __init__(self, fetch):
        self._fetch, self._value = fetch, None

    def get(self):
        if self._value is None:
            self._value = tuple(self._fetch())
            self._fetch = None
        return self._value

    def __iter__(self):
        return iter(self.get())

    def __getitem__(self, i):
        return self.get()[i]

    def __len__(self):
        return len(self.get())
