"""world_size-2 data-parallel test on CPU (gloo): drives gantts_amd.parallel.DataParallelStep --
the same orchestration bench.py uses over RCCL -- with an oracle-backed compute backend, and checks
that DP=2 on a sharded batch reproduces the single-process result on the whole batch
(losses, counts, parameters after several steps)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases as C
import gantts_oracle as O
from oracle_runner import build_oracle_model, stream_config


class OracleBackend(object):
    """Implements the backend protocol of DataParallelStep with the CPU oracle.  Loss pieces are
    additive sums normalised by the GLOBAL valid-frame count, exactly like the HIP engine's
    split-phase entry points (gt_update_*_begin / _end)."""

    def __init__(self, case):
        self.case, self.cfg = case, stream_config(case)
        self.mg, self.md = build_oracle_model(case["g"], 11), build_oracle_model(case["d"], 22)
        self.mg.training = self.md.training = False
        self.og = O.make_optimizer(case["opt_g"][0], self.mg.params, **case["opt_g"][1])
        self.od = O.make_optimizer(case["opt_d"][0], self.md.params, **case["opt_d"][1])
        self.tv = None
        self._flat = {}
        self._sums = {"D": torch.zeros(4, dtype=torch.float64), "G": torch.zeros(3, dtype=torch.float64)}

    def mask_of(self, batch):
        return batch["mask"]

    def set_loss_normalizer(self, tv):
        self.tv = tv

    def zero_grad(self):
        self.og.zero_grad(), self.od.zero_grad()

    def apply_generator(self, batch):
        self.out = O.apply_generator(self.cfg, self.mg, batch["x"], batch["R"], batch["lengths"])

    def _pack(self, which, params):
        self._flat[which] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                       for p in params]).clone()

    def _unpack(self, which, params):
        off = 0
        for p in params:
            p.grad = self._flat[which][off:off + p.numel()].view_as(p).clone()
            off += p.numel()

    def flat_grads(self, which):
        return self._flat[which]

    def scalar_sums(self, which):
        return self._sums[which]

    def update_discriminator_begin(self, batch, phase):
        r = O.update_discriminator(self.cfg, self.md, self.od, batch["x"], batch["y_static"], self.out[1],
                                   batch["lengths"], batch["mask"], phase, Tv=self.tv, step=False)
        # additive pieces: loss_real*Tv, loss_fake*Tv, counts
        self._sums["D"][:] = torch.tensor([r[2] * self.tv, r[1] * self.tv, r[3], r[4]], dtype=torch.float64)
        self._pack("D", self.md.params)

    def update_discriminator_end(self, batch, phase):
        if phase == "train":
            self._unpack("D", self.md.params)
            O.clip_grad_norm_(self.md.params, 1.0)
            self.od.step()
        s = self._sums["D"]
        lr, lf = float(s[0]) / self.tv, float(s[1]) / self.tv
        return lr + lf, lf, lr, float(s[2]), float(s[3])

    def update_generator_begin(self, batch, adv_w, mse_w, mge_w, phase):
        r = O.update_generator(self.cfg, self.mg, self.md, self.og, batch["x"], batch["y"], self.out[0],
                               batch["y_static"], self.out[1], adv_w, batch["lengths"], batch["mask"], phase,
                               mse_w=mse_w, mge_w=mge_w, Tv=self.tv, step=False)
        self._sums["G"][:] = torch.tensor([r[2] * self.tv, r[1] * self.tv, r[0] * self.tv], dtype=torch.float64)
        self._pack("G", self.mg.params)

    def update_generator_end(self, batch, adv_w, mse_w, mge_w, phase):
        if phase == "train":
            self._unpack("G", self.mg.params)
            O.clip_grad_norm_(self.mg.params, 1.0)
            self.og.step()
        s = self._sums["G"]
        adv, mge, mse = float(s[0]) / self.tv, float(s[1]) / self.tv, float(s[2]) / self.tv
        return mse, mge, adv, (mse_w * mse + mge_w * mge) + adv_w * adv


class DeferredOracleBackend(OracleBackend):
    """Same compute, but with the deferred-result protocol of the HIP backend (``*_end(defer=True)`` enqueues,
    ``*_result()`` collects): lets the CPU tests drive the non-blocking schedule of DataParallelStep."""
    deferred_results = True

    def __init__(self, case):
        OracleBackend.__init__(self, case)
        self._held = {}
        self.trace = []

    def update_discriminator_begin(self, batch, phase):
        self.trace.append("d_begin")
        OracleBackend.update_discriminator_begin(self, batch, phase)

    def update_generator_begin(self, batch, adv_w, mse_w, mge_w, phase):
        self.trace.append("g_begin")
        OracleBackend.update_generator_begin(self, batch, adv_w, mse_w, mge_w, phase)

    def apply_generator(self, batch):
        self.trace.append("apply_g")
        OracleBackend.apply_generator(self, batch)

    def update_discriminator_end(self, batch, phase, defer=False):
        r = OracleBackend.update_discriminator_end(self, batch, phase)
        if not defer:
            return r
        self._held["D"] = r

    def update_discriminator_result(self):
        self.trace.append("d_result")
        return self._held.pop("D")

    def update_generator_end(self, batch, adv_w, mse_w, mge_w, phase, defer=False):
        r = OracleBackend.update_generator_end(self, batch, adv_w, mse_w, mge_w, phase)
        if not defer:
            return r
        self._held["G"] = r

    def update_generator_result(self):
        self.trace.append("g_result")
        return self._held.pop("G")


def make_batch(case, rows):
    x_np, y_np, lengths = C.make_batch(case)
    cfg = stream_config(case)
    x, y = torch.from_numpy(x_np[rows]), torch.from_numpy(y_np[rows])
    lens = list(lengths[rows])
    T = case["T"]
    R = torch.from_numpy(O.unit_variance_mlpg_matrix(C.WINDOWS[:case["windows"]], T))
    return dict(x=x, y=y, lengths=lens, R=R, mask=O.sequence_mask(lens, T).unsqueeze(-1),
                y_static=O.get_static_features(y, cfg.num_windows, cfg.stream_sizes, cfg.has_dynamic_features))


CASE = "acoustic_mlp"
STEPS = 3


def _worker(rank, world, port, q, deferred=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gantts_amd.parallel import DataParallelStep
    case = C.CASES[CASE]
    be = DeferredOracleBackend(case) if deferred else OracleBackend(case)
    dp = DataParallelStep(be)
    rows = np.arange(case["B"])[rank::world]          # deal sequences round-robin (length-sorted batch)
    batch = make_batch(case, rows)
    hist = []
    for _ in range(STEPS):
        d, g = dp.step(batch, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"], lazy_g=deferred)
        hist.append((d, g))
    hist = [(d, tuple(g)) for d, g in hist]
    q.put((rank, hist, [p.detach().numpy().copy() for p in be.mg.params + be.md.params]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_deferred_schedule_and_lazy_results_single_process():
    """The non-blocking schedule DataParallelStep uses with a deferred-result backend: the D result is collected
    after the G step has been queued, the lazy G result after the NEXT step's forward; numbers unchanged."""
    from gantts_amd.parallel import DataParallelStep, LazyResult
    case = C.CASES[CASE]
    whole = make_batch(case, np.arange(case["B"]))
    ref = DataParallelStep(OracleBackend(case))
    ref_hist = [ref.step(whole, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"]) for _ in range(2)]
    be = DeferredOracleBackend(case)
    dp = DataParallelStep(be)
    d0, g0 = dp.step(whole, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"], lazy_g=True)
    assert isinstance(g0, LazyResult) and g0._value is None
    assert be.trace == ["apply_g", "d_begin", "g_begin", "d_result"]
    d1, g1 = dp.step(whole, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"], lazy_g=True)
    assert g0._value is not None                       # resolved by the second step ...
    assert be.trace[4:6] == ["apply_g", "g_result"]    # ... right after its forward was queued
    assert (d0, tuple(g0)) == ref_hist[0] and (d1, tuple(g1)) == ref_hist[1]
    assert len(g1) == 4 and g1[3] == ref_hist[1][1][3]
    # eager mode with the same backend blocks inside the step
    d2, g2 = dp.step(whole, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"])
    assert isinstance(g2, tuple)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("deferred", [False, True])
def test_dp2_equals_single_process_on_whole_batch(deferred):
    case = C.CASES[CASE]
    # single process, whole batch, same orchestration with world = 1
    from gantts_amd.parallel import DataParallelStep
    be = OracleBackend(case)
    dp = DataParallelStep(be)
    whole = make_batch(case, np.arange(case["B"]))
    ref_hist = [dp.step(whole, adv_w=case["adv_w"], mse_w=case["mse_w"], mge_w=case["mge_w"]) for _ in range(STEPS)]
    ref_params = [p.detach().numpy().copy() for p in be.mg.params + be.md.params]
    # and that orchestration itself equals the plain reference-shaped step (golden fixture)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", CASE + ".npz"))
    for i in range(STEPS):
        np.testing.assert_allclose(ref_hist[i][0], gold["d_scalars_%d" % i], rtol=1e-5)
        np.testing.assert_allclose(ref_hist[i][1], gold["g_scalars_%d" % i], rtol=1e-5)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, deferred)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r[0])
    for rank, hist, params in results:
        for i in range(STEPS):
            np.testing.assert_allclose(hist[i][0], ref_hist[i][0], rtol=2e-5, err_msg="rank %d step %d D" % (rank, i))
            np.testing.assert_allclose(hist[i][1], ref_hist[i][1], rtol=2e-5, err_msg="rank %d step %d G" % (rank, i))
        for a, b in zip(params, ref_params):
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-6)
    # replicas stay bit-identical to each other (same reduced gradients, same update)
    for a, b in zip(results[0][2], results[1][2]):
        assert np.array_equal(a, b)
