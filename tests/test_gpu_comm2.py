"""-m gpu: the engine's OWN data-parallel communicator (gt_comm_*, gantts_amd/csrc/eng_comm.hip) with TWO ranks.

RCCL refuses two ranks on one device and the test boxes have one MI355X, so the two processes bind a test double
(tests/fake_rccl.cpp through GT_RCCL_LIB: all-reduce over POSIX shared memory, stream-ordered, summed in rank order).
Everything above the seven nccl* symbols is the production code path: global valid-frame count, per-layer gradient
buckets merged into few messages on the communicator's stream under the backward pass, the event ring, early (global)
loss results, the join before clip + optimizer.  Each rank holds half the sequences (round-robin) and must
return the WHOLE batch's scalars and end with the whole-batch parameters of the reference-generated fixture; replicas
must be bit-identical.  Reference step being sharded: train.py:538-585."""
import os
import subprocess

import numpy as np
import pytest

import cases as C

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
FAKE = os.path.join(HERE, "libfake_rccl.so")


def build_fake_rccl():
    """hipcc / arch as the library's Makefile takes them (HIPCC, ARCH from the environment, same defaults)."""
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.isfile(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.isfile("/opt/rocm/bin/hipcc") else "hipcc")
        arch = os.environ.get("ARCH", "gfx950")
        subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=" + arch, src, "-o", FAKE, "-lrt"])
    return FAKE


def _worker(rank, world, case, idq, outq, philox=False, ipc_qs=None):
    os.environ["GT_RCCL_LIB"] = FAKE
    # both ranks of these tests share ONE device (one L2): a coarse-grained interprocess arena is coherent there, so the export may
    # fall back to it where the runtime has no fine-grained memory (gt_comm_ipc_export refuses that fallback by default)
    os.environ["GT_IPC_ALLOW_COARSE"] = "1"
    os.environ.pop("GT_COMM_FORCE_COLLECTIVES", None)
    import sys
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "oracle"), HERE, GOLDEN):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import torch
        from gantts_amd.engine import StepEngine
        from hip_runner import run_hip_case
        torch.cuda.set_device(0)
        if rank == 0:
            cid = StepEngine.comm_unique_id()
            for _ in range(world - 1):
                idq.put(cid)
        else:
            cid = idq.get(timeout=120)
        extra = {}
        held = {}
        if ipc_qs is not None:       # the two-shot all-reduce over hipIpc arenas: every rank exports, the handles cross, every rank attaches
            from gantts_amd import _lib as L

            def attach(eng):
                mine = eng.comm_ipc_export()
                for r in range(world):
                    if r != rank:
                        ipc_qs[r].put((rank, mine))
                handles = {rank: mine}
                while len(handles) < world:
                    r, h = ipc_qs[rank].get(timeout=120)
                    handles[r] = h
                eng.comm_ipc_attach(rank, world, b"".join(handles[r] for r in range(world)))
                assert len(mine) == L.IPC_HANDLE_BYTES
                held["eng"] = eng
            extra["after_comm"] = attach
        got = run_hip_case(case, shard=(rank, world), comm_id=cid, extra=extra, philox=philox)
        torch.cuda.synchronize()
        if ipc_qs is not None:
            held["eng"].check_faults()
            got["ipc_messages"] = np.asarray(held["eng"].comm_ipc_messages())
        got["philox_g"], got["philox_d"] = extra["philox_g"], extra["philox_d"]
        outq.put((rank, None, got))
    except Exception as e:      # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        outq.put((rank, "%s\n%s" % (e, traceback.format_exc()), None))


def _run_world2(case, philox=False, ipc=False):
    import torch.multiprocessing as mp
    try:
        build_fake_rccl()
    except Exception as e:      # noqa: BLE001 -- test infrastructure only: no double, no two-rank run on a one-GPU box
        pytest.skip("the RCCL test double could not be built: %s" % e)
    ctx = mp.get_context("spawn")
    idq, outq = ctx.Queue(), ctx.Queue()
    ipc_qs = [ctx.Queue(), ctx.Queue()] if ipc else None
    procs = [ctx.Process(target=_worker, args=(r, 2, case, idq, outq, philox, ipc_qs)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in procs:
            rank, err, got = outq.get(timeout=420)
            if ipc and err is not None and ("hipIpcGetMemHandle" in err or "hipIpcOpenMemHandle" in err):
                # the platform does not hand out / open interprocess handles here (e.g. HSA_ENABLE_IPC_MODE_LEGACY not 0 on a dmabuf-only
                # driver): the arenas cannot exist, which is an environment property, not a defect of the collective
                pytest.skip("hipIpc handles are not available on this box: %s" % err.splitlines()[0])
            assert err is None, "rank %d failed: %s" % (rank, err)
            results[rank] = got
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return results[0], results[1]


def _check(name, r0, r1, ref):
    from test_gpu_parity import _close, _close_state
    for rank, got in ((0, r0), (1, r1)):
        for k in ref:
            if k.startswith("g_leak_norm") or k in ("y_hat", "y_hat_static"):      # outputs are per-shard
                continue
            tag = "%s rank %d %s" % (name, rank, k)
            if "scalars" in k:
                _close(got[k], ref[k], msg=tag)
                if k.startswith("d_scalars"):
                    assert got[k][3] == ref[k][3] and got[k][4] == ref[k][4], tag       # GLOBAL counts, exact
            elif ".opt." in k:
                _close_state(got[k], ref[k], tag)
            else:
                _close(got[k], ref[k], msg=tag)
    for k in r0:
        if k.startswith(("G.", "D.")):
            assert np.array_equal(r0[k], r1[k]), "replicas differ in %s" % k           # bit-identical steps on every rank
    # the ranks hold different rows of the minibatch's ONE mask: a world-W batch has W x the distinct row masks
    for k in ("philox_g", "philox_d"):
        assert not np.array_equal(r0[k], r1[k])
        assert abs(float(r0[k].mean()) - 0.5) < 0.06 and abs(float(r1[k].mean()) - 0.5) < 0.06


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["acoustic_mlp", "acoustic_mlp_dropout", "acoustic_lstm", "vc_in2out", "duration_mlp", "acoustic_lstm_d"])
def test_engine_communicator_world_2_equals_whole_batch_reference_golden(name):
    """MLP (one merged upper-layer message + the first layer's), MLP with injected dropout (masks sharded by sequence),
    BiLSTM (a flush per layer: more collectives per step than the 8-event ring holds), In2Out (gate + MLPG inside the
    model), duration (R = None, Adam), a recurrent DISCRIMINATOR (its layers' gradients leave as one message): sharded over two ranks ==
    the reference's whole-batch fixture."""
    case = C.CASES[name]
    r0, r1 = _run_world2(case)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    _check(name, r0, r1, {k: gold[k] for k in gold.files})


class _forced_fused_dstack(object):
    """GT_FUSED_DSTACK=2 for engines created inside (the spawned ranks inherit the environment): the fused discriminator stack is
    taken by default only for passes with a 32-frame panel per CU, which no small case has."""

    def __enter__(self):
        self.old = os.environ.get("GT_FUSED_DSTACK")
        os.environ["GT_FUSED_DSTACK"] = "2"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("GT_FUSED_DSTACK", None)
        else:
            os.environ["GT_FUSED_DSTACK"] = self.old


@pytest.mark.timeout(900)
def test_engine_communicator_world_2_with_the_fused_discriminator_stack():
    """The fused discriminator stack (dstack_f32.hip.h) under the engine's communicator with two ranks: conditioned 3 x 128 MLP D with
    injected dropout masks (sharded by sequence), 69 rows per rank and pass half (ragged last panel): the unnormalised seeds of
    GT_OPT_COMM_TV_IN_SUMS, the head's partial sums handed to the collective, the generator step's backward-data chain -- sharded over
    two ranks == the reference's whole-batch fixture, replicas bit-identical."""
    case = C.CASES["acoustic_chain_d"]
    with _forced_fused_dstack():
        r0, r1 = _run_world2(case)
    gold = np.load(os.path.join(GOLDEN, "acoustic_chain_d.npz"))
    _check("acoustic_chain_d/fused", r0, r1, {k: gold[k] for k in gold.files})


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["acoustic_lstm_dropout", "acoustic_sru_dropout"])
def test_engine_communicator_world_2_equals_whole_batch_oracle(name):
    """3-layer BiLSTM with inter-layer dropout and the SRU generator with both variational dropouts (oracle-only cases,
    masks injected and sharded): per-layer flushes under the recurrences."""
    from oracle_runner import run_oracle_case
    case = C.ORACLE_ONLY_CASES[name]
    r0, r1 = _run_world2(case)
    _check(name, r0, r1, run_oracle_case(case))


PHILOX_CASES = {
    # MLP G + MLP D, both with Philox dropout; T = 48: whole 16-row groups per sequence, 64-row GEMM tiles straddle sequences
    "mlp": dict(C.CASES["acoustic_mlp_dropout"], B=4, T=48),
    # 3-layer BiLSTM with nn.LSTM inter-layer dropout (a per-element dropout kernel) + dropout-free D
    "lstm": dict(C.ORACLE_ONLY_CASES["acoustic_lstm_dropout"], B=4, T=32),
    # SRU with both variational dropouts (masks per (sequence, column), counted globally) + an MLP D with row dropout (T % 16 == 0)
    "sru": dict(C.ORACLE_ONLY_CASES["acoustic_sru_dropout"], B=4, T=32),
    # conditioned 3 x 128 MLP D through the FUSED discriminator stack (forced): its Philox sites map the panel's 16-row groups to the
    # groups the same frames have in the one-process minibatch (two-half D pass included)
    # (warm Adagrad accumulators: the cold first step lr * g / (|g| + 1e-10) turns the rounding of a near-zero gradient element -- the two
    # world sizes sum it in different orders -- into a parameter error of the size of the tolerance; one element of 16 384 sat at 1.55x)
    "mlp_fused_dstack": dict(C.CASES["acoustic_chain_d"], B=4, T=48,
                             opt_g=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4)),
                             opt_d=("Adagrad", dict(lr=0.01, weight_decay=1e-7, initial_accumulator_value=1e-4))),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", sorted(PHILOX_CASES))
def test_engine_communicator_philox_world_2_equals_world_1(name):
    """The PRODUCTION dropout path under data parallelism (VERDICT r3 missing #1; SURVEY 8(e): "RNG keyed by global sequence
    index so DP=k reproduces DP=1"): Philox ON, nothing injected.  Two ranks with half the sequences each (round-robin)
    must reproduce the one-process run of the whole minibatch -- the reference draws ONE mask over the whole minibatch
    (gantts/models.py:139 inside train.py:538-585): global scalars (counts exact), parameters and optimizer state at the
    suite's 1e-4, replicas bit-identical, and each rank's keep masks == its rows of the one-process masks."""
    from hip_runner import run_hip_case
    case = PHILOX_CASES[name]
    import contextlib
    with (_forced_fused_dstack() if name == "mlp_fused_dstack" else contextlib.nullcontext()):
        r0, r1 = _run_world2(case, philox=True)
        extra = {}
        ref = run_hip_case(case, extra=extra, philox=True)          # world 1, same seed, same step counter
    _check(name, r0, r1, ref)
    B, T = case["B"], case["T"]
    if name != "sru":                                            # (the SRU sites are per-sequence masks, not row masks)
        for k, halves in (("philox_g", 1), ("philox_d", 2)):
            whole = extra[k].reshape(halves, B, T, -1)
            for rank, got in ((0, r0), (1, r1)):
                assert np.array_equal(got[k].reshape(halves, B // 2, T, -1), whole[:, rank::2]), (k, rank)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["acoustic_mlp_dropout", "acoustic_lstm"])
def test_two_shot_ipc_allreduce_world_2_equals_whole_batch_reference_golden(name):
    """SURVEY 8(e)'s small-message collective (gantts_amd/csrc/eng_ipc.hip: gt_comm_ipc_export / _attach): two processes export their
    arenas, exchange the hipIpc handles and attach; from then on EVERY message of the step (the five- and three-double sums, the
    gradient buckets) is reduced by the engine's own publish / reduce-my-chunk-and-push / collect launches over the peers' arenas --
    the RCCL double is attached too but must not be needed.  Same bar as the RCCL path: both ranks return the WHOLE batch's scalars
    (counts exact), end with the whole-batch parameters of the reference-generated fixture, and are bit-identical replicas (every
    element of a sum is computed by exactly one rank, in rank order)."""
    case = C.CASES[name]
    r0, r1 = _run_world2(case, ipc=True)
    n0, n1 = int(r0.pop("ipc_messages")), int(r1.pop("ipc_messages"))
    assert n0 == n1 and n0 >= 4 * case["steps"], (n0, n1)       # at least count+sums, D gradient, G sums, G gradient per step
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    _check(name, r0, r1, {k: gold[k] for k in gold.files})
