"""-m gpu: the engine's OWN data-parallel communicator (gt_comm_*, engine.hip) with TWO ranks.

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
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.isfile(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.isfile("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", FAKE, "-lrt"])
    return FAKE


def _worker(rank, world, case, idq, outq):
    os.environ["GT_RCCL_LIB"] = FAKE
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
        got = run_hip_case(case, shard=(rank, world), comm_id=cid, extra=extra)
        torch.cuda.synchronize()
        got["philox"] = extra["philox"]
        outq.put((rank, None, got))
    except Exception as e:      # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        outq.put((rank, "%s\n%s" % (e, traceback.format_exc()), None))


def _run_world2(case):
    import torch.multiprocessing as mp
    build_fake_rccl()
    ctx = mp.get_context("spawn")
    idq, outq = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, case, idq, outq)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in procs:
            rank, err, got = outq.get(timeout=420)
            assert err is None, "rank %d failed: %s" % (rank, err)
            results[rank] = got
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return results[0], results[1]


def _check(name, r0, r1, ref):
    from test_gpu_parity import _close
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
                _close(got[k], ref[k], rtol=5e-4, atol=1e-9, msg=tag)
            else:
                _close(got[k], ref[k], msg=tag)
    for k in r0:
        if k.startswith(("G.", "D.")):
            assert np.array_equal(r0[k], r1[k]), "replicas differ in %s" % k           # bit-identical steps on every rank
    # the ranks' Philox streams differ (the rank is part of the dropout site): a world-W batch has W x the distinct masks
    assert not np.array_equal(r0["philox"], r1["philox"])
    assert abs(float(r0["philox"].mean()) - 0.5) < 0.06 and abs(float(r1["philox"].mean()) - 0.5) < 0.06


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["acoustic_mlp", "acoustic_mlp_dropout", "acoustic_lstm", "vc_in2out", "duration_mlp"])
def test_engine_communicator_world_2_equals_whole_batch_reference_golden(name):
    """MLP (one merged upper-layer message + the first layer's), MLP with injected dropout (masks sharded by sequence),
    BiLSTM (a flush per layer: more collectives per step than the 8-event ring holds), In2Out (gate + MLPG inside the
    model), duration (R = None, Adam): sharded over two ranks == the reference's whole-batch fixture."""
    case = C.CASES[name]
    r0, r1 = _run_world2(case)
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    _check(name, r0, r1, {k: gold[k] for k in gold.files})


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["acoustic_lstm_dropout", "acoustic_sru_dropout"])
def test_engine_communicator_world_2_equals_whole_batch_oracle(name):
    """3-layer BiLSTM with inter-layer dropout and the SRU generator with both variational dropouts (oracle-only cases,
    masks injected and sharded): per-layer flushes under the recurrences."""
    from oracle_runner import run_oracle_case
    case = C.ORACLE_ONLY_CASES[name]
    r0, r1 = _run_world2(case)
    _check(name, r0, r1, run_oracle_case(case))
