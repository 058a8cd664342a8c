"""The N > 1 exchange path EXECUTED: two ranks, one card.  RCCL refuses two ranks on one device and the GPU box has one card, so
`AMP_RCCL_LIB` points ampis_amd/csrc/comm.hip at a shared-memory stand-in for librccl (tests/fake_rccl/fake_rccl.hip: the same
nine entry points, RCCL's enqueue-on-a-stream semantics kept with stream memory operations).  Everything ABOVE those nine symbols
is the product: dlopen + symbol binding, amp_comm_init(world = 2), the id hand-over (utils.comm.attach_rccl over gloo), amp_barrier,
amp_allreduce, amp_comm_broadcast / amp_model_broadcast_params (DDP's constructor broadcast, ampis/data_utils.py:135), the seven
event-ordered gradient buckets issued from inside the backward pass, the device-side wait of amp_model_sgd_step, the MAX-of-flag
collective that keeps the f16x3 re-run decision identical on every rank, and the failure protocol (a rank whose step fails
completes the step's collective sequence; every rank returns an error; the next step is in sequence again).
What stays hardware-only: RCCL's own transport over xGMI and its timing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_DIR = os.path.join(HERE, "fake_rccl")


def build_fake_rccl():
    so = os.path.join(FAKE_DIR, "libfake_rccl.so")
    src = os.path.join(FAKE_DIR, "fake_rccl.hip")
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, src, "-lrt", "-lpthread"],
                       check=True)
    return so


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_two_ranks(tmp_path, extra_env=None, timeout=420):
    so = build_fake_rccl()
    port = _free_port()
    procs, outs = [], []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AMP_RCCL_LIB=so, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        out = str(tmp_path / f"rank{rank}.json")
        log = open(tmp_path / f"rank{rank}.log", "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(FAKE_DIR, "rank_main.py"), out], env=env, stdout=log, stderr=subprocess.STDOUT), log))
        outs.append(out)
    try:
        for p, _ in procs:
            p.wait(timeout=timeout)
    finally:
        for p, log in procs:
            if p.poll() is None:
                p.kill()                       # the exact children started above
                p.wait()
            log.close()
    logs = [open(tmp_path / f"rank{r}.log").read()[-3000:] for r in range(2)]
    assert all(p.returncode == 0 for p, _ in procs), "\n".join(logs)
    return [json.load(open(o)) for o in outs], logs


def test_two_ranks_share_the_card_through_the_whole_exchange_path(tmp_path):
    reps, logs = run_two_ranks(tmp_path)
    r0, r1 = reps
    for r, rep in enumerate(reps):
        assert rep["info"][:2] == [r, 2] and rep["info"][2] == 99999, rep["info"]          # the stand-in's version code: AMP_RCCL_LIB was honoured
        assert rep["allreduce_sum_exact"] and rep["allreduce_max"] == [20, 0] and rep["broadcast_from_1"] == 8, rep
        assert rep["params_are_rank0s"] and rep["momentum_is_rank0s"], "amp_model_broadcast_params"
        assert rep["ranks_have_different_grads"] and not rep["exchanged_before"]
        assert rep["explicit_scale"] == 0.5 and rep["exchanged_after_explicit"] and rep["explicit_sum_exact"], "overlap off: utils.comm must exchange"
        assert rep["second_exchange_refused"]
        assert rep["overlapped_sum_exact"], "buckets issued from inside the backward pass: SUM over both ranks, bit for bit, three rounds"
        assert rep["overlap_scale"] == 0.5 and rep["no_double_sum"]
        assert len(rep["bucket_us"]) == 7 and all(u > 0 for u in rep["bucket_us"]), rep["bucket_us"]
        assert rep["stats"]["span_ms"] > 0
        assert rep["failed_step_error"] is not None and rep["sgd_after_failed_step_refused"], rep.get("failed_step_error")
    # the failing rank reports its own error, the other one that a peer failed
    assert "capacity" in r1["failed_step_error"] and "another rank" in r0["failed_step_error"], (r0["failed_step_error"], r1["failed_step_error"])
    # replicas stay identical: after the first update, and after the failed step + recovery
    assert r0["params_after_step"] == r1["params_after_step"] and r0["momentum_after_step"] == r1["momentum_after_step"]
    assert r0["params_after_recovery"] == r1["params_after_recovery"]
    assert r0["params_after_recovery"] != r0["params_after_step"]
    if "range_step" in r0:
        # only rank 1 overflowed the f16 range; BOTH re-ran in fp32 (otherwise the collective sequences differ and the stand-in reports it)
        assert r0["range_step"] == r1["range_step"] == "ok", (r0["range_step"], r1["range_step"])
        assert r0["range_grads_hash"] == r1["range_grads_hash"]
        assert "re-running the step in AMP_CONV_F32" in logs[0] and "re-running the step in AMP_CONV_F32" in logs[1]
    assert "[fake_rccl]" not in logs[0] + logs[1], "the stand-in reported a sequence mismatch or a timeout"


def test_blocking_variant_of_the_stand_in_agrees(tmp_path):
    """Same run with the stand-in's host-blocking mode (no stream memory operations): a cross-check of the stand-in itself."""
    reps, _ = run_two_ranks(tmp_path, {"FAKE_RCCL_BLOCKING": "1"})
    assert all(r["overlapped_sum_exact"] and r["explicit_sum_exact"] and r["params_are_rank0s"] for r in reps)
    assert reps[0]["params_after_recovery"] == reps[1]["params_after_recovery"]
