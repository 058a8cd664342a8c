"""One rank of tests/test_comm_world2_gpu.py: the library's RCCL path (ampis_amd/csrc/comm.hip + the collective protocol of
amp_model_forward_backward) with a communicator of TWO ranks sharing cuda:0, through the shared-memory stand-in for librccl that
AMP_RCCL_LIB selects (tests/fake_rccl/fake_rccl.hip).  Writes a JSON report; the parent test asserts on it."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main(out_path):
    import torch.distributed as dist
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    from ampis_amd.utils import comm

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rep = {"rank": rank}
    ctx = _lib.Context(0)
    info = comm.attach_rccl(ctx)
    rep["info"] = list(info)
    ctx.barrier(); ctx.barrier()

    # ---- plain collectives: SUM of floats, MAX of ints, broadcast of bytes ----
    n = (3 << 20) + 17                                               # more than one 8-MiB round of the stand-in, odd tail
    a = np.random.default_rng(100 + rank).standard_normal(n).astype(np.float32)
    both = [np.random.default_rng(100 + r).standard_normal(n).astype(np.float32) for r in range(world)]
    d = ctx.malloc(a.nbytes)
    ctx.h2d(d, a)
    ctx.allreduce(d, n, ctx.F32, ctx.SUM)
    got = np.empty_like(a); ctx.sync(); ctx.d2h(got, d)
    want = both[0].copy()
    for r in range(1, world):
        want += both[r]
    rep["allreduce_sum_exact"] = bool(np.array_equal(got, want))
    iv = np.array([10 * (rank + 1), -5 * rank], dtype=np.int32)
    ctx.h2d(d, iv); ctx.allreduce(d, 2, ctx.I32, ctx.MAX); ctx.sync(); ctx.d2h(iv, d)
    rep["allreduce_max"] = iv.tolist()
    by = np.full(1000, 7 + rank, dtype=np.uint8)
    ctx.h2d(d, by); ctx.broadcast(d, by.nbytes, root=1); ctx.sync(); ctx.d2h(by, d)
    rep["broadcast_from_1"] = int(by[0]) if np.all(by == by[0]) else -1
    ctx.free(d)

    # ---- a model per rank, DIFFERENT initial weights and momentum; rank 0's must win ----
    K, B, H, W = 2, 2, 192, 256
    imgs, gts = synth.batch(B, H, W, seed=9 + rank)                  # different data per rank
    gts = [dict(boxes=g["boxes"][:40], classes=g["classes"][:40], polygons=g["polygons"][:40]) for g in gts]
    model = MaskRCNN(ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    model.load_params(P.init_params(K, seed=2 + rank, style="spread"))
    mom = model.momentum()
    model.momentum(np.full_like(mom, 0.001 * (rank + 1)))
    names = ("roi_heads.mask_head.mask_fcn2.weight", "roi_heads.box_head.fc2.bias", "backbone.fpn_output3.weight",
             "backbone.bottom_up.res4.0.shortcut.weight", "backbone.bottom_up.stem.conv1.weight")
    model.broadcast_params(0)
    ref0 = P.init_params(K, seed=2, style="spread")
    rep["params_are_rank0s"] = all(np.array_equal(model.get_tensor(k), ref0[k]) for k in names)
    rep["momentum_is_rank0s"] = bool(np.all(model.momentum() == np.float32(0.001)))

    def arena():
        p, nf = model.grad_arena()
        g = np.empty(nf, dtype=np.float32)
        ctx.comm_wait(); ctx.sync(); ctx.d2h(g, p)       # the arena is read after the exchange, not beside it
        return g

    def gathered(x):
        import torch
        xs = [torch.empty(x.shape, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(xs, torch.from_numpy(x))
        return [t.numpy() for t in xs]

    # ---- local gradients (no exchange), then the expected SUM in rank order ----
    model.set_grad_overlap(False)
    l_local = model.forward_losses(imgs, gts, seed=3, backward=True)
    rep["exchanged_before"] = model.grads_exchanged()
    g_local = arena()
    parts = gathered(g_local)
    want = parts[0].copy()
    for r in range(1, world):
        want += parts[r]
    rep["ranks_have_different_grads"] = not np.array_equal(parts[0], parts[1])
    # the explicit exchange through utils.comm (what DefaultTrainer calls): overlap off -> it must hand the buckets over itself
    scale = comm.all_reduce_gradients(model, ctx)
    rep["explicit_scale"] = scale
    rep["exchanged_after_explicit"] = model.grads_exchanged()
    g = arena()
    plan = model.grad_buckets()
    rep["explicit_sum_exact"] = all(np.array_equal(g[o:o + n_], want[o:o + n_]) for _, o, n_ in plan)
    try:
        model.allreduce_grads()
        rep["second_exchange_refused"] = False
    except _lib.AmpError:
        rep["second_exchange_refused"] = True

    # ---- the overlapped exchange from inside the backward pass, several rounds through the same events ----
    model.set_grad_overlap(True)
    ok = True
    for _ in range(3):
        l_ov = model.forward_losses(imgs, gts, seed=3, backward=True)
        ok = ok and l_ov == l_local and model.grads_exchanged()
        g = arena()
        ok = ok and all(np.array_equal(g[o:o + n_], want[o:o + n_]) for _, o, n_ in plan)
    rep["overlapped_sum_exact"] = bool(ok)
    rep["bucket_us"] = ctx.comm_bucket_stats()
    rep["stats"] = ctx.comm_stats()
    rep["overlap_scale"] = comm.all_reduce_gradients(model, ctx)     # nothing left to do: must not exchange a second time
    g2 = arena()
    rep["no_double_sum"] = bool(np.array_equal(g, g2))
    model.sgd_step(0.01, 0.9, 1e-4, grad_scale=rep["overlap_scale"])
    rep["params_after_step"] = {k: sha(model.get_tensor(k)) for k in names}
    rep["momentum_after_step"] = sha(model.momentum())

    # ---- a step that FAILS on rank 1 only (batch beyond the model's capacity): both ranks must return an error, nobody hangs ----
    try:
        if rank == 1:
            big = np.concatenate([imgs, imgs[:1]])
            model.forward_losses(big, gts + gts[:1], seed=3, backward=True)
        else:
            model.forward_losses(imgs, gts, seed=3, backward=True)
        rep["failed_step_error"] = None
    except _lib.AmpError as e:
        rep["failed_step_error"] = str(e)[:200]
    try:
        model.sgd_step(0.01, 0.9, 1e-4, grad_scale=0.5)
        rep["sgd_after_failed_step_refused"] = False
    except _lib.AmpError:
        rep["sgd_after_failed_step_refused"] = True
    # ... and the NEXT step is in sequence again on both ranks
    model.forward_losses(imgs, gts, seed=4, backward=True)
    model.sgd_step(0.01, 0.9, 1e-4, grad_scale=0.5)
    rep["params_after_recovery"] = {k: sha(model.get_tensor(k)) for k in names}

    # ---- the f16x3 range flag raised on rank 1 only: both ranks must re-run in fp32 (same collective sequence), same update ----
    if ctx.conv_mode == ctx.CONV_F16X3:
        if rank == 1:      # activations beyond 65504 behind this layer, on this rank only; weights stay below the split's own range check
            w = model.get_tensor("backbone.fpn_output2.weight")
            lib = _lib.lib()
            import ctypes as C
            big_w = np.ascontiguousarray(w * np.float32(3e4 / max(np.abs(w).max(), 1e-9)))
            shape = (C.c_longlong * big_w.ndim)(*big_w.shape)
            _lib.check(lib.amp_model_load_tensor(model._h, b"backbone.fpn_output2.weight", big_w.ctypes.data_as(C.c_void_p), shape, big_w.ndim), "load")
            _lib.check(lib.amp_model_finalize(model._h), "finalize")
        try:
            model.forward_losses(imgs, gts, seed=5, backward=True)
            rep["range_step"] = "ok"
        except _lib.AmpError as e:
            rep["range_step"] = "error: " + str(e)[:160]
        g = arena()
        rep["range_grads_hash"] = sha(np.nan_to_num(g))
    ctx.barrier()
    model.close()
    comm.detach_rccl()
    ctx.close()
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(rep, f)


if __name__ == "__main__":
    main(sys.argv[1])
