"""RCCL behind the C ABI on real hardware (amp_comm_*, ampis_amd/csrc/comm.hip).  The GPU box has ONE card, and RCCL refuses
two ranks on one device, so the communicator here has size 1: what is checked is that librccl loads, initialises and runs the
collectives on the library's communication stream, that the bucketed exchange issued from inside the backward pass is ordered
correctly against the compute stream (gradients and the SGD step equal those of a run without a communicator), and the plan of
the real arena.  The N = 2 data path is covered on CPU (tests/test_comm_cpu.py); the 8-GPU run is the driver's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_ctx():
    from ampis_amd import _lib
    ctx = _lib.Context(0)
    ctx.comm_init(0, 1, _lib.Context.comm_unique_id())
    yield ctx
    ctx.close()


def test_communicator_of_size_one_comes_up(rccl_ctx):
    rank, world, version = rccl_ctx.comm_info()
    assert (rank, world) == (0, 1) and version >= 20000, version      # NCCL-style version code, RCCL 2.x
    rccl_ctx.barrier()
    rccl_ctx.barrier()
    from ampis_amd import _lib
    with pytest.raises(_lib.AmpError):
        rccl_ctx.comm_init(0, 1, _lib.Context.comm_unique_id())      # one communicator per context


def test_allreduce_runs_on_the_comm_stream_and_orders_against_the_compute_stream(rccl_ctx):
    ctx = rccl_ctx
    n = 1 << 20
    a = np.random.default_rng(0).standard_normal(n).astype(np.float32)
    d = ctx.malloc(a.nbytes)
    ctx.h2d(d, a)
    ctx.allreduce(d, n, ctx.F32, ctx.SUM)
    out = np.empty_like(a)
    ctx.d2h(out, d)
    assert np.array_equal(out, a)                                     # one rank: the sum is the value itself
    t = np.array([3.25], dtype=np.float64)
    ctx.h2d(d, t)
    ctx.allreduce(d, 1, ctx.F64, ctx.MAX)
    ctx.d2h(t, d)
    assert t[0] == 3.25
    ctx.free(d)


def test_context_without_communicator_reports_it():
    from ampis_amd import _lib
    ctx = _lib.Context(0)
    assert ctx.comm_info() == (0, 0, 0)
    with pytest.raises(_lib.AmpError):
        ctx.barrier()
    ctx.close()


def _train_setup(ctx, seed=3):
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    K, B, H, W = 2, 2, 192, 256
    imgs, gts = synth.batch(B, H, W, seed=9)
    gts = [dict(boxes=g["boxes"][:40], classes=g["classes"][:40], polygons=g["polygons"][:40]) for g in gts]
    npp = P.init_params(K, seed=2, style="spread")
    model = MaskRCNN(ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    model.load_params(npp)
    return model, imgs, gts


# tensors whose gradients do not pass through the RoIAlign backward (float atomics): bitwise reproducible run to run
HEAD = ("roi_heads.mask_head.mask_fcn2.weight", "roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.predictor.bias",
        "roi_heads.box_head.fc1.weight", "roi_heads.box_head.fc2.bias", "roi_heads.box_predictor.bbox_pred.weight")
TRUNK = ("proposal_generator.rpn_head.conv.weight", "backbone.fpn_output3.weight", "backbone.fpn_lateral5.bias",
         "backbone.bottom_up.res5.2.conv3.weight", "backbone.bottom_up.res4.0.shortcut.weight", "backbone.bottom_up.res3.0.conv1.weight")


def test_overlapped_exchange_leaves_gradients_and_sgd_step_intact(rccl_ctx):
    """Same batch, same seed: (a) a context without a communicator, (b) the RCCL context, buckets issued from inside the backward
    pass.  With one rank the reduced gradient is the gradient, so any ordering mistake between the two streams (a bucket sent
    before its last wgrad, an SGD kernel that does not wait) shows up as a difference."""
    from ampis_amd import _lib
    plain = _lib.Context(0)
    m0, imgs, gts = _train_setup(plain)
    m1, _, _ = _train_setup(rccl_ctx)
    l0 = m0.forward_losses(imgs, gts, seed=3, backward=True)
    for rep in range(3):                                               # several rounds through the same events
        l1 = m1.forward_losses(imgs, gts, seed=3, backward=True)
        st = rccl_ctx.comm_stats()
        assert st["span_ms"] > 0.0 and st["exposed_ms"] >= 0.0 and np.isfinite(st["span_ms"]), st
    assert l0 == l1
    for name in HEAD:
        assert np.array_equal(m0.get_tensor(name, grad=True), m1.get_tensor(name, grad=True)), name
    for name in TRUNK:
        a, b = m0.get_tensor(name, grad=True), m1.get_tensor(name, grad=True)
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(a).max(), 1e-12), name     # atomics in RoIAlign backward: fp32 rounding only
    from ampis_amd.utils import comm
    assert comm.all_reduce_gradients(m0, plain) == 1.0 and comm.all_reduce_gradients(m1, rccl_ctx) == 1.0
    m0.sgd_step(0.01, 0.9, 1e-4)
    m1.sgd_step(0.01, 0.9, 1e-4)
    for name in HEAD:
        assert np.array_equal(m0.get_tensor(name), m1.get_tensor(name)), name
    for name in TRUNK:
        a, b = m0.get_tensor(name), m1.get_tensor(name)
        assert np.abs(a - b).max() <= 1e-6 * max(np.abs(a).max(), 1e-12), name
    m0.close(); m1.close(); plain.close()


def test_explicit_exchange_with_the_overlap_switched_off(rccl_ctx):
    m, imgs, gts = _train_setup(rccl_ctx)
    m.set_grad_overlap(False)
    m.forward_losses(imgs, gts, seed=3, backward=True)
    before = {n: m.get_tensor(n, grad=True) for n in HEAD}
    m.allreduce_grads()                                                # all buckets at once, after the backward pass
    for n in HEAD:
        assert np.array_equal(before[n], m.get_tensor(n, grad=True)), n
    m.sgd_step(0.01, 0.9, 1e-4)
    m.close()


def test_plan_of_the_real_arena(rccl_ctx):
    from ampis_amd import params as P
    m, _, _ = _train_setup(rccl_ctx)
    plan = m.grad_buckets()
    _, nfloats = m.grad_arena()
    assert [b for b, _, _ in plan] == sorted(b for b, _, _ in plan) and {b for b, _, _ in plan} == set(range(7))
    spans = sorted((o, o + n) for _, o, n in plan)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "ranges are disjoint"
    assert spans[-1][1] <= nfloats
    shapes = P.param_shapes(2)
    trainable = sum(int(np.prod(s)) for k, s in shapes.items() if ".norm." not in k and not k.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2")))
    total = sum(n for _, _, n in plan)
    # the arena pads the mask predictor and the fused predictors to multiples of 4 rows, and every tensor to 64 floats
    assert trainable <= total <= trainable + 64 * len(shapes) + 8 * 1024, (trainable, total)
    m.close()
