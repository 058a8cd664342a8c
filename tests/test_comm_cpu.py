"""CPU tests of the multi-GPU exchange: the bucket plan of the gradient arena (host-only C-ABI
functions, no GPU), and the N>1 data path driven bucket by bucket over world_size-2 gloo on an arena-shaped tensor -- the
staged twin of what RCCL does on the device arena (ampis_amd/csrc/comm.hip)."""
import os

import numpy as np
import pytest
import torch

from ampis_amd import params as P
from ampis_amd.utils import comm

K = 2


def _arena():
    """Trainable + frozen tensors of R50-FPN laid out like the library's arena (state_dict order, 64-float alignment)."""
    shapes = {k: v for k, v in P.param_shapes(K).items() if ".norm." not in k}
    return comm.arena_layout(shapes)


def test_bucket_of_follows_the_backward_order():
    from ampis_amd import _lib
    L = _lib.lib()
    f = lambda n: L.amp_grad_bucket_of(n.encode())
    assert f("roi_heads.mask_head.mask_fcn1.weight") == 0 and f("roi_heads.mask_head.predictor.bias") == 0
    assert f("roi_heads.box_head.fc1.weight") == 1 and f("roi_heads.box_predictor.cls_score.weight") == 1 and f("roi_heads.box_predictor") == 1
    assert f("proposal_generator.rpn_head.conv.weight") == 2 and f("proposal_generator.rpn_head.pred") == 2
    assert f("backbone.fpn_lateral2.weight") == 3 and f("backbone.fpn_output5.bias") == 3
    assert f("backbone.bottom_up.res5.2.conv3.weight") == 4
    assert f("backbone.bottom_up.res4.0.shortcut.weight") == 5
    assert f("backbone.bottom_up.res3.1.conv1.weight") == 6
    assert f("backbone.bottom_up.res2.0.conv1.weight") == -1 and f("backbone.bottom_up.stem.conv1.weight") == -1   # FREEZE_AT = 2


def test_plan_covers_every_trainable_tensor_once():
    names, offs, sizes, total = _arena()
    plan = comm.plan_buckets(names, offs, sizes)
    assert [b for b, _, _ in plan] == sorted(b for b, _, _ in plan), "ranges come in issue order (bucket 0 first)"
    assert {b for b, _, _ in plan} == set(range(7))
    cover = np.zeros(total, dtype=np.int32)
    for _, o, n in plan:
        assert 0 <= o and o + n <= total
        cover[o:o + n] += 1
    assert cover.max() == 1, "ranges are disjoint"
    from ampis_amd import _lib
    L = _lib.lib()
    trainable = 0
    for nm, o, n in zip(names, offs, sizes):
        b = L.amp_grad_bucket_of(nm.encode())
        if b >= 0:
            assert cover[o:o + n].min() == 1, nm
            trainable += n
        else:
            assert cover[o:o + n].max() == 0, f"frozen tensor {nm} must not be exchanged"
    # merging only bridges the alignment padding
    assert trainable <= sum(n for _, _, n in plan) < trainable + 64 * len(names)
    assert len(plan) <= 16, "a handful of contiguous ranges, not one call per tensor"


def test_plan_rejects_a_gap_that_swallows_another_bucket():
    from ampis_amd import _lib
    with pytest.raises(_lib.AmpError):
        comm.plan_buckets(["roi_heads.mask_head.a", "backbone.fpn_lateral2.weight", "roi_heads.mask_head.b"], [0, 100, 200], [100, 100, 100], max_gap=1000)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    names, offs, sizes, total = _arena()
    plan = comm.plan_buckets(names, offs, sizes)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    mine = flat.clone()
    scale = comm.all_reduce_buckets_(flat, plan)
    comm.synchronize()
    # what the other rank had (same generator): the exchanged ranges must hold the sum, everything else must be untouched
    other = torch.randn(total, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    inside = torch.zeros(total, dtype=torch.bool)
    for _, o, n in plan:
        inside[o:o + n] = True
    ok_sum = bool(torch.equal(flat[inside], (mine + other)[inside]))
    ok_rest = bool(torch.equal(flat[~inside], mine[~inside]))
    q.put((rank, ok_sum, ok_rest, scale, int(inside.sum())))
    dist.destroy_process_group()


def test_bucketed_all_reduce_of_an_arena_shaped_tensor_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert all(r[1] and r[2] for r in res), res
    assert all(r[3] == 0.5 for r in res)
    assert res[0][4] == res[1][4] > 43_000_000       # R50-FPN, K=2: 43.7 M trainable floats


def _launch_main(path):
    from ampis_amd.utils import comm as c
    import torch.distributed as dist
    t = torch.tensor([float(c.get_rank() + 1)])
    dist.all_reduce(t)
    c.synchronize()
    with open(f"{path}.{c.get_rank()}", "w") as f:
        f.write(f"{c.get_world_size()} {t.item()} {os.environ['LOCAL_RANK']}")


def test_engine_launch_spawns_one_process_per_rank(tmp_path):
    """detectron2.engine.launch surface: workers are spawned with RANK / WORLD_SIZE and a gloo side channel."""
    from ampis_amd.engine import launch
    launch(_launch_main, 2, args=(str(tmp_path / "out"),))
    got = sorted(open(tmp_path / f"out.{r}").read() for r in range(2))
    assert got == ["2 3.0 0", "2 3.0 1"]


def test_bench_self_launches_when_rank_is_unset(monkeypatch):
    """`python bench.py --gpus 2` from a plain shell must start its own ranks (it asserted instead).  Here there is no
    GPU, so the ranks fail at their first assertion -- but they must have been started through torch.distributed.run."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["AMP_BENCH_DRY_LAUNCH"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    ranks = sorted(l for l in r.stdout.splitlines() if l.startswith("dry-launch rank"))
    assert ranks == ["dry-launch rank 0 of 2 local 0", "dry-launch rank 1 of 2 local 1"], (r.stdout, r.stderr[-2000:])
    assert r.returncode == 0
