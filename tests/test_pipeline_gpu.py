"""amp_pipeline (several batches in flight from one calling thread): results bit-identical to the plain call, in submission order;
the ticket protocol refuses misuse; a failing batch reports its own error and the lane keeps working."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sig(out):
    return [(o["boxes"].tobytes(), o["scores"].tobytes(), o["classes"].tobytes(), tuple(m["counts"] for m in o["masks"])) for o in out]


def test_pipeline_matches_plain_calls_in_order():
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import InferPipeline, MaskRCNN
    K, B, H, W, D = 2, 2, 256, 320, 40
    params = P.init_params(K, seed=3, style="spread")
    batches = [synth.batch(B, H, W, first_index=10 * i)[0] for i in range(7)]
    ctx = _lib.Context(0)
    m = MaskRCNN(ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
    m.load_params(params)
    want = [_sig(m.infer(b)) for b in batches]
    assert len({str(w) for w in want}) == len(want)                     # the batches really differ
    for depth in (1, 2, 3):
        pipe = InferPipeline(0, K, depth=depth, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
        pipe.load_params(params)
        got = [_sig(o) for o in pipe.map(batches)]
        assert got == want, depth
        # device-resident frames, explicit tickets, results collected late
        d = [ctx.malloc(b.nbytes) for b in batches[:depth]]
        for p, b in zip(d, batches):
            ctx.h2d(p, b)
        ctx.sync()
        tickets = [pipe.submit(device_ptr=p, shape=(B, H, W)) for p in d]
        with pytest.raises(_lib.AmpError, match="in flight"):
            pipe.submit(batches[0])                                      # depth + 1 uncollected batches
        for i, t in enumerate(tickets):
            assert _sig(pipe.wait(t)) == want[i]
        with pytest.raises(_lib.AmpError):
            pipe.wait(tickets[0])                                        # collected already
        # a batch beyond the capacity fails with ITS error at wait(); the lane takes the next batch
        big = np.concatenate([batches[0], batches[1]])
        t = pipe.submit(big)
        with pytest.raises(_lib.AmpError, match="capacity"):
            pipe.wait(t)
        assert _sig(pipe.wait(pipe.submit(batches[2]))) == want[2]
        for p in d:
            ctx.free(p)
        pipe.close()
    m.close(); ctx.close()
