"""How precisely does the reference's own fp32 arithmetic define an output?  (oracle/exact.py)

The fp32 oracle (torch CPU fp32 = the arithmetic of detectron2's CPU path) is run against an exact-convolution evaluation of the
same network (every conv / linear in fp64, rounded once) on the images the end-to-end tests use, and the distance is measured WITH
THE GATE ITSELF (oracle/gate.py in its measuring mode): boxes, scores, and -- what round 3 lacked -- the MASKS: how many masks have
differing pixels, how many of those pixels lie beyond the fixed probability-noise margin, how many masks fall below IoU 0.999 and the
lowest IoU.  The numbers go to tests/golden/oracle_noise_floor.json; the end-to-end tests and smoke() bind their caps to them
(HIP-vs-oracle <= 1.5 x oracle-vs-exact-oracle, gate.assert_floor).

python tools/oracle_noise_floor.py [--write]      -- CPU only; a few minutes per 1024x1024 image (the fp64 convolutions)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ampis_amd import params as P, synth   # noqa: E402
from oracle import exact, gate, maskrcnn as O    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "oracle_noise_floor.json")

# name -> (images, params seed, K, D): the SAME inputs the tests that use the entry build
CASES = {
    # tests/test_fullsize_gpu.py::test_fullsize_batch_against_the_oracle: images 0 and 5 of synth.batch(8, 1024, 1024, first_index=7)
    # (indices 7 and 12); the other six images of that batch make the floor a RATE over 1600 instances instead of a count over 400
    "fullsize_1024": dict(hw=(1024, 1024), first_index=[7, 12, 8, 9, 10, 11, 13, 14], seed=None, pseed=0, K=2, D=200),
    # __graft_entry__.smoke()
    "smoke_192x256": dict(hw=(192, 256), first_index=[None], seed=77, pseed=11, K=2, D=40),
}


def as_hip(r):
    """an oracle result in the shape gate.check_image expects of the product path"""
    return dict(boxes=r["boxes"].numpy(), scores=r["scores"].numpy(), classes=r["classes"].numpy(), masks=list(r["masks"].numpy()))


def measure(case):
    c = CASES[case]
    p = O.to_torch_params(P.init_params(c["K"], seed=c["pseed"], style="spread"))
    cfg = O.Cfg(num_classes=c["K"], detections_per_image=c["D"])
    h, w = c["hw"]
    per_image = []
    for fi in c["first_index"]:
        img, _ = synth.batch(1, h, w, seed=c["seed"]) if fi is None else synth.batch(1, h, w, first_index=fi)
        t0 = time.time()
        r32 = O.infer(img, p, cfg)[0]
        t1 = time.time()
        with exact.exact_convs():
            rex = O.infer(img, p, cfg)[0]
        t2 = time.time()
        # the exact evaluation is the reference point, the fp32 oracle plays the part the HIP path plays in the tests
        st = gate.check_image(as_hip(r32), rex, h, w, lambda m: m, strict=False)
        viol = st.pop("violation_list")
        print(f"[{case}] image {fi}: fp32 oracle {t1 - t0:.0f} s, exact {t2 - t1:.0f} s | {gate.summary(st)} | violations of the gate rule: "
              f"{st['violations']} {viol[:3]}", flush=True)
        per_image.append(st)
    tot = gate.merge(per_image)
    return dict(per_image=per_image, total=tot, inputs={k: v for k, v in c.items()})


if __name__ == "__main__":
    torch.set_num_threads(min(16, torch.get_num_threads()))
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(CASES)
    res = {}
    if os.path.exists(OUT):
        res = json.load(open(OUT))
    for n in names:
        res[n] = measure(n)
    res["_meta"] = dict(made_by="tools/oracle_noise_floor.py", torch=torch.__version__, threads=torch.get_num_threads(),
                        what="gate statistics of the fp32 torch-CPU oracle against oracle/exact.py (fp64 convolutions rounded once) -- the "
                             "reference arithmetic's own noise floor; tests bind HIP-vs-oracle to 1.5 x these (oracle/gate.py assert_floor)",
                        prob_noise=gate.PROB_NOISE, box_tol=gate.BOX_TOL, box_rel=gate.BOX_REL)
    if "--write" in sys.argv:
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        print("wrote", OUT)
