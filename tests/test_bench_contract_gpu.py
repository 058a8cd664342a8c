"""The driver's contract with bench.py, held on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` prints exactly ONE JSON line on stdout
with the fields the driver and the judge read -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload, a `roofline` object for the dominant kernel (live HIP-event numbers) and a `cpu_baseline` object
(the oracle timed on a bounded sample) -- and the numbers are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_json_line_with_roofline_and_cpu_baseline():
    K, W = 4, 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(K), "--warmup", str(W), "--train-steps", "0", "--x101-steps", "0",
                        "--no-two-pipelines", "--no-host-inclusive", "--no-strict", "--cpu-images", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"{len(lines)} lines on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == K and d["warmup"] == W and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "synthetic" in d["data"] and d["unit"] == "images/s" and "workload" in d["config"] and "model" not in d["config"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["global_batch"] == 8
    assert d["value"] == pytest.approx(8 * 1e3 / d["ms_per_step"], rel=1e-3) and 100 < d["value"] < 2000
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == pytest.approx(833.3, abs=0.1) and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=2e-3)
    assert 0.2 < rf["frac"] < 1.0 and "conv_split_kernel<128x256>" in rf["kernel"]
    assert rf["launches_per_step"] == 23 and rf["kernel_ms_per_step"] < d["ms_per_step"]          # the dominant kernel alone, tagged at its launch site (the mask tail is mask_tail_kernel)
    assert rf["traffic"] is None or rf["traffic"] > 1e8                                            # HBM bytes per launch from the committed PMC summary
    pp = rf["power_probe"]                                                                        # the dominant layer on random and on all-zero operands
    assert "error" not in pp and pp["random_operands_tflops"] > 200 and pp["zero_operands_tflops"] > 200 and 0.9 < pp["zero_over_random"] < 2.0      # (a diagnostic: ~1.35 on a power-limited box)
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["unit"] == "images/s" and cb["cores"] >= 1 and 0 < cb["value"] < d["value"]


def test_two_ranks_without_a_communicator_still_print_the_inference_line():
    """`bench.py --gpus 2` when RCCL cannot create the communicator (here: both ranks on the one card of the box, which RCCL refuses): the ranks agree
    over the side channel, the timing collectives fall back to gloo, the line is printed with "rccl_error" and n_gpus = 2, the training leg reports
    an error object instead of a number, exit code 0 -- the inference path has no data-path collective, its number stands."""
    env = dict(os.environ, AMP_BENCH_ONE_CARD="1", MASTER_ADDR="127.0.0.1")
    env.pop("AMP_RCCL_LIB", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--train-steps", "1", "--train-warmup", "0", "--x101-steps", "0",
                        "--no-two-pipelines", "--no-host-inclusive", "--no-strict", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["value"] > 50
    assert "rccl_error" in d and "NOT CREATED" in d["config"]["parallelism"]
    assert "error" in d["train"] and "RCCL" in d["train"]["error"]
    assert "RCCL communicator NOT created" in r.stderr
