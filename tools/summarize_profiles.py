"""gpurun_out/<tag>/ (tools/profile_round.sh) -> profiles/<round>/: kernel statistics CSV, the bench line printed under rocprofv3,
and pmc_summary.json (per kernel: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of
MI355X_MICROARCH.md's HBM section, MFMA utilisation from SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE, achieved HBM GB/s)."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag, dst, suffix = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
src = os.path.join("gpurun_out", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"(conv_\w+_kernel)<([^>]*)>", name)
    if m:
        args = [t.strip() for t in m.group(2).split(",")]
        if m.group(1) == "conv_glds_kernel":      # <BN, STEM, EPI, F16>
            tag = ("[stem]" if len(args) > 1 and args[1] == "true" else "") + ("[f16x3, both operands pre-split]" if len(args) > 3 and args[3] == "true" else "")
        elif m.group(1) == "conv_split_kernel":   # <BM, BN, EPI>: both operands pre-split, 3-buffer ring
            return f"conv_split_kernel<{args[0]}x{args[1]}>"
        elif m.group(1) == "conv_f16x3_kernel":   # <BN, EPI, STEM, SCALED>
            tag = ("[stem]" if len(args) > 2 and args[2] == "true" else "") + ("[scaled input]" if len(args) > 3 and args[3] == "true" else "")
        else:
            tag = ""
        return f"{m.group(1)}<{args[0]}>{tag}"
    m = re.search(r"(wgrad_\w+_kernel)<([^>]*)>", name)
    if m:
        return m.group(1)
    m = re.search(r"::(\w+_kernel)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


prefix = sys.argv[4] if len(sys.argv) > 4 else "bench"          # "train" for the training-step profile
stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{prefix}_kernel_stats{suffix}.csv"))
shutil.copy(os.path.join(src, "bench_line_under_rocprof.json"), os.path.join(dst, f"{prefix}_line_under_rocprof{suffix}.json"))
# kernel time per short name from the trace of the stats run
dur = defaultdict(float); cnt = defaultdict(int)
for r in csv.DictReader(open(glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv"))[0])):
    k = short(r["Kernel_Name"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; cnt[k] += 1

ctr = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for d in glob.glob(os.path.join(src, "pmc_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); c = r["Counter_Name"]
            ctr[k][c] += float(r["Counter_Value"]); n[k][c] += 1
out = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "
                  "--train-steps 0 --x101-steps 0 --no-strict  (one set per pass: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                  "SQ_BUSY_CYCLES | GRBM_GUI_ACTIVE)",
       "corrections": "FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-B/lane streaming reads, "
                      "MI355X_MICROARCH.md HBM section; validated in round 1 on the stem: 2*FETCH = 128.8 MiB vs the 128 MiB input). "
                      "GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).",
       "kernels": {}}
for k in sorted(ctr, key=lambda k: -dur.get(k, 0)):
    if dur.get(k, 0) < 2e-4:
        continue
    e = {"launches_in_stats_run": cnt[k], "avg_duration_us": round(dur[k] / max(cnt[k], 1) * 1e6, 1)}
    if "FETCH_SIZE" in ctr[k]:
        rd = ctr[k]["FETCH_SIZE"] * 1024 * 2 / n[k]["FETCH_SIZE"]; wr = ctr[k]["WRITE_SIZE"] * 1024 / max(n[k]["WRITE_SIZE"], 1)
        e.update(hbm_read_bytes_per_launch=int(rd), hbm_write_bytes_per_launch=int(wr), hbm_bytes_per_launch=int(rd + wr))
        if cnt[k]:
            e["hbm_GBps"] = round((rd + wr) / (dur[k] / cnt[k]) / 1e9, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ctr[k] and "GRBM_GUI_ACTIVE" in ctr[k]:
        busy = ctr[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
        act = ctr[k]["GRBM_GUI_ACTIVE"] / n[k]["GRBM_GUI_ACTIVE"]
        e["mfma_util"] = round(busy / (act / 8 * 1024), 4)
    out["kernels"][k] = e
out["command"] = out["command"] if prefix == "bench" else out["command"].replace("--train-steps 0", "--steps 2 --warmup 1 --train-steps 6 --train-warmup 2")
json.dump(out, open(os.path.join(dst, f"pmc_summary{'_train' if prefix != 'bench' else ''}{suffix}.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e)
