"""Per-kernel time of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <kernel_stats.csv> <steps incl. warmup> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(int(r["TotalDurationNs"]) for r in rows)
conv = 0.0
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    ms = int(r["TotalDurationNs"]) / 1e6 / steps
    print(f"{n[:64]:64s} calls/step {int(r['Calls']) / steps:6.1f}  {ms:7.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
for r in rows:
    if "conv_" in r["Name"] and "kernel" in r["Name"]:
        conv += int(r["TotalDurationNs"]) / 1e6 / steps
print(f"all kernels {tot / 1e6 / steps:.3f} ms/step; conv kernels {conv:.3f}; others {tot / 1e6 / steps - conv:.3f}")
