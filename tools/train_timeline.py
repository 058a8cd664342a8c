"""True idle time of one training step from a rocprofv3 kernel trace (two streams overlap: consecutive-start gaps mean nothing).
usage: python tools/train_timeline.py <kernel_trace.csv> [step index]     (steps are delimited by anchor_match_kernel launches)"""
import collections, csv, re, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "anchor_match_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
a, b = idx[k], idx[k + 1]
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
t0, t1 = S(rows[a]), S(rows[b])
# union of the busy intervals
iv = sorted((S(r), E(r), n) for r, n in zip(rows[a:b], names[a:b]))
gaps, cur_end, cur_name, union = [], iv[0][0], "(step start)", 0
for s, e, n in iv:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
        union += 0
        seg_start = s
    if e > cur_end:
        union += e - max(s, cur_end)
        cur_end, cur_name = e, n
wall = t1 - t0
print(f"one training step: {b - a} launches, wall {wall / 1e6:.3f} ms, some kernel running {union / 1e6:.3f} ms, NO kernel running {(wall - union) / 1e6:.3f} ms, "
      f"sum of kernel durations {sum(e - s for s, e, _ in iv) / 1e6:.3f} ms (overlap {(sum(e - s for s, e, _ in iv) - union) / 1e6:.3f} ms)")
qs = collections.Counter(r.get("Queue_Id", "?") for r in rows[a:b])
print("launches per queue:", dict(qs))
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in iv:
    key = re.sub(r"<.*", "", n)[-44:]
    agg[key][0] += 1; agg[key][1] += (e - s) / 1e6
for key, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"   {key:46s} {c:5d} {ms:8.3f} ms")
print("largest intervals with NO kernel running:")
for g, p, n in sorted(gaps, reverse=True)[:14]:
    print(f"   {g / 1e3:8.1f} us  after {p[-44:]:44s} before {n[-44:]}")
print(f"   ({len(gaps)} such intervals, {sum(g for g, _, _ in gaps) / 1e6:.3f} ms; {sum(1 for g, _, _ in gaps if g > 20000)} above 20 us)")
