"""True idle time of inference steps from a rocprofv3 kernel trace of bench.py (steps delimited by stem_pool_u8_kernel launches).
usage: python tools/infer_timeline.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "stem_pool_u8_kernel" in n]
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
out = []
for k in range(2, min(len(idx) - 1, 8)):
    a, b = idx[k], idx[k + 1]
    iv = sorted((S(r), E(r), n) for r, n in zip(rows[a:b], names[a:b]))
    cur, union, gaps = iv[0][0], 0, []
    prev = "(start)"
    for s, e, n in iv:
        if s > cur:
            gaps.append((s - cur, prev, n))
        if e > cur:
            union += e - max(s, cur); cur, prev = e, n
    wall = S(rows[b]) - S(rows[a])
    tail = wall - (cur - iv[0][0])          # from the last kernel's end to the next step's first kernel
    out.append((wall, union, tail, gaps))
for wall, union, tail, gaps in out:
    print(f"step: wall {wall / 1e6:.3f} ms, some kernel running {union / 1e6:.3f} ms, idle {(wall - union) / 1e6:.3f} ms of which between the steps {tail / 1e6:.3f} ms; "
          f"largest gaps inside: " + ", ".join(f"{g / 1e3:.0f} us after {p[-28:]}" for g, p, n in sorted(gaps, reverse=True)[:4]))
