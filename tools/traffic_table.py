"""Per-launch HBM/fabric traffic of the convolutions of one inference step next to their ALGORITHMIC bytes.

On the GPU box (tools/traffic_table.sh):   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/layer_roofline.py --steps 2 --json L.json
                                           (and once more with WRITE_SIZE: one counter set per pass, MI355X_MICROARCH.md HBM section)
Here:   python tools/traffic_table.py <dir with fetch/ write/ L.json> > profiles/r04/traffic_per_launch.txt
The i-th convolution dispatch of a step is the i-th record of amp_prof_launches (same stream, same order).  FETCH_SIZE / WRITE_SIZE are KiB,
FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads (the guide's correction, validated in round 1 on the stem's input)."""
import csv, glob, json, os, sys

src = sys.argv[1]
rows = json.load(open(os.path.join(src, "L.json")))["rows"]
per = len(rows)


def is_conv(name):
    return any(t in name for t in ("conv_split_kernel", "conv_glds_kernel", "conv_f16x3_kernel", "conv_mfma_kernel", "conv3x3_c64_kernel", "stem_pool"))


def per_dispatch(sub, counter):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    vals = {}
    names = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or not is_conv(r["Kernel_Name"]):
            continue
        d = int(r["Dispatch_Id"])
        vals[d] = vals.get(d, 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    ids = sorted(vals)
    return [vals[i] for i in ids], [names[i] for i in ids]


def short(n):
    import re
    m = re.search(r"(conv\w*_kernel|stem_pool\w+)<?([^>(]*)", n)
    if not m:
        return n[:30]
    a = [t.strip() for t in m.group(2).split(",")] if m.group(2) else []
    if m.group(1) == "conv_split_kernel":
        return f"split<{a[0]}x{a[1]}{',2buf' if len(a) > 3 and a[3] == '2' else ''}>"
    return (m.group(1).replace("conv_", "").replace("_kernel", "") + ("<" + a[0] + ">" if a else ""))


fe, names = per_dispatch("fetch", "FETCH_SIZE")
wr, _ = per_dispatch("write", "WRITE_SIZE")
assert len(fe) % per == 0 and len(fe) == len(wr), (len(fe), len(wr), per)
fe, wr, names = fe[-per:], wr[-per:], names[-per:]          # the last step
print(f"# one inference step (B=8, 1024x1024, f16x3), conv launches: counter bytes (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, per dispatch, 2 x FETCH) against algorithmic bytes")
print(f"{'#':>3} {'kernel':<22} {'M':>8} {'N':>5} {'K':>6} {'us':>8} {'alg MB':>8} {'read MB':>8} {'write MB':>8} {'ctr/alg':>7}")
agg = {}
for i, r in enumerate(rows):
    rd, w_ = fe[i] * 1024 * 2 / 1e6, wr[i] * 1024 / 1e6
    alg = r["bytes"] / 1e6
    k = short(names[i])
    print(f"{i:3d} {k:<22} {r['M']:8d} {r['N']:5d} {r['K']:6d} {r['ms'] * 1e3:8.1f} {alg:8.1f} {rd:8.1f} {w_:8.1f} {(rd + w_) / alg:7.2f}")
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += alg; a[2] += rd; a[3] += w_; a[4] += r["ms"]
print("# per kernel: launches, algorithmic MB, counter read MB, counter write MB, counter / algorithmic, ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][4]):
    print(f"# {k:<22} {a[0]:3d} {a[1]:9.1f} {a[2]:9.1f} {a[3]:9.1f} {(a[2] + a[3]) / a[1]:6.2f} {a[4]:7.3f}")
