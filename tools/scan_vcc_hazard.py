"""Scan a gfx950 ISA listing (hipcc -S) for the pattern that produced wrong results under multi-kernel concurrency:
a VALU instruction writes VCC / an SGPR pair (v_cmp*, v_div_scale, v_add_co ...), a VALU instruction reads it as a mask (v_cndmask, v_div_fmas,
v_addc ...) and the wait states between the two that the compiler counted consist partly of packed-FP32 instructions (v_pk_*).
usage: python tools/scan_vcc_hazard.py file.s [...]"""
import re
import sys


def scan(path):
    kernel = "?"
    ins = []
    for raw in open(path):
        l = raw.strip()
        m = re.match(r"(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        ins.append((kernel, l))
    out = []
    for i, (k, l) in enumerate(ins):
        m = re.match(r"(v_cmpx?_\w+|v_div_scale_f32|v_add_co_u32\w*|v_sub_co_u32\w*)\s+(.*)", l)
        if not m:
            continue
        ops = m.group(2)
        dst = None
        first = ops.split(",")[0].strip()
        if first == "vcc" or re.match(r"s\[\d+:\d+\]$", first):
            dst = first
        if m.group(1) == "v_div_scale_f32":
            parts = [p.strip() for p in ops.split(",")]
            dst = parts[1] if len(parts) > 1 and (parts[1] == "vcc" or parts[1].startswith("s[")) else None
        if not dst:
            continue
        real, pk = 0, 0
        for j in range(i + 1, min(i + 9, len(ins))):
            k2, l2 = ins[j]
            if k2 != k:
                break
            op = l2.split()[0]
            rest = l2.split(None, 1)[1] if len(l2.split(None, 1)) > 1 else ""
            if op.startswith("s_nop"):
                real += int(rest) + 1
                continue
            parts = [x.strip() for x in rest.split(",")]
            dst_slots = [0] if re.match(r"v_cmpx?_", op) and (parts[0] == "vcc" or parts[0].startswith("s[")) else ([1] if re.match(r"(v_div_scale|v_add_co|v_sub_co|v_subrev_co|v_addc_co|v_subb_co)", op) else [])
            src_hits = [i for i, x in enumerate(parts) if dst in x and i not in dst_slots]
            reads_mask = op.startswith("v_") and bool(src_hits)
            if op.startswith("v_") and not src_hits and any(dst in parts[i] for i in dst_slots if i < len(parts)):
                break                      # the mask is redefined before anybody read it
            if reads_mask:
                if real < (4 if op.startswith("v_div_fmas") else 2) and pk > 0:      # v_div_scale -> v_div_fmas wants four wait states
                    out.append((k[:60], l, " | ".join(x[1] for x in ins[i + 1:j + 1])))
                break
            if re.match(r"(v_cmp|v_div_scale|s_)", op) and dst in rest.split(",")[0]:
                break                      # redefined
            if op.startswith("v_pk_"):
                pk += 1
            else:
                real += 1
    return out


for f in sys.argv[1:]:
    hits = scan(f)
    print(f"{f}: {len(hits)} suspicious VALU-writes-mask -> VALU-reads-mask windows filled with packed ops")
    for k, a, b in hits[:12]:
        print("   ", k, "::", a, "->", b)
