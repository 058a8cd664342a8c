"""Scan gfx950 ISA listings (`make -C ampis_amd/csrc listings`, or any `hipcc -S`) for the packed-FP32 operand form that returns wrong lanes
on this hardware while kernels of other queues run (round 4; tools/pk_probe/pk_opsel_probe.hip measures it directly):

    v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32  with  op_sel:[x,1(,x)]      -- src1's HIGH dword selected for the LOW result

In one pass of the instruction (16 consecutive lanes) src1 is then read as 0: ~1e-4 of the executions beside two inference threads, never
alone; op_sel_hi, src0 / src2 selects, v_fma_mix_f32 and v_pk_mov_b32 never fail (1.7e9 executions per form).  box_candidates_kernel's
`x2 = cx + 0.5 w` was such an instruction (the SLP vectoriser's shuffle) -- hence boxes whose x2 was the box centre, for 16 proposals in a row.
Round 3 had blamed VALU-mask wait states; hand-edited ISA variants of the kernel refuted that (the wait states made real: still failing;
only the three op_sel adds made scalar, every window left as it was: 0 failures), so the old window scan is gone.

usage: python tools/scan_pk_opsel.py file.s [...]      -> exit status 1 when a listing contains the form"""
import re
import sys

PK = re.compile(r"\s*(v_pk_(?:add|mul|fma)_f32)\s+(.*?)\bop_sel:\[([01,]+)\]")


def scan(path):
    """[(kernel, instruction)] for every packed-FP32 instruction of the listing whose op_sel bit of src1 is set"""
    kernel, out = "?", []
    for raw in open(path):
        m = re.match(r"(_Z\w+):", raw)
        if m:
            kernel = m.group(1)
            continue
        m = PK.match(raw)
        if m and len(m.group(3).split(",")) >= 2 and m.group(3).split(",")[1] == "1":
            out.append((kernel, raw.strip()))
    return out


def census(path):
    """kernel -> number of packed-FP32 instructions (any form): what the scan looked at"""
    kernel, n = "?", {}
    for raw in open(path):
        m = re.match(r"(_Z\w+):", raw)
        if m:
            kernel = m.group(1)
        elif re.match(r"\s*v_pk_(add|mul|fma)_f32\s", raw):
            n[kernel] = n.get(kernel, 0) + 1
    return n


if __name__ == "__main__":
    bad = 0
    for f in sys.argv[1:]:
        hits = scan(f)
        bad += len(hits)
        print(f"{f}: {sum(census(f).values())} packed-FP32 instructions, {len(hits)} with src1.hi -> lo (op_sel:[x,1])")
        for k, ins in hits[:12]:
            print("   ", k[:70], "::", ins)
    sys.exit(1 if bad else 0)
