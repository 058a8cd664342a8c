"""Every kernel file's gfx950 ISA must be free of the packed-FP32 operand form that returns wrong lanes on this hardware while kernels of
other queues run (round 4: v_pk_{add,mul,fma}_f32 with op_sel selecting src1's HIGH dword for the LOW result -- measured by
tools/pk_probe, the cause of round 3's wrong boxes; csrc/common.h, csrc/Makefile NOSLP).  The listings come from `make listings`, i.e. with
exactly the flags the library is built with, per file.  No GPU needed."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ampis_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_packed_fp32_instruction_selects_the_high_half_of_src1(tmp_path):
    import scan_pk_opsel as scan
    lst = str(tmp_path / "lst")
    subprocess.run(["make", "-C", CSRC, "-j6", "listings", f"LISTDIR={lst}"], check=True, stdout=subprocess.DEVNULL)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    listings = sorted(glob.glob(os.path.join(lst, "*.s")))
    assert len(srcs) >= 15 and len(listings) == len(srcs)
    bad = {os.path.basename(l): scan.scan(l) for l in listings}
    bad = {k: v[:3] for k, v in bad.items() if v}
    assert not bad, bad
    # the scan had something to look at: the hand-packed kernels do use packed FP32
    packed = {os.path.basename(l): sum(scan.census(l).values()) for l in listings}
    assert packed["conv.s"] > 1000 and packed["train_bwd.s"] > 100 and packed["roi_align.s"] > 10, packed
    # no kernel may contain a call (round 3's per-kernel target attribute made the HIP headers' inline functions calls; it is gone)
    calls = {}
    for l in listings:
        name = None
        for line in open(l):
            if line.startswith("_Z") and line.rstrip().endswith(":"):
                name = line.split(":")[0]
            elif "s_swappc_b64" in line and name:
                calls[name] = calls.get(name, 0) + 1
    assert not calls, calls
    # Codegen guard of the hot kernels (round 4: a run-time flag in conv_split_kernel's staging lambda spilled 12 B of scratch at 250 VGPRs and put a
    # vmcnt(0) into every K-step -- 35 % slower on every launch, and every test stayed green): no scratch, and the ring kernels' K loops keep
    # their counted waits (the full drains that exist are in prologue / epilogue: 15 in the plain instantiations).
    import re
    def kernels(path):
        txt = open(path).read()
        for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)s_endpgm(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
            yield m.group(1), m.group(2), m.group(3)
    hot = {"conv.s": ("conv_split_kernel", "conv3x3_c64_kernel", "conv_glds_kernel", "conv_f16x3_kernel", "stem_pool"), "wgrad.s": ("wgrad_split_kernel", "wgrad_f16x3_kernel"),
           "grouped_bwd.s": ("grouped_wgrad9_kernel",), "roi_align.s": ("roi_align_split",)}
    seen = 0
    for fname, keys in hot.items():
        for name, body, meta in kernels(os.path.join(lst, fname)):
            if not any(k in name for k in keys):
                continue
            seen += 1
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
            assert scratch == 0 and "scratch_" not in body, f"{name}: {scratch} B of scratch in a hot kernel"
            if "conv_split_kernel" in name and "ELi3ELi3E" not in name:      # plain instantiations (EPI 1 / 2); the fused tails (EPI 3) drain more in their epilogues
                drains = sum(1 for l in body.split("\n") if "s_waitcnt vmcnt(0)" in l)
                assert drains <= 16, f"{name}: {drains} full vmcnt drains (15 before: one inside the K loop would serialise the LDS-DMA ring)"
    assert seen >= 40, seen
    # the scanner does find the form where it is known to be: box_infer.hip built WITH the SLP vectoriser (round 3's failing kernel)
    out = tmp_path / "box_infer_slp.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                    os.path.join(CSRC, "box_infer.hip"), "-o", str(out)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    hits = scan.scan(str(out))
    assert hits and all("box_candidates_kernel" in h[0] for h in hits), hits
    assert all("op_sel:[0,1]" in h[1] for h in hits)
