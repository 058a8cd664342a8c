"""VGPR / SGPR / scratch and instruction mix of the kernels in a `hipcc -S --cuda-device-only` listing whose mangled name contains a substring.
python tools/isa_stats.py /tmp/conv.s conv_split_kernel      -- a kernel at 250 VGPRs spills on the smallest change: check before every GPU run."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)s_endpgm(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    name, body, meta = m.group(1), m.group(2), m.group(3)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    g = lambda k: re.search(rf"\.amdhsa_{k} (\d+)", meta).group(1)
    print(f"{name[:100]:100s} vgpr {g('next_free_vgpr'):>3} sgpr {g('next_free_sgpr'):>3} scratch {g('private_segment_fixed_size'):>4} B | {len(ins)} instr, "
          f"waitcnt {c['s_waitcnt']} (vmcnt(0): {sum(1 for l in body.split(chr(10)) if 's_waitcnt vmcnt(0)' in l)}), mfma {sum(v for k, v in c.items() if 'mfma' in k)}, "
          f"scratch ops {sum(v for k, v in c.items() if 'scratch' in k)}")
