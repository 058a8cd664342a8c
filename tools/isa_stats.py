"""ISA summary of one kernel family in a hipcc -S listing: registers, LDS, scratch, MFMA / LDS-DMA / barrier / waitcnt counts.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S file.hip -o /tmp/x.s
       python tools/isa_stats.py /tmp/x.s conv_split_kernel"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\S*' + re.escape(pat) + r'\S*):', s, flags=re.M):
    name = m.group(1)
    end = s.find('s_endpgm', m.end())
    code = s[m.end():end]
    k = s.find('.amdhsa_kernel ' + name)
    meta = s[k:s.find('.end_amdhsa_kernel', k)]
    g = lambda key: (re.search(re.escape(key) + r'\s+(\S+)', meta) or [None, None])[1]
    dem = re.search(r'; -- Begin function (\S+)', s[max(0, m.start() - 400):m.start()])
    n_dma = len(re.findall(r'buffer_load_dwordx4[^\n]*lds', code))
    n_vm0 = len(re.findall(r'vmcnt\(0\)', code))
    print(name[:100])
    print(f"   vgpr {g('.amdhsa_next_free_vgpr')} accum_offset {g('.amdhsa_accum_offset')} sgpr {g('.amdhsa_next_free_sgpr')} lds {g('.amdhsa_group_segment_fixed_size')} "
          f"scratch {g('.amdhsa_private_segment_fixed_size')} | mfma {code.count('v_mfma')} lds-dma {n_dma} "
          f"ds_read {code.count('ds_read')} barrier {code.count('s_barrier')} waitcnt {len(re.findall(r's_waitcnt', code))} "
          f"vmcnt(0) {n_vm0} scratch-ops {code.count('scratch_')}")
