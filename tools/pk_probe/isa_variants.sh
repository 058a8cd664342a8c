#!/bin/bash
# Which instruction of round 3's failing box_candidates_kernel is the cause?  (results: profiles/r04/pk_hazard_isa_variants.txt)
# Builds libampis_hip.so variants whose box_infer device code is a HAND-EDITED listing of the kernel compiled WITH the SLP vectoriser (the
# code round 3 shipped before its fix), one edit per hypothesis, and runs each under three concurrent contexts (conc_boxes.py: mismatching
# batches of 630).      build (no GPU):  bash tools/pk_probe/isa_variants.sh build      run (GPU box):  bash tools/pk_probe/isa_variants.sh run
#   pk  unmodified                                                                     -> fails   (7, 16, 14 of 630 in three runs)
#   e1  s_nop 1 behind both v_cmp whose wait states were filled with v_pk_* (round 3's theory: stale VCC) -> STILL fails (18)
#   e2  s_nop 1 between v_cndmask and the v_pk_mul that consumes its result                   -> STILL fails (10)
#   e3  s_nop 7 behind the second v_exp_f32 (transcendental-result theory)                    -> STILL fails (9)
#   e4  the three `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` as scalar v_add / v_sub, every window and every other packed op untouched -> 0, 0
#   e5  s_nop 7 in front of each of those three                                               -> STILL fails (36, 39): not a wait-state matter
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); CS=$ROOT/ampis_amd/csrc; LL=/opt/rocm/lib/llvm/bin; W=$ROOT/tools/pk_probe/variants
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off -I$CS -I$ROOT/include"
if [ "$1" = build ]; then
  mkdir -p $W && cd $W
  cp $CS/box_infer.hip box_infer_pk.hip
  /opt/rocm/bin/hipcc $FL --cuda-device-only -S box_infer_pk.hip -o pk.s 2>/dev/null      # no -fno-slp-vectorize: the round-3 code
  python3 - <<'PY'
s = open('pk.s').read().split('\n')
def find(pat, start=0):
    for i in range(start, len(s)):
        if s[i].strip() == pat: return i
    raise SystemExit('listing changed, missing: ' + pat)
a = find('v_cmp_ngt_f32_e32 vcc, s35, v3'); b = find('v_cmp_nlt_f32_e32 vcc, s76, v3', a)
c = find('v_cndmask_b32_e32 v29, v34, v4, vcc', b); d = find('v_cndmask_b32_e32 v25, v34, v25, vcc'); e = find('v_exp_f32_e32 v26, v26')
A = find('v_pk_add_f32 v[28:29], v[26:27], v[24:25] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]')
B = find('v_pk_add_f32 v[26:27], v[26:27], v[24:25] op_sel:[0,1] op_sel_hi:[1,0]')
C = find('v_pk_add_f32 v[24:25], v[40:41], v[42:43] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]')
def variant(name, inserts=(), replace=()):
    out = list(s)
    for i, t in replace: out[i] = '\t' + t
    for i, t in sorted(inserts, reverse=True): out.insert(i, '\t' + t)
    open(name, 'w').write('\n'.join(out))
variant('e1.s', [(a + 1, 's_nop 1'), (b + 1, 's_nop 1')])
variant('e2.s', [(c + 1, 's_nop 1'), (d + 1, 's_nop 1')])
variant('e3.s', [(e + 1, 's_nop 7')])
variant('e4.s', replace=[(A, 'v_sub_f32_e32 v28, v26, v25'), (B, 'v_add_f32_e32 v26, v26, v25'), (C, 'v_sub_f32_e32 v24, v40, v43')])
variant('e5.s', [(A, 's_nop 7'), (B, 's_nop 7'), (C, 's_nop 7')])
PY
  for v in pk e1 e2 e3 e4 e5; do
    $LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $v.s -o $v.dev.o
    $LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $v.out $v.dev.o
    $LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$v.out -output=$v.hipfb
    /opt/rocm/bin/hipcc $FL --cuda-host-only -c box_infer_pk.hip -Xclang -fcuda-include-gpubinary -Xclang $v.hipfb -o $v.host.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libamp_$v.so $(ls $CS/build/*.o | grep -v box_infer.o) $v.host.o
  done
  rm -f *.o *.out *.hipfb; ls -la $W/*.so
else
  O=$ROOT/gpurun_out/hazard; mkdir -p $O
  for v in pk e1 e2 e3 e4 e5 pk e4; do
    AMP_LIB=$W/libamp_$v.so NT=3 timeout -k 10 300 python3 $ROOT/tools/pk_probe/conc_boxes.py > $O/$v.$RANDOM.log 2>&1
    echo "variant $v: $(grep -h '^threads' $O/$v.*.log | tail -1 | cut -c1-160)"
  done
  NT=3 timeout -k 10 300 python3 $ROOT/tools/pk_probe/conc_boxes.py > $O/shipped.log 2>&1; echo "shipped library: $(grep '^threads' $O/shipped.log | cut -c1-160)"
fi
