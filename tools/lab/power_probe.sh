#!/bin/bash
# The wide 3x3 layer (FPN output conv at p2: B = 8, 256 x 256 pixels, 256 -> 256) on the ring kernel and on the patch kernel:
# wall time on random and on all-zero operands, HBM / L2 counters, and (lab build, -DAMP_STAMP) in-loop clock and cycles per K-step.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/power_probe
mkdir -p $OUT
cd $ROOT
{
  echo "# tools/lab/time_conv.py 8 256 256 256 256 <AMP_PATCH256>:<AMP_KORDER>   (3 alternations, 50 launches each; HIP events)"
  timeout -k 10 200 python3 tools/lab/time_conv.py 8 256 256 256 256 0:0 1:0 0:1 2>&1 | grep -v amdgpu.ids
  AMP_LAB_ZERO=1 timeout -k 10 200 python3 tools/lab/time_conv.py 8 256 256 256 256 0:0 1:0 0:1 2>&1 | grep -v amdgpu.ids
} > $OUT/time_randn_zero.txt
bash tools/lab/pmc_p256.sh > $OUT/pmc_patch_vs_ring.txt 2>&1
rm -f ampis_amd/csrc/build/conv.o && make -C ampis_amd/csrc EXTRA=-DAMP_STAMP > $OUT/stamp_build.txt 2>&1 || exit 1
{
  echo "# lab build (-DAMP_STAMP): ring kernel (AMP_PATCH256=0), then patch kernel (AMP_PATCH256=1); random operands"
  AMP_PATCH256=0 AMP_STAMP_CLOCK=1 timeout -k 10 200 python3 tools/stamp_conv.py fpn.out.p2 deconv.gemm 2>&1 | grep -v amdgpu.ids
  AMP_PATCH256=1 AMP_STAMP_TILES=patch AMP_STAMP_CLOCK=1 timeout -k 10 200 python3 tools/stamp_conv.py fpn.out.p2 2>&1 | grep -v amdgpu.ids
} > $OUT/stamps_ring_vs_patch.txt
cat $OUT/time_randn_zero.txt $OUT/stamps_ring_vs_patch.txt
tail -22 $OUT/pmc_patch_vs_ring.txt
