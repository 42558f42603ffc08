#!/bin/bash
# Developer tool: assembly + resource summary of render_kernel_sm<false,false,false,true> (the kernel C3 / C5 run) only.
#   bash tools/asm_sm.sh out.s [extra -D flags]
out=$1; shift
cd $(dirname $0)/../pyrite_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-fast-math -ffp-contract=off -mllvm -disable-vector-combine -fno-slp-vectorize \
  --cuda-device-only -DPYR_DEV_ONLY_SM "$@" -S kernels.hip -o $out.all 2>&1 | grep -v hip-link
awk '/^_ZN3pyr16render_kernel_smILb0ELb0ELb0ELb1ELb0ELb0EE.*:/{f=1} f{print} /s_endpgm/{if(f){exit}}' $out.all > $out
awk '/\.name:.*render_kernel_smILb0ELb0ELb0ELb1ELb0ELb0EE/{f=1} f&&/private_segment_fixed_size|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill/{printf "%s ", $0} f&&/\.wavefront_size/{print ""; exit}' $out.all
echo "VALU $(grep -c '^\s*v_' $out)  lines $(wc -l < $out)"
