#!/bin/bash
# Register, scratch and spill figures of every trace_kernel instantiation as the library is built
# (the compiler's -Rpass-analysis=kernel-resource-usage remarks, one line per kernel).
# Usage: tools/kernel_usage.sh [source dir]    default: ray_tracing_amd/csrc
dir=${1:-ray_tracing_amd/csrc}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize \
  -mllvm -amdgpu-use-amdgpu-trackers=1 $EXTRA -x hip -c "$dir/rt_kernels.hip" --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -o /dev/null 2>&1 |
  grep -E "Function Name|TotalSGPRs:| VGPRs:|Scratch|Spill|Occupancy|LDS Size" |
  sed -e 's/^[^ ]* remark: *//' -e 's/ \[-Rpass.*//' -e 's/Function Name: _ZN2rt12_GLOBAL__N_1[0-9]*//' \
      -e 's/EEvNS_[0-9]*[A-Za-z]*E$//' | paste - - - - - - - - |
  sed -e 's/ILb0/<0/' -e 's/ILb1/<1/' -e 's/ELb0/,0/g' -e 's/ELb1/,1/g' -e 's/\t/  /g'
