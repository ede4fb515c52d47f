#!/bin/bash
# Register / LDS usage of the kernels in one object of the library:
#   tools/kernel_stats.sh igemm_fwd [grep-pattern]
set -e
OBJ=$(cd "$(dirname "$0")/../gaia_seg_amd/lib/obj" && pwd)/$1.o
TMP=$(mktemp -d /tmp/ks.XXXXXX)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$OBJ" "$TMP/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --input="$TMP/fat.bin" --output="$TMP/k.co" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/k.co" \
  | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|group_segment_fixed|vgpr_spill|agpr_count" \
  | sed 's/^ *//' | paste - - - - - -  | grep -E "${2:-.}" | sed 's/\t/ /g'
rm -rf "$TMP"
