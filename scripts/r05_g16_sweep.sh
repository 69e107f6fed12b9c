#!/bin/bash
# Round 5: DMA placement tables (Place<0..6>) of Kernel16 on the row/row and row/contraction-major products (variant library
# built with scripts/build_gemm_variant.sh g16sweep -DTN_G16_SWEEP)
L=$(pwd)/touchnet_amd/_lib/variants/g16sweep/libtouchnet_amd.so
for pl in ${PLACES:-6 0 1 2 3 4 5 6}; do
  echo "--- Place<$pl>"
  TN_AMD_LIB=$L TN_G16_PLACE=$pl python scripts/r05_mode_bench.py 2>&1 | grep -E "dgrad MLP"
done
