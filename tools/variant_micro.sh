#!/bin/bash
# micro-benchmarks of the k=3 LDS-halo kernel over diagnostic builds (build_variants/lib_<name>.so, selected through DYCON_LIB)
for lib in build_variants/lib_*.so; do
  echo "=== $lib"
  for shape in "32 32 48" "64 64 24" "16 32 48" "32 16 48" "64 128 24" "48 16 96"; do
    DYCON_LIB=$PWD/$lib python tools/conv_micro.py $shape 30 || exit 1
  done
done
