#!/bin/bash
# diagnostic build of the whole library with extra flags:  tools/build_variant.sh <name> <flags...>  ->  build_variants/lib_<name>.so
name=$1; shift
cd "$(dirname "$0")/../dycon_paper_replication_amd/csrc" || exit 1
mkdir -p /tmp/bv_$name ../../build_variants
for f in conv norm spatial losses reflosses optim eval; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $f.hip -o /tmp/bv_$name/$f.o || exit 1 &
done
wait
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -c error.cpp -o /tmp/bv_$name/error.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/lib_$name.so /tmp/bv_$name/*.o && echo built build_variants/lib_$name.so
