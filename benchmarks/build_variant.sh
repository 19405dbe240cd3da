#!/bin/bash
# build_variant.sh NAME [-Dflag ...]: benchmarks/zpn_variant.hip with the flags -> benchmarks/bin/lib_NAME.so
# (the library's other objects as `make -C openseize_amd/csrc` left them).  Then, on the GPU box:
#   benchmarks/bin/ab_chain -r 7 benchmarks/bin/lib_base.so benchmarks/bin/lib_NAME.so ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/benchmarks/bin"
O=$root/openseize_amd/lib/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I"$root/include" "$@" \
    -c "$root/benchmarks/zpn_variant.hip" -o "$root/benchmarks/bin/zpn6_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/benchmarks/bin/lib_$name.so" $O/lib.o $O/sos.o $O/fir.o $O/chain.o \
    $O/chain_spec.o $O/chain_zp.o $O/chain_zpn_2.o $O/chain_zpn_4.o "$root/benchmarks/bin/zpn6_$name.o" $O/chain_zpn_8.o \
    $O/poly.o $O/spec.o $O/misc.o $O/rccl.o $O/glue.o $O/hostpool.o -L/opt/rocm/lib -lrocfft -ldl -pthread -Wl,-rpath,/opt/rocm/lib
echo "built benchmarks/bin/lib_$name.so"
