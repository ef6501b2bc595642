#!/bin/bash
# Ablation builds of the product library: libdpc_abl_<name>.so returns early from one kernel phase.
# usage: build_ablate.sh  (writes into differentiable-point-clouds_amd/csrc/, git-ignored)
cd "$(dirname "$0")/../differentiable-point-clouds_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -w -shared"
for n in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DDPC_ABLATE_SPLAT=$n -o libdpc_abl_s$n.so dpc_kernels.hip &
done
for n in 0 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DDPC_ABLATE_GATHER=$n -o libdpc_abl_g$n.so dpc_kernels.hip &
done
wait
ls -la libdpc_abl_*.so
