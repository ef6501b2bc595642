#!/bin/bash
# the chunk-sparse experiment (csrc built with -DDPC_CHUNK_SPARSE=1 -> libdpc_cs.so): GPU parity subset against the variant
# (buffers NaN-poisoned: a read of a chunk nobody wrote shows), then per-kernel A/B against the shipped library
TAG=${1:-r04cs}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
CS=differentiable-point-clouds_amd/csrc/libdpc_cs.so
DPC_GPU_LIB=$CS timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider \
   -k "goldens or knife or cfg2_full_batch_against or cfg5_full or degenerate or test_fused_dropout or fused_candidate_loss or training_shape_at or fused_l2 or asymmetric or fused_path_against or d256 or edge_planes" > "$OUT/pytest_variant.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_variant.log"; tail -4 "$OUT/pytest_variant.log"
for SH in ${AB_SHAPES:-32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3}; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py differentiable-point-clouds_amd/csrc/libdpc_hip.so $CS 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
