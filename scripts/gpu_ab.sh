#!/bin/bash
# dev: per-kernel A/B of library builds on the headline shapes (interleaved rounds in one process), after a parity subset
# on the default build.   gpurun -- 'LIBS="libdpc_prev.so libdpc_hip.so libdpc_x.so" bash scripts/gpu_ab.sh r05ab'
TAG=${1:-r05ab}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests/test_chunk_sparse.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider \
     -k "${TESTS_K:-chunk or goldens or knife or cfg2_full_batch or cfg5_full or degenerate or fused_dropout or fused_candidate_loss or training_shape_at or fused_l2 or asymmetric or fused_path_against or d256 or edge_planes}" > "$OUT/pytest.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
fi
P=""; for L in ${LIBS:-libdpc_prev.so libdpc_hip.so}; do P="$P $C/$L"; done
for SH in ${AB_SHAPES:-32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,3.0 320,8000,64,21,0.8 320,8000,64,21,0.3}; do
  echo "== $SH" | tee -a "$OUT/ab.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $P 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
done
