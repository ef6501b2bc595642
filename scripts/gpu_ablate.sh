#!/bin/bash
# per-phase ablation of k_splat_xy / k_gather_yx (libs from scripts/build_ablate.sh): only the ablated kernel's column is meaningful
TAG=${1:-abl}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
C=differentiable-point-clouds_amd/csrc
LIBS="$C/libdpc_hip.so $C/libdpc_abl_s0.so $C/libdpc_abl_s1.so $C/libdpc_abl_s2.so $C/libdpc_abl_s3.so $C/libdpc_abl_s4.so $C/libdpc_abl_g0.so $C/libdpc_abl_g1.so $C/libdpc_abl_g2.so"
echo "== 320,8000,64,21,3.0" | tee -a "$OUT/abl.txt"
AB_SHAPE=320,8000,64,21,3.0 timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | sed -E 's/(points_bwd|pose_finalize|zbwd|zfwd|zsort)=[0-9.]+ ?//g' | tee -a "$OUT/abl.txt"
echo "== cfg2" | tee -a "$OUT/abl.txt"
timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | sed -E 's/(points_bwd|pose_finalize|zbwd|zfwd|zsort)=[0-9.]+ ?//g' | tee -a "$OUT/abl.txt"
