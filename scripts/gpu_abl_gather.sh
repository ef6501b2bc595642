cd $GRAFT_REPO_ROOT
C=differentiable-point-clouds_amd/csrc
for SH in 320,8000,64,21,3.0 8,16000,256,11,2.0; do
echo "== $SH"
AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_abl_g0.so $C/libdpc_abl_g1.so $C/libdpc_abl_g2.so $C/libdpc_abl_g3.so 2>&1 | grep -v amdgpu.ids | sed -E 's/(points_bwd|pose_finalize|zbwd|zfwd|zsort|splat_xy|zhist|zscatter)=[0-9.]+ ?//g'
done
