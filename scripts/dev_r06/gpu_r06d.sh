#!/bin/bash
# Round 6: (a) the depth sort's three forms, (b) phase ablation of k_splat_xy / k_gather_yx at cfg2 and cfg5
TAG=${1:-r06d}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
C=differentiable-point-clouds_amd/csrc
timeout 600 python -m pytest tests/test_round6_cases.py -m gpu -x -q -p no:cacheprovider -k "depth_sort or walks" > "$OUT/01_pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/01_pytest.log"; tail -3 "$OUT/01_pytest.log"
for SH in 32,8000,128,11,1.6 8,8000,128,11,1.6 64,4000,64,11,1.2 320,8000,64,21,3.0; do
  for F in 0 2; do
    echo "== $SH DPC_ZSORT_SPLIT=$F" | tee -a "$OUT/ab_zsort.txt"
    DPC_ZSORT_SPLIT=$F AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_zsort.txt"
  done
done
LIBS="$C/libdpc_hip.so $C/libdpc_abl_s0.so $C/libdpc_abl_s1.so $C/libdpc_abl_s2.so $C/libdpc_abl_s3.so $C/libdpc_abl_s4.so $C/libdpc_abl_g0.so $C/libdpc_abl_g1.so $C/libdpc_abl_g2.so $C/libdpc_abl_g3.so"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0; do
  echo "== $SH" | tee -a "$OUT/abl.txt"
  AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $LIBS 2>&1 | grep -v amdgpu.ids | sed -E 's/(points_bwd|pose_finalize|zbwd|zfwd|zsort|zhist|zscatter|memset_small)=[0-9.]+ ?//g' | tee -a "$OUT/abl.txt"
done
B() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 "$@" > "$OUT/03_bench_$NAME.json" 2> "$OUT/03_bench_$NAME.err"; echo "bench $NAME rc=$?"; python -c "
import json,sys; j=json.load(open('$OUT/03_bench_$NAME.json')); print(j['value'], j['ms_per_step'], j['timing']['ms_per_step_median'], j['roofline']['kernel_ms_per_step'])"; }
B cfg2 --steps 50 --warmup 10 --no-cpu-baseline
