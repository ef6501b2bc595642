#!/bin/bash
# Round 6: balanced 1024-thread work-groups of the z kernels (DPC_ZBIG) against the 256-thread form, same library
TAG=${1:-r06b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$REPO"
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
C=differentiable-point-clouds_amd/csrc
timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_r06.log" 2>&1
echo "pytest r06 exit $?" | tee -a "$OUT/01_pytest_r06.log"; tail -3 "$OUT/01_pytest_r06.log"
DPC_ZBIG=1 timeout 900 python -m pytest tests/test_round6_cases.py tests/test_chunk_sparse.py -m gpu -x -q -p no:cacheprovider > "$OUT/01_pytest_r06_big.log" 2>&1
echo "pytest r06 (DPC_ZBIG=1) exit $?" | tee -a "$OUT/01_pytest_r06_big.log"; tail -3 "$OUT/01_pytest_r06_big.log"
for SH in 32,8000,128,11,1.6 8,16000,256,11,2.0 320,8000,64,21,0.8 320,8000,64,21,0.3 32,8000,128,7,1.0; do
  for Z in 0 1; do
    echo "== $SH DPC_ZBIG=$Z" | tee -a "$OUT/ab.txt"
    DPC_ZBIG=$Z AB_SHAPE=$SH timeout 300 python scripts/ab_libs.py $C/libdpc_hip.so $C/libdpc_hip.so@walk0 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab.txt"
  done
done
B() { NAME=$1; shift; timeout 400 python bench.py --gpus 1 "$@" > "$OUT/03_bench_$NAME.json" 2> "$OUT/03_bench_$NAME.err"; echo "bench $NAME rc=$?"; python -c "
import json,sys; j=json.load(open('$OUT/03_bench_$NAME.json')); print(j['value'], j['ms_per_step'], j['timing']['ms_per_step_median'], j['roofline']['kernel_ms_per_step'])"; }
B cfg2 --steps 50 --warmup 10 --no-cpu-baseline
B cfg5 --steps 30 --warmup 5 --config 5 --no-cpu-baseline
du -sh "$OUT"
