mkdir -p gpurun_out/r04q
export DPC_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04q/share2_cfg2.json 2> gpurun_out/r04q/share2_cfg2.err; echo "cfg2 rc=$?"
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --scaling strong > gpurun_out/r04q/share2_cfg2_strong.json 2> gpurun_out/r04q/share2_cfg2_strong.err; echo "strong rc=$?"
timeout 400 python bench.py --gpus 2 --config 3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04q/share2_train.json 2> gpurun_out/r04q/share2_train.err; echo "train rc=$?"
timeout 400 python bench.py --gpus 2 --config 3 --graph --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04q/share2_train_graph.json 2> gpurun_out/r04q/share2_train_graph.err; echo "train graph rc=$?"
for f in gpurun_out/r04q/*.json; do echo $f; head -c 600 $f; echo; done
tail -3 gpurun_out/r04q/*.err
