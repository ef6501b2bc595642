import sys, time, torch
sys.path.insert(0,'.')
import bench
for thr in (16, 32, 64, 128):
    torch.set_num_threads(thr)
    t0=time.time(); r=bench.cpu_baseline(2, 6.0); print(thr, round(r['value'],2), 'views/s', round(time.time()-t0,1),'s', flush=True)
