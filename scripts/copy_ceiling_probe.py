import sys, json, torch
sys.path.insert(0,'/root/repo')
import dpc_amd, bench
print(json.dumps(bench.copy_ceiling(dpc_amd.get_library(), torch.device('cuda')), indent=1))
