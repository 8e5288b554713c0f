import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, richdem_amd as rd
from richdem_amd.sharded import GpuShardEngine, row_split
n, S = 40000, 8
Z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(Z, seed=3)
blocks = [Z[a:b] for a, b in row_split(n, S)]
for rep in range(2):
    e = GpuShardEngine(); k, ed = e.begin_dev(blocks[3].clone(), True, True, 8); e.abort()
rd.profile_reset(); rd.profile_enable(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
e = GpuShardEngine(); blk = blocks[3].clone(); k, ed = e.begin_dev(blk, True, True, 8)
torch.cuda.synchronize(); t1 = time.perf_counter()
rd.profile_enable(False)
tot = rd.profile_totals()
print(json.dumps({"begin_ms": round((t1 - t0) * 1e3, 2), "kernels": {k: [round(v[0], 3), v[1]] for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]}, "sum": round(sum(v[0] for v in tot.values()), 2)}))
