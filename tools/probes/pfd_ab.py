"""PriorityFloodFlowdirs at 40000^2, a level flood per pass against the tree iteration (RDGPU_PFD_TREE_ITER=1), two calls each in one
process (the first call of the process also grows the workspace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import richdem_amd as rd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(z, seed=3)
d = torch.empty((n, n), dtype=torch.uint8, device="cuda")
ref = None
var = sys.argv[2] if len(sys.argv) > 2 else "RDGPU_PFD_TREE_ITER"   # the switch to alternate: its =0 against its =1
for mode in ("0", "1", "0", "1"):
    os.environ[var] = mode
    torch.cuda.synchronize(); t = time.time()
    rd.pf_flowdirs_dev(z, -9999.0, d)
    torch.cuda.synchronize(); dt = time.time() - t
    same = None if ref is None else bool(torch.equal(ref, d))
    if ref is None:
        ref = d.clone()
    print(var, mode, "seconds", round(dt, 3), rd.pf_flowdirs_stats(), "same_as_first", same, flush=True)
