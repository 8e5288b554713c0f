import sys, time, numpy as np, torch, ctypes
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import richdem_amd as rd
from richdem_amd import _lib
lib = _lib.lib()
for n in (10000, 40000):
    z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(z, seed=3)
    d = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t = time.time()
    rc = lib.rdgpu_pf_flowdirs_dev_f32(ctypes.c_void_p(z.data_ptr()), ctypes.c_float(-9999.0), n, n, ctypes.c_void_p(d.data_ptr()), None)
    torch.cuda.synchronize(); dt = time.time() - t
    print(n, "rc", rc, "seconds", round(dt, 3), rd.pf_flowdirs_stats(), "Mcells/s", round(n * n / 1e6 / dt, 1), flush=True)
    del z, d
