import numpy as np, richdem_amd as rd, oracle
from richdem_amd.synth import fractal_dem
import sys
sys.path.insert(0,'tests')
from test_mfd_gpu import ulp_diff_f32
dem = fractal_dem(300,220,301); nd=np.float32(-9999)
for m,x in [("Holmgren",2.0),("Holmgren",0.5),("Freeman",1.1),("Freeman",4.0),("Holmgren",8.0)]:
    got = rd.FlowProportions(dem, m, nodata=nd, exponent=x); exp = oracle.port.fm_mfd(dem, nd, m, x)
    u = ulp_diff_f32(got, exp)
    print(m, x, "max ulp", u.max(), "count>0", (u>0).sum(), "count>1", (u>1).sum(), "of", (exp>0).sum())
    idx = np.argwhere(u>1)[:3]
    for i in idx:
        y,xx,k = i
        print("  cell", y, xx, "got", got[y,xx], "exp", exp[y,xx])
