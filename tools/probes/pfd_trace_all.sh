#!/bin/bash
# PriorityFloodFlowdirs at 40000^2 (equal elevations: the tie-order passes) under rocprofv3 --kernel-trace, ONE call, every
# kernel: where the 18 s go -- floods (fill kernels per nesting level) against the rank computations (sorts, tree passes)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pfdt && mkdir -p /tmp/pfdt
cat > /tmp/pfd_one.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, richdem_amd as rd
n = 40000
z = torch.empty((n, n), dtype=torch.float32, device="cuda"); rd.synth_dem_dev(z, seed=3)
d = torch.empty((n, n), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.time()
rd.pf_flowdirs_dev(z, -9999.0, d)
torch.cuda.synchronize(); print("wall_s", round(time.time() - t, 4), rd.pf_flowdirs_stats(), flush=True)
PY
RDGPU_PFD_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfdt -- python /tmp/pfd_one.py > /tmp/pfdt/out.txt 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pfd_trace_all.txt
import csv, glob, collections
rows = []
for path in glob.glob('/tmp/pfdt/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'k_synth' not in r['Kernel_Name']]
t0, t1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
per = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
    dt = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    per[k][0] += 1; per[k][1] += dt; busy += dt
print("span_ms", (t1 - t0) / 1e6, "kernel_ms", busy / 1e6, "launches", len(rows))
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{v[1] / 1e6:10.2f} ms {v[0]:7d}  {k}")
PY
grep -E "pfd tie pass|wall_s" /tmp/pfdt/out.txt >> $GRAFT_REPO_ROOT/gpurun_out/pfd_trace_all.txt
