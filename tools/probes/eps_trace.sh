#!/bin/bash
# PriorityFloodEpsilon at 40000^2 under rocprofv3 --kernel-trace: the sequence of k_eps_relax launches (duration by round)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/epst && mkdir -p /tmp/epst
rocprofv3 --kernel-trace --output-format csv -d /tmp/epst -- python $GRAFT_REPO_ROOT/tools/ab_stage.py --stage eps --size 40000 --reps 1 --cfg "" > /tmp/epst/out.txt 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/eps_trace.txt
import csv, glob
rows = []
for path in glob.glob('/tmp/epst/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rel = [r for r in rows if 'k_eps_relax' in r['Kernel_Name']]
# the LAST call's rounds: split by k_eps_init
inits = [i for i, r in enumerate(rows) if 'k_eps_init' in r['Kernel_Name']]
last = rows[inits[-1]:]
rel = [r for r in last if 'k_eps_relax' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rel]
print("rounds", len(d), "total_ms", sum(d) / 1e3)
for a, b in ((0, 4), (4, 16), (16, 64), (64, 128), (128, 256), (256, 400), (400, 10000)):
    seg = d[a:b]
    if seg: print(f"rounds {a}-{min(b, len(d))}: sum {sum(seg) / 1e3:.2f} ms, mean {sum(seg) / len(seg):.1f} us, max {max(seg):.1f} us")
comp = [r for r in last if 'k_tiles_compact' in r['Kernel_Name']]
dc = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in comp]
print("compact launches", len(dc), "total_ms", sum(dc) / 1e3)
t0, t1 = int(last[0]['Start_Timestamp']), int(last[-1]['End_Timestamp'])
print("span_ms", (t1 - t0) / 1e6, "kernel_ms", sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6)
PY
tail -1 /tmp/epst/out.txt | cut -c1-400 >> $GRAFT_REPO_ROOT/gpurun_out/eps_trace.txt
