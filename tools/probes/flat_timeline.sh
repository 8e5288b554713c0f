#!/bin/bash
# timeline of ONE directions + flat resolution call at S3 (rocprofv3 kernel trace): start offset, duration, gap to the previous
# kernel's end on the same queue, queue, kernel -> gpurun_out/flat_timeline.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt && mkdir -p /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $GRAFT_REPO_ROOT/tools/ab_stage.py --stage flats --size ${1:-40000} --reps 2 --cfg "${2:-}" > /tmp/rt/out.txt 2>&1
f=$(find /tmp/rt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/flat_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_dirs_classify" in r["Kernel_Name"]]
# the second-to-last classification starts a timed (not instrumented) repetition
start = idx[-2] if len(idx) >= 2 else idx[-1]
end = idx[-1] if len(idx) >= 2 else len(rows)
rows = rows[start:end]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    short = n.split("(")[0].replace("void rdgpu::", "").replace("rdgpu::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:7.1f}  q{q}  {short}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
PY
tail -3 /tmp/rt/out.txt | cut -c1-300
