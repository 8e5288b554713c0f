#!/bin/bash
# timeline of ONE directions + flat resolution call at S3 (start offset, duration, queue) from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt && mkdir -p /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $GRAFT_REPO_ROOT/tools/ab_stage.py --stage ${1:-flats} --size 40000 --reps 1 --cfg "" > /tmp/rt/out.txt 2>&1
f=$(find /tmp/rt -name '*kernel_trace.csv' | head -1)
python - "$f" ${2:-k_dirs_classify} <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/flat_timeline_${1:-flats}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
print("kernel\tstart_ms\tdur_ms\tqueue")
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rdgpu::", "")[:46]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n}\t{(s - t0) / 1e6:.3f}\t{(e - s) / 1e6:.3f}\t{r.get('Queue_Id', '')}")
PY
grep -v "rocprofv3\|Opened result" /tmp/rt/out.txt | tail -3
