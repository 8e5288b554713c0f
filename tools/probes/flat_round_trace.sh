#!/bin/bash
# per-launch durations of the bitmap search's rounds (one flats stage at S3), from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt && mkdir -p /tmp/rt
rocprofv3 --kernel-trace --output-format csv -d /tmp/rt -- python $GRAFT_REPO_ROOT/tools/ab_stage.py --stage flats --size 40000 --reps 1 --cfg "" > /tmp/rt/out.txt 2>&1
f=$(find /tmp/rt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/round_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last stage call: take the last contiguous series after the last k_flat_classify
idx = [i for i, r in enumerate(rows) if "k_flat_classify" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
prev_end = None
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    short = n.split("(")[0][-40:]
    print(f"{short}\t{(e - s) / 1e3:.1f}\t{gap:.1f}")
PY
grep -v "rocprofv3\|Opened result" /tmp/rt/out.txt | tail -15
