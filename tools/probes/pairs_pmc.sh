#!/bin/bash
# HBM bytes fetched / written by the fill's kernels (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes), fill only
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pp_$c -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-host --no-stages > /tmp/pp_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for c, mul in (("FETCH_SIZE", 2), ("WRITE_SIZE", 1)):
    per = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(f'/tmp/pp_{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != c: continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][0] += 1; per[k][1] += float(row["Counter_Value"])
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:6]:
        print(c, k[:60], v[0], round(v[1] * 1024 * mul / 1e9 / v[0], 2), "GB/launch")
PY
