#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "LDS" | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*" | sort -u | head -40 > $GRAFT_REPO_ROOT/gpurun_out/lds_counters.txt
rm -rf /tmp/lp && mkdir -p /tmp/lp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN --output-format csv -d /tmp/lp -- python $GRAFT_REPO_ROOT/tools/ab_stage.py --stage ${STAGE:-accum} --size 40000 --reps 1 --cfg "" > /tmp/lp/out.txt 2>&1
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/lds_probe.txt
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for path in glob.glob('/tmp/lp/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        per[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, v in per.items():
    if any(x in k for x in ("acc_","descent","scan","finalize","ccl","relax","eps")): print(k, dict(v))
PY
tail -3 /tmp/lp/out.txt | cut -c1-300
