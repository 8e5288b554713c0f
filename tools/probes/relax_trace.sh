#!/bin/bash
# kernel trace of the relaxation rounds of one flat resolution at S3 -> gpurun_out/relax_trace.txt (start us, duration us, kernel, grid)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/probes/flat_only.py "$@" > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv,sys,os
rows=list(csv.DictReader(open(sys.argv[1])))
keys=("k_flat_relax","k_tiles_compact","k_flat_init_towards","k_flat_seed","k_relax_bits","k_bits_prepare")
rel=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Grid_Size_X") or r.get("Grid_Size")) for r in rows if any(k in r["Kernel_Name"] for k in keys)]
rel.sort()
os.makedirs(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out",exist_ok=True)
with open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/relax_trace.txt","w") as f:
    t0=rel[0][0]
    for s,e,k,g in rel:
        nm="relax" if ("k_flat_relax" in k or "k_relax_bits" in k) else "compact" if "compact" in k else "init" if "init" in k else "seed" if "seed" in k else "prep"
        f.write(f"{(s-t0)/1e3:.1f} {(e-s)/1e3:.1f} {nm} {g}\n")
PY
