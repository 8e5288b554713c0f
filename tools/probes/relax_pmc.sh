#!/bin/bash
# per-dispatch SQ counters of the relaxation kernels of one flat resolution -> gpurun_out/relax_pmc.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/tools/probes/flat_only.py "$@" > /tmp/pm.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
python - "$f" <<PY
import csv,sys,os,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.OrderedDict()
for r in rows:
    if not any(k in r["Kernel_Name"] for k in ("k_relax_bits", "k_flat_relax", "k_relax_planes", "k_flat_dirs_q", "k_planes_prepare")): continue
    k=(int(r["Dispatch_Id"]))
    d.setdefault(k,{"name":r["Kernel_Name"][:40],"grid":r.get("Grid_Size")})[r["Counter_Name"]]=float(r["Counter_Value"])
with open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/relax_pmc.txt","w") as f:
    for k,v in d.items():
        f.write(" ".join([str(k),v["name"].replace(" ","_"),str(v["grid"])]+[f"{c}={v.get(c,0):.0f}" for c in ("SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES")])+"\n")
PY
tail -3 /tmp/pm.log
