#!/bin/bash
# What each phase of the persistent pair pass costs: the fill's per-kernel times with phases switched off
# (RDGPU_FILL_PAIRS_ABLATE: 4 = pair loop, 8 = proposals + records, 16 = boundary list; k_pairs16 only: 32 = the pair loop
# without its table updates, 64 = the pair loop's gathers alone; results are wrong then, times are not)
# The switch is compiled in only with -DRDGPU_PROBES (r06): build with  make -C richdem_amd/csrc clean all CXXFLAGS+=" -DRDGPU_PROBES"  first.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for cfg in "${@:-RDGPU_FILL_PAIRS_BPC=6}"; do
for ab in 0 4 8 16 32 64; do
  env $cfg RDGPU_FILL_PAIRS_ABLATE=$ab python bench.py --steps 3 --no-stages --no-host --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels_ms_per_step']
print('$cfg ablate $ab: fill.scan', k.get('fill.scan'), 'fill', d['ms_per_step'])"
done; done | tee gpurun_out/pairs_ablate.txt
