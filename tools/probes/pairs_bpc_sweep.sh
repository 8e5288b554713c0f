for b in 3 4 5 6 7 8; do echo "BPC=$b"; RDGPU_FILL_PAIRS_BPC=$b python bench.py --no-stages --no-host --cpu-sample 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_ms_per_step']['fill.scan'], d['kernels_ms_per_step']['fill.descent'])"; done
