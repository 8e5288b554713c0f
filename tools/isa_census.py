#!/usr/bin/env python3
"""Static instruction census of one kernel of a hipcc --save-temps .s file, split at its s_barriers (and optionally at
labels): VALU / SALU / LDS / VMEM counts per section -- how the instruction-count work on k_pairs16 / k_descent16 was steered
(the kernels are bound by VALU issue: tools/probes/valu_issue.hip).

    hipcc -O3 ... --save-temps -c richdem_amd/csrc/fill.hip   (in a scratch directory)
    python tools/isa_census.py fill-hip-amdgcn-amd-amdhsa-gfx950.s k_pairs16IfLi8ELb1 [--labels]
"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    s = open(path).read()
    m = re.search(r'^(_Z\w*' + re.escape(pat) + r'\w*):[^\n]*\n', s, re.M)
    if not m:
        raise SystemExit("kernel not found")
    start = m.end()
    end = s.index('.Lfunc_end', start)
    kinds = ('valu', 'salu', 'lds', 'vmem', 'other')
    cur = dict.fromkeys(kinds, 0)
    segs, names = [], ["entry"]
    for ln in s[start:end].split('\n'):
        t = ln.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            if t.startswith('.LBB') and t.endswith(':') and '--labels' in sys.argv:
                segs.append(cur); cur = dict.fromkeys(kinds, 0); names.append(t[:-1])
            continue
        op = t.split()[0]
        if op.endswith(':'):
            if '--labels' in sys.argv:
                segs.append(cur); cur = dict.fromkeys(kinds, 0); names.append(op[:-1])
            continue
        if op == 's_barrier':
            segs.append(cur); cur = dict.fromkeys(kinds, 0); names.append("barrier"); continue
        k = ('valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_')
             else 'vmem' if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch') else 'other')
        cur[k] += 1
    segs.append(cur)
    print(m.group(1)[:100])
    for n, c in zip(names, segs):
        print(f"{n:14s}", " ".join(f"{k} {c[k]:5d}" for k in kinds))
    print(f"{'total':14s}", " ".join(f"{k} {sum(c[k] for c in segs):5d}" for k in kinds))
    for key in ("num_vgpr", "numbered_sgpr", "private_seg_size"):
        mm = re.search(re.escape(m.group(1)) + r"\." + key + r", (\d+)", s)
        if mm:
            print(key, mm.group(1))


if __name__ == "__main__":
    main()
