#!/usr/bin/env python3
"""Produces the per-round measurement artifacts for profiles/ on the GPU box:

    python tools/profile_round.py <tag> [--steps K] [--sq]

runs, back to back and all on the same command (`python bench.py --steps K --warmup 1`):
  1. the plain bench               -> gpurun_out/<tag>_fill40k_bench.json
  2. rocprofv3 --kernel-trace --stats  -> gpurun_out/<tag>_fill40k_kernel_stats.csv
  3. rocprofv3 --pmc FETCH_SIZE  and  4. rocprofv3 --pmc WRITE_SIZE (separate passes, kernel-trace only)
                                   -> gpurun_out/<tag>_fill40k_pmc_summary.csv, gpurun_out/pmc_traffic.json
  5. (--sq) rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
     SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS (one more pass, kernel-trace only)
                                   -> gpurun_out/<tag>_fill40k_sq_summary.csv: where the waves of each kernel spend
                                      their cycles (parked / issuing / VALU / LDS), as fractions of SQ_WAVE_CYCLES
Counters are KiB; WRITE_SIZE x1 and FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md; calibrated in r01a on kernels of
known byte count: k_synth writes exactly 4 B/cell, k_count_pits reads exactly 4 B/cell).  Copy the four files into
profiles/ and commit them."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT).returncode


def counters(directory, name):
    per = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != name:
                    continue
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                per[k][0] += 1
                per[k][1] += float(row["Counter_Value"])
    return per


def kernel_shas():
    """sha1 (first 12 hex digits) of every kernel source: the evidence names the engine state it was taken on"""
    import hashlib
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "richdem_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "richdem_amd", "csrc", "*.inc"))):
        with open(f, "rb") as fh:
            out.append(os.path.basename(f) + ":" + hashlib.sha1(fh.read()).hexdigest()[:12])
    return " ".join(out)


def git_sha():
    """the commit the tree was at when it was sent to the GPU box (tools/gpurun.sh writes .gitsha; the box has no .git)"""
    if os.environ.get("RDGPU_GIT_SHA"):
        return os.environ["RDGPU_GIT_SHA"]
    try:
        with open(os.path.join(ROOT, ".gitsha")) as f:
            return f.read().strip()
    except OSError:
        return "unknown"


def main():
    tag = sys.argv[1]
    steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "3"
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "1"]
    # the profiling passes run the fill only, or (--path) the fill and the stages after it; never the host path
    prof_tail = (["--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-host", "--no-pf-flowdirs", "--no-draining-mfd"]
                 + ([] if "--path" in sys.argv else ["--no-stages"]))
    what = "path40k" if "--path" in sys.argv else "fill40k"
    os.makedirs(OUT, exist_ok=True)
    log = os.path.join(OUT, f"{tag}_bench.log")
    run(bench, log)
    line = [l for l in open(log) if l.startswith("{")][-1]
    with open(os.path.join(OUT, f"{tag}_fill40k_bench.json"), "w") as f:
        f.write(line)
    for sub, extra in (("stats", ["--kernel-trace", "--stats"]), ("fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"]),
                       ("write", ["--kernel-trace", "--pmc", "WRITE_SIZE"])):
        d = os.path.join(OUT, f"{tag}_{sub}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3"] + extra + ["--output-format", "csv", "-d", d, "-o", "run", "--"] + bench[:2] + prof_tail
        rc = run(cmd, os.path.join(OUT, f"{tag}_{sub}.log"))
        print(sub, "rc", rc)
    ks = glob.glob(os.path.join(OUT, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(OUT, f"{tag}_{what}_kernel_stats.csv"))
    fetch = counters(os.path.join(OUT, f"{tag}_fetch"), "FETCH_SIZE")
    write = counters(os.path.join(OUT, f"{tag}_write"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch[k][0], write[k][0])
        if not k.startswith("rdgpu::") or n == 0:
            continue
        fg = fetch[k][1] * 1024 * 2 / 1e9 / n
        wg = write[k][1] * 1024 / 1e9 / n
        rows.append((k, n, fg, wg, fg + wg))
    with open(os.path.join(OUT, f"{tag}_{what}_pmc_summary.csv"), "w") as f:
        f.write(f"# {tag} PMC summary: fill, 40000x40000 f32, 1 step (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)\n")
        f.write(f"# git {git_sha()}  kernel sources sha1: {kernel_shas()}\n")
        f.write("# counters are KiB; WRITE_SIZE x1, FETCH_SIZE x2 (gfx950 half-count; calibration: k_synth writes 6.4e9 B, k_count_pits reads 6.4e9 B)\n")
        f.write("kernel,launches,fetch_GB_per_launch(x2),write_GB_per_launch,total_GB_per_launch\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},{r[2]:.3f},{r[3]:.3f},{r[4]:.3f}\n")
    # per-launch HBM traffic of the fill's kernels, under the names the library's profiler (and bench.py) uses
    names = {"k_scan": "fill.scan", "k_pairs16": "fill.scan", "k_descent": "fill.descent", "k_descent16": "fill.descent", "k_tile_label": "fill.tile_label",
             "k_finalize": "fill.finalize", "k_finalize_tiled": "fill.finalize", "k_finalize16": "fill.finalize",
             "k_edge_round": "fill.edge_round", "k_resolve_nodes": "fill.resolve_nodes"}
    per = {}
    for kern, prof_name in names.items():
        sel = [r for r in rows if r[0].split("<")[0] == "rdgpu::" + kern]
        if sel:
            cnt = sum(r[1] for r in sel)
            per[prof_name] = round(sum(r[4] * r[1] for r in sel) / cnt, 3)
    if per:
        with open(os.path.join(OUT, "pmc_traffic.json"), "w") as f:
            sys.path.insert(0, ROOT)
            from richdem_amd.roofline import engine_sha

            fill_kernels = ("k_descent", "k_descent16", "k_resolve_nodes", "k_node_levels", "k_finalize16", "k_tile_label", "k_scan",
                            "k_edge_round", "k_finalize", "k_finalize_tiled", "k_hook",
                            "k_chase_links", "k_update_basins", "k_compact_roots", "k_best_reset", "k_init_tables",
                            "k_compact_alive", "k_sum_segments", "k_chase", "k_label_cells", "k_pairs16", "k_stripe_offsets")
            # launches counted by the profiling command's fills (1 timed + the instrumented pass of bench.py)
            nfill = max(1, sum(r[1] for r in rows if r[0].split("<")[0] in ("rdgpu::k_descent", "rdgpu::k_descent16")))
            per_fill = sum(r[4] * r[1] for r in rows if r[0].split("<")[0].replace("rdgpu::", "") in fill_kernels) / nfill
            json.dump({"size": 40000, "GB_per_launch": per, "GB_per_fill": round(per_fill, 2), "engine_sha": engine_sha(),
                       "git_sha": git_sha(),
                       "source": f"profiles/{tag}_{what}_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 per MI355X_MICROARCH.md)"}, f)
        # the bench line above was printed before these passes ran: give it this round's traffic figure
        d = json.loads(line)
        if d.get("roofline") and d["roofline"].get("kernel") in per:
            d["roofline"]["traffic"] = per[d["roofline"]["kernel"]]
            d["roofline"]["traffic_unit"] = "GB per launch (rocprofv3 PMC, profiles/)"
        line = json.dumps(d)
        with open(os.path.join(OUT, f"{tag}_fill40k_bench.json"), "w") as f:
            f.write(line + "\n")
    if "--sq" in sys.argv:
        names_sq = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                    "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS"]
        # r05: scalar instructions too (a scalar instruction costs a SIMD ~5 cycles, tools/probes/valu_issue.hip): a second pass,
        # so that the first keeps its eight counters
        names_sq2 = ["SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES"]
        d = os.path.join(OUT, f"{tag}_sq")
        shutil.rmtree(d, ignore_errors=True)
        cmd = (["rocprofv3", "--kernel-trace", "--pmc"] + names_sq + ["--output-format", "csv", "-d", d, "-o", "run", "--"]
               + bench[:2] + prof_tail)
        print("sq rc", run(cmd, os.path.join(OUT, f"{tag}_sq.log")))
        cols = {n: counters(d, n) for n in names_sq}
        d2 = os.path.join(OUT, f"{tag}_sq2")
        shutil.rmtree(d2, ignore_errors=True)
        cmd2 = (["rocprofv3", "--kernel-trace", "--pmc"] + names_sq2 + ["--output-format", "csv", "-d", d2, "-o", "run", "--"]
                + bench[:2] + prof_tail)
        print("sq2 rc", run(cmd2, os.path.join(OUT, f"{tag}_sq2.log")))
        cols2 = {n: counters(d2, n) for n in names_sq2}
        shutil.rmtree(d2, ignore_errors=True)
        with open(os.path.join(OUT, f"{tag}_{what}_sq_summary.csv"), "w") as f:
            f.write(f"# {tag} SQ counters, fill 40000x40000 f32, 1 step (rocprofv3 --kernel-trace --pmc SQ_*; quad-cycle units; "
                    "fractions of SQ_WAVE_CYCLES)\n")
            f.write(f"# git {git_sha()}  kernel sources sha1: {kernel_shas()}\n")
            f.write("kernel,launches,wave_cycles,wait_any,wait_inst_any,active_inst_any,active_valu,active_lds,insts_valu,insts_lds,insts_salu,insts_smem\n")
            for k in sorted(cols["SQ_WAVE_CYCLES"]):
                wc = cols["SQ_WAVE_CYCLES"][k][1]
                if not k.startswith("rdgpu::") or wc <= 0:
                    continue
                frac = [cols[n][k][1] / wc for n in names_sq[1:6]]
                f.write(f"{k},{cols['SQ_WAVE_CYCLES'][k][0]},{wc:.4g}," + ",".join(f"{x:.3f}" for x in frac)
                        + f",{cols['SQ_INSTS_VALU'][k][1]:.4g},{cols['SQ_INSTS_LDS'][k][1]:.4g}"
                        + f",{cols2['SQ_INSTS_SALU'][k][1]:.4g},{cols2['SQ_INSTS_SMEM'][k][1]:.4g}\n")
    print(line.strip())


def _drop_raw(tag):
    """the raw rocprofv3 output (hundreds of MB with the stages' thousands of launches) stays on the box: gpurun merges at
    most 64 MiB back, and the summaries above are what profiles/ keeps"""
    for sub in ("stats", "fetch", "write", "sq"):
        shutil.rmtree(os.path.join(OUT, f"{tag}_{sub}"), ignore_errors=True)


if __name__ == "__main__":
    main()
    _drop_raw(sys.argv[1])
