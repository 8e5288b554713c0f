#!/usr/bin/env python3
"""bench.py -- Mcells/s of the Priority-Flood-equivalent fill on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--no-stages] [--no-host]

A step = one fill (rdgpu_fill_dev_f32 through the C-ABI) of one synthetic float32 DEM, G(seed) of
SURVEY.md section 8d, already resident in HBM when the timed region starts.  At N=1 the workload is
BASELINE configs[2] "40000x40000 float32 DEM, Priority-Flood (+ flat resolution) on 1 MI355X".  One JSON
line on rank 0.  Besides `value` (the fill) the line carries

  roofline       whole fill first (8 algorithmic bytes per cell ONCE for all of its kernels), then the
                 dominant raster kernel of the fill with its own launch time (HIP events on the launch stream,
                 taken in a separate instrumented pass -- the timed steps run without event recording)
  stages         the rest of the path on the filled DEM (BASELINE configs[2] and [4]): D8 directions,
                 directions + flat resolution, d8_flow_accum, ResolveFlatsEpsilon, FA_D8 on the epsilon-resolved
                 DEM -- each with ms, Mcells/s, SURVEY 8d algorithmic bytes and the fraction of 8 TB/s
  end_to_end_host  rdgpu_fill_f32 on a host array (the drop-in boundary: H2D + fill + D2H), never `value`
  cpu_baseline   the compiled reference on this box's host: on the WHOLE bench DEM (a child process started after the timed
                 fills, running beside the stages: `--cpu-full`), with the bounded-window figure as the secondary key
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
#: SURVEY.md section 8d, algorithmic bytes per cell
STAGE_BYTES = {"d8_flow_directions": 5, "directions_plus_flat_resolution": 6, "d8_flow_accum": 9,
               "resolve_flats_epsilon": 8, "fa_d8": 20, "priority_flood_epsilon": 8, "priority_flood_flowdirs": 5,
               "dinf_flow_directions": 8, "fa_tarboton": 20}   # D-infinity: f32 in + f32 angle out; f32 dem + f64 weights in + f64 out


def cpu_baseline(Z, sample: int, full=None):
    """Times the reference's FillDepressions<D8> (or the C port) on this box's host cores (1 thread: the reference fill
    has no OpenMP).  Reported baseline only.  `full` (start_full_reference's handle): the compiled reference on the WHOLE
    bench DEM, started in a child process while the GPU stages ran -- when it delivers, THAT is `value` (the metric's own
    configuration) and the bounded window becomes the secondary key `window`."""
    import numpy as np

    import oracle  # checker, used here only for the reported CPU baseline

    s = min(sample, Z.shape[0], Z.shape[1])
    win = Z[:s, :s].cpu().numpy().copy()
    if oracle.ref.available:
        be, kind, what = oracle.ref, "reference", "PriorityFlood_Zhou2016 (unmodified reference headers, oracle/_ref)"
    else:
        if not oracle.port.available:
            oracle.build()
        be, kind, what = oracle.port, "port", "oracle/oracle.c Priority-Flood (Barnes2014 improved)"
    t0 = time.perf_counter()
    out = be.fill(win, 8)
    dt = time.perf_counter() - t0
    assert out.shape == win.shape and np.isfinite(out).all()
    res = {
        "value": round(s * s / 1e6 / dt, 3),
        "unit": "Mcells/s",
        "cores": 1,
        "kind": kind,
        "sample": f"{s}x{s} top-left window of the bench DEM, {what}, {dt:.2f} s",
    }
    got = finish_full_reference(full) if full is not None else None
    if got is not None and "seconds" in got:
        n = got["size"]
        res["window"] = {"value": res["value"], "sample": res["sample"],
                         "note": "ran beside the full-size child process (one more busy core)"}
        res["value"] = round(n * n / 1e6 / got["seconds"], 3)
        res["kind"] = got["kind"]
        res["sample"] = (f"{n}x{n}: the WHOLE bench DEM (the metric's own configuration), {got['what']}, {got['seconds']:.1f} s of one "
                         f"host core, measured in this run in a child process beside the GPU stages")
        res["full_size"] = {k: got[k] for k in ("seconds", "cells_raised", "output_equals_gpu_fill", "bands_compared") if k in got}
        return res
    # The whole 40000 x 40000 DEM through the reference was not measured in this run (switched off, or the child failed /
    # ran out of time): quoted from earlier builder runs and labelled as such.
    full_q = {"note": "quoted, not measured in this run: the compiled reference on the full 40000 x 40000 bench DEM"}
    if got is not None:
        full_q["child"] = got
    try:
        with open(os.path.join(ROOT, "profiles", "r02_parity40k.json")) as f:
            p = json.load(f)
        full_q["gpu_box_host_r02"] = {k: p[k] for k in ("ref_fill_s", "ref_fill_Mcells_s", "ref_flat_resolution_s",
                                                         "ref_d8_flow_accum_s") if k in p}
        full_q["gpu_box_host_r02"]["source"] = "profiles/r02_parity40k.json (tests/tools/parity40k.py on a GPU box's host, round 2)"
    except (OSError, ValueError):
        pass
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "ref_s3_digests.npz"))
        full_q["build_container_r03"] = {k.split("/")[1] + "_s": float(g[k]) for k in g.files if k.startswith("ref_seconds/")}
        full_q["build_container_r03"]["source"] = ("tests/golden/ref_s3_digests.npz (make_golden.py --s3-digests: the run that produced the "
                                                   "full-size parity digests; 8-core build container, a slower host)")
    except (OSError, ValueError):
        pass
    res["full_size"] = full_q
    return res


# ---- the reference on the WHOLE bench DEM, in a child process beside the GPU stages (cpu_baseline leg) ---------------------
def start_full_reference(Z, W, timeout_s: float):
    """Writes the (unfilled) bench DEM to a scratch file and starts `bench.py --cpu-full-child`, which runs the compiled
    reference's PriorityFlood_Zhou2016 on all of it (minutes of ONE host core, ~13 GB of host memory) and leaves its time
    and the band digests of its output.  Called AFTER the timed fills; the GPU stages run meanwhile.  Returns a handle for
    finish_full_reference, or None when the compiled reference is not on this box."""
    import subprocess
    import tempfile

    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from digest import band_digests_torch   # (test infrastructure: the same digests the S3 parity tests use)

    n = Z.shape[0]
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > 2 * Z.numel() * 4 \
        else tempfile.gettempdir()
    raw = os.path.join(base, f"rdgpu_bench_dem_{os.getpid()}.f32")
    res = raw + ".json"
    Z.cpu().numpy().tofile(raw)
    gpu_digests = band_digests_torch(W)              # the GPU fill's output, for the child's answer to be checked against
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-full-child", raw, str(n), res],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return {"proc": child, "raw": raw, "res": res, "size": n, "gpu_digests": gpu_digests, "deadline": time.time() + timeout_s}


def finish_full_reference(h):
    """Waits for the child (up to the handle's deadline), removes the scratch files, compares the digests."""
    import subprocess

    try:
        try:
            h["proc"].wait(timeout=max(1.0, h["deadline"] - time.time()))
        except subprocess.TimeoutExpired:
            h["proc"].kill()                           # (the exact process started above)
            h["proc"].wait()
            return {"error": "the full-size reference did not finish inside --cpu-full-timeout"}
        if h["proc"].returncode != 0 or not os.path.exists(h["res"]):
            return {"error": f"the full-size reference child exited with {h['proc'].returncode}"}
        with open(h["res"]) as f:
            got = json.load(f)
        import numpy as np

        ref_d = np.array(got.pop("digests"), dtype=np.uint64)
        got["bands_compared"] = int(ref_d.size)
        got["output_equals_gpu_fill"] = bool(ref_d.size == h["gpu_digests"].size and (ref_d == h["gpu_digests"]).all())
        got["size"] = h["size"]
        return got
    finally:
        for p in (h["raw"], h["res"]):
            try:
                os.remove(p)
            except OSError:
                pass


def cpu_full_child(raw: str, n: int, res: str) -> None:
    """`bench.py --cpu-full-child <raw f32 file> <n> <result json>`: the child of start_full_reference."""
    import numpy as np

    import oracle  # checker: the reported CPU baseline

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from digest import band_digests_np

    dem = np.fromfile(raw, dtype=np.float32).reshape(n, n)
    t0 = time.perf_counter()
    out = oracle.ref.fill(dem, 8)
    dt = time.perf_counter() - t0
    got = {"seconds": dt, "kind": "reference", "what": "PriorityFlood_Zhou2016 (unmodified reference headers, oracle/_ref)",
           "cells_raised": int((out != dem).sum()), "digests": [int(d) for d in band_digests_np(out)]}
    with open(res + ".tmp", "w") as f:
        json.dump(got, f)
    os.replace(res + ".tmp", res)


def _best(fn, reps, sync):
    best = 1e30
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        best = min(best, time.perf_counter() - t0)
    return best


def stage_entry(seconds: float, cells: int, bytes_per_cell: int, n_gpus: int = 1) -> dict:
    gbs = cells * bytes_per_cell / seconds / 1e9
    return {"ms": round(seconds * 1e3, 3), "Mcells_s": round(cells / 1e6 / seconds, 1), "alg_bytes_per_cell": bytes_per_cell,
            "alg_GBps": round(gbs, 1), "frac_of_peak": round(gbs / (HBM_PEAK_GBS * n_gpus), 4)}


def run_stages(rd, torch, W, nodata: float, reps: int = 2, Z=None, pf_flowdirs: bool = True, draining_mfd: bool = True) -> dict:
    """The path after the fill on one GPU, HBM-resident, through the C-ABI `_dev_` entry points.  W = the filled DEM,
    Z = the unfilled one (for PriorityFloodEpsilon, the other fill of depressions.hpp)."""
    n_cells = W.numel()
    sync = torch.cuda.synchronize
    out = {}
    dirs = torch.empty(W.shape, dtype=torch.uint8, device="cuda")
    rd.d8_flow_directions_dev(W, nodata, dirs)                                   # workspace growth: not timed
    out["d8_flow_directions"] = stage_entry(_best(lambda: rd.d8_flow_directions_dev(W, nodata, dirs), reps, sync), n_cells, 5)
    rd.d8_flow_directions_dev(W, nodata, dirs, flats=True)
    out["directions_plus_flat_resolution"] = stage_entry(
        _best(lambda: rd.d8_flow_directions_dev(W, nodata, dirs, flats=True), reps, sync), n_cells, 6)
    fs = rd.flat_stats()
    out["directions_plus_flat_resolution"].update({"noflow_cells": fs["noflow"], "rounds_towards": fs["towards"],
                                                   "rounds_away": fs["away"], "tail_visits": fs["tail_visits"],
                                                   "tail_live_tiles": fs["tail_live_tiles"],
                                                   "engine": "level fields as bit planes per 64 x 64 tile (csrc/flat_planes.inc)",
                                                   "note": "tail_live_tiles = the active lists' lengths when the rounds hand over to the "
                                                           "resident wavefronts, NOT the tiles the tails visit: the fronts move on through "
                                                           "~3e5 tiles -- 2.3 working visits per tile and search over the whole call, 15 % of "
                                                           "them ahead of their order (profiles/r06a_flat_visit_hist.json)"})
    area = torch.empty(W.shape, dtype=torch.float64, device="cuda")
    rd.d8_flow_accum_dev(dirs, area)
    out["d8_flow_accum"] = stage_entry(_best(lambda: rd.d8_flow_accum_dev(dirs, area), reps, sync), n_cells, 9)
    out["d8_flow_accum"]["input"] = "flat-resolved directions of the filled DEM; float64 out"
    out["d8_flow_accum"]["max_area"] = float(area.max().item())
    del dirs
    E = W.clone()
    rd.resolve_flats_epsilon_dev(E, nodata)
    t_rfe = 1e30
    for _ in range(reps):
        E.copy_(W)
        t_rfe = min(t_rfe, _best(lambda: rd.resolve_flats_epsilon_dev(E, nodata), 1, sync))
    out["resolve_flats_epsilon"] = stage_entry(t_rfe, n_cells, 8)
    area.fill_(1.0)
    rd.fa_d8_dev(E, nodata, area)
    t_fa = 1e30
    for _ in range(reps):
        area.fill_(1.0)
        t_fa = min(t_fa, _best(lambda: rd.fa_d8_dev(E, nodata, area), 1, sync))
    out["fa_d8"] = stage_entry(t_fa, n_cells, 20)
    out["fa_d8"]["input"] = "fill -> ResolveFlatsEpsilon output, an array of ones as weights (the engine reads it once to find out)"
    out["fa_d8"]["max_accum"] = float(area.max().item())
    # the same when the CALLER says the weights are ones (rdgpu_fa_d8_unit_dev_<T>: what rd.FlowAccumulation(weights=None) and
    # rd_flow_accumulation's accum(dem, 1) amount to): 12 algorithmic bytes per cell -- the DEM in, the accumulation out
    t_fu = _best(lambda: rd.fa_d8_dev(E, nodata, area, unit_weights=True), reps, sync)
    out["fa_d8"]["unit_weights_declared"] = {**stage_entry(t_fu, n_cells, 12), "max_accum": float(area.max().item())}
    # D-infinity (north_star names it beside D8): the angle raster (dinf_flow_directions) and FA_Tarboton on the epsilon-resolved
    # DEM, where every cell drains
    ang = torch.empty(W.shape, dtype=torch.float32, device="cuda")
    rd.dinf_flow_directions_dev(E, nodata, ang)
    out["dinf_flow_directions"] = stage_entry(_best(lambda: rd.dinf_flow_directions_dev(E, nodata, ang), reps, sync), n_cells,
                                              STAGE_BYTES["dinf_flow_directions"])
    del ang
    # FA_Tarboton on the FILLED DEM (its flats stop the flow, as in tools/mfd_bench.py and rd_flow_accumulation after
    # rd_depressions_flood), and once on the epsilon-resolved DEM where every cell drains: the engine works through lists of
    # completed cells, one launch per step of the longest flow path, and that path is ~1e5 cells long there
    area.fill_(1.0)
    rd.fa_tarboton_dev(W, nodata, area)
    t_ft = 1e30
    for _ in range(reps):
        area.fill_(1.0)
        t_ft = min(t_ft, _best(lambda: rd.fa_tarboton_dev(W, nodata, area), 1, sync))
    out["fa_tarboton"] = stage_entry(t_ft, n_cells, STAGE_BYTES["fa_tarboton"])
    out["fa_tarboton"]["input"] = "the filled DEM, unit weights"
    out["fa_tarboton"]["max_accum"] = float(area.max().item())
    if draining_mfd:
        area.fill_(1.0)
        t_fd = _best(lambda: rd.fa_tarboton_dev(E, nodata, area), 1, sync)
        out["fa_tarboton"]["on_the_draining_dem"] = {**stage_entry(t_fd, n_cells, STAGE_BYTES["fa_tarboton"]),
                                                     "input": "fill -> ResolveFlatsEpsilon output (every cell drains), unit weights, run once",
                                                     "max_accum": float(area.max().item())}
    del area
    if Z is not None:
        sync()
        rd.release_workspace()            # the stages above keep ~80 GB of scratch; the two below bring their own
        torch.cuda.empty_cache()
        rd.fill_epsilon_dev(E.copy_(Z), nodata)
        t_eps = 1e30
        for _ in range(reps):
            E.copy_(Z)
            t_eps = min(t_eps, _best(lambda: rd.fill_epsilon_dev(E, nodata), 1, sync))
        out["priority_flood_epsilon"] = stage_entry(t_eps, n_cells, STAGE_BYTES["priority_flood_epsilon"])
        es = rd.epsilon_stats()
        out["priority_flood_epsilon"].update({"input": "the unfilled bench DEM (PriorityFloodEpsilon_Original semantics)",
                                              **{k: es[k] for k in ("rounds", "tie_sources") if k in es}})
        out["priority_flood_epsilon"]["cells_differing_from_reference"] = _differing(torch, E, "epsilon")
        if not pf_flowdirs:
            return out
        # PriorityFloodFlowdirs_Barnes2014: one fill per nesting level of the depressions (seconds, not milliseconds: once)
        del E
        sync()
        rd.release_workspace()
        torch.cuda.empty_cache()
        pdirs = torch.empty(W.shape, dtype=torch.uint8, device="cuda")
        prev = os.environ.get("RDGPU_PFD_TIE_PASSES")                              # (a value the user exported is put back)
        try:
            os.environ["RDGPU_PFD_TIE_PASSES"] = "1"                               # workspace growth (~100 GB of sort and tree buffers):
            rd.pf_flowdirs_dev(Z, nodata, pdirs)                                   # one pass of the tie order, not timed
        finally:
            if prev is None:
                os.environ.pop("RDGPU_PFD_TIE_PASSES", None)
            else:
                os.environ["RDGPU_PFD_TIE_PASSES"] = prev
        t_pf = _best(lambda: rd.pf_flowdirs_dev(Z, nodata, pdirs), 1, sync)
        out["priority_flood_flowdirs"] = stage_entry(t_pf, n_cells, STAGE_BYTES["priority_flood_flowdirs"])
        ps = rd.pf_flowdirs_stats()
        out["priority_flood_flowdirs"].update({"input": "the unfilled bench DEM", "levels": ps["levels"],
                                               "cells_with_an_equal_elevation_twin": ps["twins"],
                                               "tie_order_passes": ps["tie_passes"], "timed_runs": 1,
                                               "cells_with_an_unsettled_tie_order": ps["unresolved"]})
        out["priority_flood_flowdirs"]["cells_differing_from_reference"] = _differing(torch, pdirs, "flowdirs")
    return out


def _differing(torch, out, which: str):
    """The stage's output against the COMPILED REFERENCE's values at a fixed sample of cells of the 40000 x 40000 bench DEM
    (tests/golden/ref_s3_f2_<which>.npz, make_golden.py --s3-f2; the same comparison as tests/test_s3_f2_gpu.py).  The
    reference's result with equal elevations follows its queue's order: this says how far the stage's output is from it."""
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", f"ref_s3_f2_{which}.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if int(g["size"]) != out.shape[0] or out.shape[0] != out.shape[1]:
        return None
    k, stride = int(g["sample_k"]), int(g["sample_stride"])
    pos = (torch.arange(k, dtype=torch.int64, device=out.device) * stride) % out.numel()
    nd = int((out.reshape(-1)[pos] != torch.from_numpy(g["sample"]).to(out.device)).sum().item())
    return {"sample_cells": k, "sample_differing": nd, "estimated_fraction": nd / k, "estimated_cells": int(round(nd / k * out.numel())),
            "source": f"tests/golden/ref_s3_f2_{which}.npz (compiled reference, {float(g['ref_seconds']):.0f} s of one core)"}


def host_path(rd, torch, Z, reps: int = 2) -> dict:
    """The drop-in boundary as rd_depressions_flood sees it: rdgpu_fill_f32 on a host array (H2D + fill + D2H)."""
    host = Z.cpu().numpy()
    best = 1e30
    for _ in range(reps):
        a = host.copy()
        t0 = time.perf_counter()
        rd.FillDepressions(a, in_place=True)
        best = min(best, time.perf_counter() - t0)
    out = {"entry": "rdgpu_fill_f32 (host pointer: H2D + fill + D2H into the same buffer, pageable numpy array)",
           "ms": round(best * 1e3, 1), "Mcells_s": round(Z.numel() / 1e6 / best, 1),
           "GB_over_pcie": round(2 * host.nbytes / 1e9, 2)}
    # the link's ceiling, measured here with a PINNED 1 GiB buffer (what hipHostRegister of the caller's array would buy --
    # at 120 - 200 ms per call for 6.4 GB, tools/probes/pcie_probe.hip / profiles/r06d_pcie_probe.txt, more than it saves)
    try:
        pin = torch.empty(1 << 28, dtype=torch.float32).pin_memory()
        dev = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        rates = {}
        for name, dst, src in (("h2d", dev, pin), ("d2h", pin, dev)):
            bt = 1e30
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                bt = min(bt, time.perf_counter() - t0)
            rates[name] = pin.numel() * 4 / bt / 1e9
        copies = host.nbytes / 1e9 / rates["h2d"] + host.nbytes / 1e9 / rates["d2h"]
        out["pcie_pinned_GBps"] = {k: round(v, 1) for k, v in rates.items()}
        out["pcie_floor_ms"] = round(copies * 1e3, 1)     # both copies at the pinned rate, nothing else
        out["note"] = ("ms - pcie_floor_ms = the fill plus what pageable staging costs; registering the caller's buffer for the "
                       "call costs more than the pinned rate saves (profiles/r06d_pcie_probe.txt)")
        del pin, dev
    except Exception as e:   # (no pinned memory on this box: the plain figure stands)
        out["pcie_pinned_GBps"] = f"not measured: {e}"
    return out


def main():
    if len(sys.argv) == 5 and sys.argv[1] == "--cpu-full-child":
        return cpu_full_child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=40000, help="DEM is size x size cells")
    ap.add_argument("--cpu-sample", type=int, default=14000, help="window edge for the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-full", type=int, default=-1,
                    help="1: also run the compiled reference on the WHOLE DEM in a child process beside the GPU stages and report it "
                         "as cpu_baseline.value (minutes of one host core); 0: window only; default: on at the metric's size 40000")
    ap.add_argument("--cpu-full-timeout", type=float, default=900.0, help="seconds the child gets before it is given up")
    ap.add_argument("--no-stages", action="store_true", help="skip the directions / flat resolution / accumulation stages")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer end-to-end measurement")
    ap.add_argument("--no-draining-mfd", action="store_true",
                    help="skip FA_Tarboton's run on the epsilon-resolved DEM (7 s: thousands of rounds; the profiling passes)")
    ap.add_argument("--no-pf-flowdirs", action="store_true",
                    help="skip the PriorityFloodFlowdirs stage (324 more fills: the profiling passes' per-launch averages are the headline fill's)")
    args = ap.parse_args()

    # stdout must carry exactly ONE line, the JSON of rank 0.  Libraries print there too (RCCL: "Librccl path : ...",
    # flushed from the C stdio buffer at exit, on every rank), so file descriptor 1 is pointed at stderr for the
    # whole run and the JSON line is written to the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch

    import richdem_amd as rd
    from richdem_amd.roofline import fill_roofline

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("RDGPU_BENCH_FORCE_SHARDED") == "1":   # the env switch runs the N>1 code path on 1 rank
        import torch.distributed as dist

        # RCCL logs to stdout by default; route them to a file so stdout carries exactly one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rdgpu_rccl_%h_%p.log")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from richdem_amd.sharded import bench_sharded

        args.cpu_baseline_fn = cpu_baseline   # (the oracle stays out of the product package: bench.py hands the leg over)
        out = bench_sharded(args, rank, world)
        if out is not None:   # rank 0
            emit(out)
        return

    n = args.size
    cells = n * n
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=args.seed)
    bufs = [Z.clone() for _ in range(args.steps)]   # fill is in place: one pristine copy per timed step
    scratch = Z.clone()
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        scratch.copy_(Z)
        rd.fill_depressions_dev(scratch)
    torch.cuda.synchronize()

    # the timed region: K fills, nothing else (no event recording)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        rd.fill_depressions_dev(bufs[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = rd.fill_stats()
    changed = float((bufs[0] != Z).float().mean())
    W = bufs[0]                        # the filled DEM: input of the stages below
    del bufs[1:]

    # per-kernel times: a separate, instrumented pass (HIP events around every launch, on the launch stream)
    prof_steps = 2
    rd.profile_reset()
    rd.profile_enable(True)
    for _ in range(prof_steps):
        scratch.copy_(Z)
        rd.fill_depressions_dev(scratch)
    torch.cuda.synchronize()
    rd.profile_enable(False)
    prof = rd.profile_totals()
    del scratch

    full = None
    if args.cpu_sample > 0 and (args.cpu_full == 1 or (args.cpu_full < 0 and n == 40000)):
        full = start_full_reference(Z, W, args.cpu_full_timeout)   # (after every timed fill; the stages below run beside it)

    ms_step = dt * 1e3 / args.steps
    value = cells / 1e6 / (dt / args.steps)
    roofline = fill_roofline(prof, stats, cells, prof_steps, dt / args.steps,
                             os.path.join(ROOT, "profiles", "pmc_traffic.json"), n)
    out = {
        "metric": "Mcells/s Priority-Flood fill, 40k x 40k f32 DEM",
        "value": round(value, 2),
        "unit": "Mcells/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{n}x{n} float32 fractal value-noise DEM G(seed={args.seed}), FillDepressions<D8>, HBM-resident",
            "cells": cells,
            "basins": stats["basins"],
            "boruvka_rounds": stats["rounds"],
            "pair_records": stats["edge_records"],
            "jump_passes": stats["jump_passes"],
            "host_syncs_per_fill": stats.get("host_syncs"),
            "cells_raised_frac": round(changed, 4),
            "parallelism": "1 GPU",
        },
        "roofline": roofline,
        "kernels_ms_per_step": {k: round(v[0] / prof_steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
    }
    if not args.no_stages:
        out["stages"] = {"fill": stage_entry(dt / args.steps, cells, 8)}
        out["stages"].update(run_stages(rd, torch, W, -9999.0, Z=Z, pf_flowdirs=not args.no_pf_flowdirs,
                                        draining_mfd=not args.no_draining_mfd))
    if not args.no_host:
        del W
        bufs.clear()
        rd.release_workspace()
        torch.cuda.empty_cache()
        out["end_to_end_host"] = host_path(rd, torch, Z)
    if args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(Z, args.cpu_sample, full)
    emit(out)


if __name__ == "__main__":
    main()
