#!/usr/bin/env python3
"""Fill and directions + flat resolution at --size on a QUIET host and with every host core busy (one spinning process per
core, started and killed by PID here): how much of a stage's time is the host's (r05u saw the fill go from
16.2 to 18.2 ms and the flats from 31 to 41.8 ms on a busy host).  Median and best of --reps, one JSON object."""
import argparse, json, os, statistics, subprocess, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--hogs", type=str, default="", help="comma-separated numbers of spinning processes (default: half the cores, all but two, all)")
    a = ap.parse_args()
    import torch

    import richdem_amd as rd

    n = a.size
    Z = torch.empty((n, n), dtype=torch.float32, device="cuda")
    rd.synth_dem_dev(Z, seed=3)
    W = Z.clone()
    rd.fill_depressions_dev(W)
    scratch = Z.clone()
    dirs = torch.empty((n, n), dtype=torch.uint8, device="cuda")

    def timed(fn, prep=None):
        ts = []
        for _ in range(a.reps):
            if prep:
                prep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return {"median_ms": round(statistics.median(ts), 3), "best_ms": round(min(ts), 3), "worst_ms": round(max(ts), 3)}

    def measure():
        out = {"fill": timed(lambda: rd.fill_depressions_dev(scratch), lambda: scratch.copy_(Z)),
               "directions_plus_flat_resolution": timed(lambda: rd.d8_flow_directions_dev(W, -9999.0, dirs, flats=True))}
        out["fill"]["host_syncs"] = rd.fill_stats()["host_syncs"]
        return out

    measure()   # warm-up: workspaces
    cores = os.cpu_count() or 1
    res = {"size": n, "reps": a.reps, "host_cores": cores, "quiet": measure()}
    loads = [int(x) for x in a.hogs.split(",")] if a.hogs else [cores // 2, max(1, cores - 2), cores]
    for nh in loads:
        hogs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(nh)]
        try:
            time.sleep(1.0)
            m = measure()
        finally:
            for p in hogs:
                p.kill()
            for p in hogs:
                p.wait()
        for k in ("fill", "directions_plus_flat_resolution"):
            m[k]["over_quiet_median"] = round(m[k]["median_ms"] / res["quiet"][k]["median_ms"], 3)
        res[f"busy_{nh}_of_{cores}_cores"] = m
    print(json.dumps(res))


if __name__ == "__main__":
    main()
