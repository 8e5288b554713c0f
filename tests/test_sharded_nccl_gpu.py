"""The torch.distributed drivers on the real backend (nccl = RCCL), world_size 1: every collective and every
device-resident code path of richdem_amd/sharded.py runs, and the results must equal the single-GPU entry points.
(World sizes 2 and 3 run on CPU ranks with gloo around model engines: tests/test_sharded_dist.py; the shard
engines themselves are checked for tiling invariance on one GPU in the *_gpu tests.)"""
import os
import socket

import numpy as np
import pytest

from richdem_amd.synth import fractal_dem, fractal_dem_int

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_DEBUG="WARN")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_sharded_drivers_on_rccl(rd, orc, nccl_world1):
    import torch

    from richdem_amd.sharded import (d8_flow_accum_sharded, d8_flow_directions_sharded, fill_depressions_sharded,
                                     flat_resolution_sharded)

    z = fractal_dem(700, 500, 71)
    exp = orc.port.fill(z)
    blk = torch.from_numpy(z.copy()).cuda()
    fill_depressions_sharded(blk)
    assert np.array_equal(blk.cpu().numpy(), exp)
    blk64 = torch.from_numpy(z.astype(np.float64)).cuda()                  # float64 that fits float32
    fill_depressions_sharded(blk64, topology="D4")
    assert np.array_equal(blk64.cpu().numpy(), orc.port.fill(z, 4).astype(np.float64))
    bad = blk64 + 1e-9
    with pytest.raises(rd.RdgpuError):
        fill_depressions_sharded(bad)
    zi = torch.from_numpy(orc.port.fill(fractal_dem_int(600, 400, 72, 0.05))).cuda()
    dirs = d8_flow_directions_sharded(zi, -9999, flats=True)
    exp_dirs = orc.port.flat_resolution(zi.cpu().numpy(), np.int32(-9999))
    assert np.array_equal(dirs.cpu().numpy(), exp_dirs)
    assert np.array_equal(flat_resolution_sharded(zi, -9999).cpu().numpy(), exp_dirs)
    area = torch.empty(dirs.shape, dtype=torch.float64, device="cuda")
    d8_flow_accum_sharded(dirs.contiguous(), area)
    assert np.array_equal(area.cpu().numpy(), orc.port.d8_flow_accum(exp_dirs, 255, np.float64))


def test_bench_line_of_the_sharded_path():
    """`bench.py --gpus N` for N > 1 goes through richdem_amd.sharded.bench_sharded; RDGPU_BENCH_FORCE_SHARDED=1 runs that
    path on one rank over RCCL, launched the way the driver launches it.  The JSON line must carry the contract's fields,
    the stages of BASELINE configs[4] and `exchanges == 1` for the accumulation -- and nothing else on stdout."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RDGPU_BENCH_FORCE_SHARDED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--size", "3000"], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Mcells/s" and d["value"] > 0
    assert "row-block" in d["config"]["parallelism"]
    st = d["stages"]
    assert st["d8_flow_accum"]["exchanges"] == 1 and st["directions_plus_flat_resolution"]["ms"] > 0
    assert d["roofline"]["bound"] == "hbm"
