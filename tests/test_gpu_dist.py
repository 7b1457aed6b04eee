"""The N>1 path of bench.py on the one GPU a test box has: two ranks share cuda:0 and talk over gloo (RCCL
refuses two ranks on one device), so everything but the RCCL transport itself runs exactly as it does on an
8-GPU node — rank 0 builds + flattens + bvhgpu_scene_export, ONE broadcast, the peer bvhgpu_scene_imports and
traverses its own shard of the seed-0 ray stream."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("scene_dist", ["bcast", "replicate", "auto"])
def test_bench_two_ranks_one_gpu(scene_dist):
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    cubes, R = 2000, 60_000
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--cubes", str(cubes), "--rays", str(R),
           "--backend", "gloo", "--one-device", "--scene-dist", scene_dist]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]               # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["rays_per_gpu"] == R and out["value"] > 0
    assert "roofline" in out and "cpu_baseline" not in out  # the CPU leg is rank 0 at N=1 only
    if scene_dist == "auto":
        assert out["config"]["scene_dist"] in ("bcast", "replicate")
        assert set(out["scene_dist_probe_ms_per_step"]) == {"bcast", "replicate"}
    else:
        assert out["config"]["scene_dist"] == scene_dist

    # the two shards together == one process over the first 2R rays of the stream (oracle as the checker)
    from bvh_amd import testbase as tb
    from oracle import orc
    _, aabbs = tb.create_n_cubes(cubes)
    flat = orc.flatten(orc.build(aabbs).nodes)
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, orc.create_rays(0, 2 * R))
    assert out["hits_all_ranks"] == len(idx)
