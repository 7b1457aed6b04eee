"""The N>1 path of bench.py on the one GPU a test box has: two ranks share cuda:0 and talk over gloo (RCCL
refuses two ranks on one device), so the sharding, the plan probe, the timing protocol and the import of a received
scene run as they do on an 8-GPU node; the transport is the torch.distributed fallback (scene blob: rank 0 builds +
flattens + bvhgpu_scene_export, ONE broadcast, the peer bvhgpu_scene_imports).  The RCCL transport of the C ABI
(bvhgpu_comm_* / bvhgpu_bcast*) is exercised with the one rank this box allows in test_rccl_comm_single_rank and from
plain C (tests/c_abi/abi_roundtrip.c)."""
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


def test_rccl_comm_single_rank():
    """bvhgpu_comm_unique_id / comm_init_rank / bcast / bcast_known with nranks = 1 through the Python mirror: RCCL itself
    runs (ncclCommInitRank, ncclBroadcast on the ctx's stream), the tree survives, a wrong announcement is refused."""
    import bvh_amd
    from bvh_amd import Bvh, Context, RayBatch, dist as bdist, testbase as tb
    from bvh_amd._lib import BvhGpuError
    from oracle import orc
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    ctx = Context(0)
    comm = bdist.Communicator(ctx, 1, 0, bdist.Communicator.unique_id())
    _, aabbs = tb.create_n_cubes(300)
    bvh = Bvh.from_aabbs(aabbs, ctx)
    with pytest.raises(BvhGpuError):
        comm.bcast(bvh, 0)                                   # not flattened yet
    bvh.flatten_in_place()
    assert comm.bcast(bvh, 0) is bvh
    assert comm.bcast(bvh, 0, "f32", len(aabbs)) is bvh
    with pytest.raises(BvhGpuError):
        comm.bcast(bvh, 0, "f32", len(aabbs) + 1)            # the root's tree is not what the call announces
    rays = orc.create_rays(0, 20_000)
    off, idx, _, _ = bvh.traverse_batch(RayBatch(len(rays), np.float32, host=rays))
    ooff, oidx, _, _ = orc.traverse_flat(orc.flatten(orc.build(aabbs).nodes), aabbs, rays)
    assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
    comm.close()


@pytest.mark.parametrize("scene_dist", ["bcast", "replicate", "auto", "strong"])
def test_bench_two_ranks_one_gpu(scene_dist):
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    cubes, R = 2000, 60_000
    strong = scene_dist == "strong"
    if strong:
        scene_dist = "replicate"
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--cubes", str(cubes), "--rays", str(R),
           "--backend", "gloo", "--one-device", "--scene-dist", scene_dist, "--no-extra"] + (["--scaling", "strong"] if strong else [])
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]               # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == ("strong" if strong else "weak")
    assert out["config"]["rays_per_gpu"] == (R // 2 if strong else R) and out["value"] > 0
    assert "roofline" in out and "cpu_baseline" not in out  # the CPU leg is rank 0 at N=1 only
    assert out["parity"]["equal"] is True                   # rank 0's shard against the oracle, in-process
    torch_plan = {"bcast": "bcast-torch", "replicate": "replicate"}   # over gloo there is no RCCL: the blob transport stands in
    if scene_dist == "auto":
        assert out["config"]["scene_dist"] in ("bcast-torch", "replicate")
        assert set(out["scene_dist_probe_ms_per_step"]) == {"bcast-torch", "replicate"}
    else:
        assert out["config"]["scene_dist"] == torch_plan[scene_dist]

    # the two shards together == one process over the first 2R rays of the stream (oracle as the checker)
    from bvh_amd import testbase as tb
    from oracle import orc
    _, aabbs = tb.create_n_cubes(cubes)
    flat = orc.flatten(orc.build(aabbs).nodes)
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, orc.create_rays(0, R if strong else 2 * R))
    assert out["hits_all_ranks"] == len(idx)


def test_bench_launches_its_own_ranks():
    """plain `python bench.py --gpus 2 …` with NO launcher in the environment starts two ranks itself (it used to run
    one and print n_gpus = 1).  Two ranks share cuda:0 over gloo here; on a node the same command runs one rank per GPU over RCCL."""
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    cubes, R = 2000, 60_000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cubes", str(cubes),
           "--rays", str(R), "--backend", "gloo", "--one-device", "--no-extra"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch"]["ranks_seen"] == 2 and out["launch"]["self_launched"] is True
    assert out["launch"]["world_size"] == 2 and len(out["launch"]["devices"]) == 2
    assert out["rccl"] is None                    # gloo: no RCCL communicator (RCCL refuses two ranks on one GPU) — and the line says so
    assert out["config"]["rays_total"] == 2 * R and out["parity"]["equal"] is True
    # --gpus 2 under a launcher that made 1 or 3 ranks: refused, nothing printed
    for ws in ("1", "3"):
        bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0"), cwd=ROOT, capture_output=True, text=True, timeout=120)
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_exchange_plan_cannot_take_the_line_down():
    """N > 1, --scene-dist auto: the replicate plan (no data-path collective) is measured FIRST; the exchange plan — whose RCCL form has
    never run on more than one GPU — runs under a watchdog.  A collective that never returns (simulated) ends the run after
    --collective-timeout with the replicate line on stdout, rc 0, and the line says what happened."""
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    cubes, R = 2000, 60_000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cubes", str(cubes),
           "--rays", str(R), "--backend", "gloo", "--one-device", "--no-extra", "--no-parity", "--collective-timeout", "8"]
    p = subprocess.run(cmd, env=dict(env, BVH_BENCH_TEST_HANG_EXCHANGE="1"), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["scene_dist"] == "replicate" and out["value"] > 0
    assert "did not finish within 8 s" in out["collective_watchdog"]
    # the plan that did not come back is NAMED where a reader looks for the plans, beside the replicate result
    plans = out["scene_dist_plans"]
    assert plans["replicate"]["value"] == out["value"] and plans["bcast-torch"]["timed_out"] is True and plans["bcast-torch"]["after_s"] == 8
    assert out["rccl"]["formed"] is False
    assert out["roofline"]["kernel"] and out["launch"]["ranks_seen"] == 2
    # without the simulated hang both plans are on the line, the faster one is the headline
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert "collective_watchdog" not in out and set(out["scene_dist_plans"]) == {"replicate", "bcast-torch"}
    best = min(out["scene_dist_plans"], key=lambda k: out["scene_dist_plans"][k]["ms_per_step"])
    assert out["config"]["scene_dist"] == best and out["ms_per_step"] == out["scene_dist_plans"][best]["ms_per_step"]
    assert out["scene_dist_plans"]["replicate"]["hits_all_ranks"] == out["scene_dist_plans"]["bcast-torch"]["hits_all_ranks"]


def test_bench_four_ranks_share_the_gpu():
    """More ranks than the two-rank tests: four processes on cuda:0 over gloo, both plans (the 8-rank rehearsal of round 4 found a peer
    importing a scene blob that had not arrived yet — the torch transport now waits for it)."""
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    cubes, R = 3000, 50_000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "2", "--cubes", str(cubes),
           "--rays", str(R), "--backend", "gloo", "--one-device", "--no-extra"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 4 and out["launch"]["ranks_seen"] == 4 and out["parity"]["equal"] is True
    assert set(out["scene_dist_plans"]) == {"replicate", "bcast-torch"}
    from bvh_amd import testbase as tb
    from oracle import orc
    _, aabbs = tb.create_n_cubes(cubes)
    off, idx, _, _ = orc.traverse_flat(orc.flatten(orc.build(aabbs).nodes), aabbs, orc.create_rays(0, 4 * R))
    assert out["scene_dist_plans"]["replicate"]["hits_all_ranks"] == out["scene_dist_plans"]["bcast-torch"]["hits_all_ranks"] == len(idx)


def test_bench_n2_extras_carry_both_plans():
    """N > 1 with the extra_configs section ON (what the driver's scaling run executes): configs[3] strong — here a 4 M-ray stream instead of
    100 M, the stand-in at detail 4 — must come back with BOTH plans in its own scene_dist_plans, equal whole-job hit counts, parity on rank 0's
    shard, and the headline's plans beside it (the first SCALE record answers the exchange question whichever plan wins)."""
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BVH_BENCH_STRONG_RAYS"] = "4000000"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cubes", "2000", "--rays", "60000",
           "--backend", "gloo", "--one-device", "--extra-steps", "2", "--standin-detail", "4", "--settle-steps", "5"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and "collective_watchdog" not in out and set(out["scene_dist_plans"]) == {"replicate", "bcast-torch"}
    assert len(out["extra_configs"]) == 1
    e = out["extra_configs"][0]
    assert "error" not in e, e
    assert e["workload"] == "standin-incoherent" and e["scaling"] == "strong" and e["rays_total"] == 4_000_000 and e["rays_this_rank"] == 2_000_000
    assert set(e["scene_dist_plans"]) == {"replicate", "bcast-torch"}
    assert e["scene_dist_plans"]["replicate"]["hits_all_ranks"] == e["scene_dist_plans"]["bcast-torch"]["hits_all_ranks"] == e["hits_all_ranks"] > 0
    assert e["parity"]["equal"] is True and e["parity"]["checked_rays"] == 2_000_000


def test_rccl_info_names_the_one_rccl_of_the_process():
    """bvhgpu_rccl_info names the RCCL the C ABI resolved (the copy torch already holds must be shared, not a second one),
    and a communicator reports its own size."""
    import bvh_amd
    from bvh_amd import Context, dist as bdist
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    import torch  # noqa: F401  (bvh_amd loads torch first when it is installed: its bundled librccl is the one to share)
    info = bdist.rccl_info()
    assert info["library"] and "rccl" in info["library"].lower(), info
    assert info["version_code"] >= 20000, info
    mapped = {os.path.realpath(ln.split()[-1]) for ln in open("/proc/self/maps") if "librccl" in ln}
    assert len(mapped) == 1 and os.path.realpath(info["library"]) in mapped, (mapped, info)     # ONE RCCL in the process
    ctx = Context(0)
    comm = bdist.Communicator(ctx, 1, 0, bdist.Communicator.unique_id())
    ci = comm.info()
    assert (ci["nranks"], ci["first_rank"], ci["n_local"]) == (1, 0, 1) and ci["version"] == info["version"]
    comm.close()
