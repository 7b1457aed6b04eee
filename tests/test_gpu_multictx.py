"""The multi-GPU protocol of the C ABI with SEVERAL ranks on the one GPU of a test box (round 3).

Real RCCL wants one GPU per rank and this project's GPU boxes have one, so `bvhgpu_bcast*` had only ever run with a
communicator of one rank.  tests/c_abi/fake_rccl.cpp is a single-process stand-in for the eight RCCL entry points
csrc/comm.hip calls (loaded through BVHGPU_RCCL_LIB; a broadcast = an event-ordered device copy; mismatched collectives are
an error instead of a hang).  With it K ctxs on device 0 are K ranks: the peers' receive path, the status header, the
error / rebroadcast paths and exact_only propagation run for real; the transport itself (RCCL over xGMI) does not.
On a multi-GPU node `python tests/multi_ctx_driver.py <ndev> real` runs the same checks over real RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "c_abi", "libfakerccl.so")


def _build_fake():
    src = os.path.join(ROOT, "tests", "c_abi", "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", FAKE, src])


@pytest.mark.parametrize("K", [2, 3, 8])
def test_multi_rank_protocol_on_one_gpu(K):
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    _build_fake()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_ctx_driver.py"), str(K)], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["unflattened_root_status"] == out["unflattened_root_expected"]
    assert out["bcast_header_form_equal"] and out["bcast_known_equal"]
    assert out["async_step_equal"] and out["root_nodes_equal"]
    assert out["wrong_announcement_ok"], out["wrong_announcement"]
    assert out["nan_input_ok"], out["nan_input"]
    assert out["unbalanced_rebroadcast_ok"], out["unbalanced_first_statuses"]
    assert out["unbalanced_steady_state_ok"]
    assert out["exact_only_hits"] > 0 and all(out["exact_only_travels"].values()), out["exact_only_travels"]
    assert all(out["triangles_travel"].values()), out["triangles_travel"]


@pytest.mark.parametrize("K", [2, 4, 8])
def test_rank_per_thread_protocol(K):
    """The process-per-GPU form (bvhgpu_comm_init_rank; every peer's root is REMOTE) with K threads as the K processes of torchrun
    and the stand-in library's barrier in ncclGroupEnd: bench.py's N > 1 step, the header form, a wrong announcement, the
    rebroadcast of an unbalanced first build."""
    import bvh_amd
    if bvh_amd.device_count() <= 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    _build_fake()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_thread_driver.py"), str(K)], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert not out["errors"] and not any(out["hung"]), out
    inv, reb = out["expect"]["INVALID_ARG"], out["expect"]["REBROADCAST"]
    for r, o in enumerate(out["ranks"]):
        assert o["async_step_equal"] and o["header_form_equal"], (r, o)
        assert o["wrong_announcement"] == inv, (r, o)
        assert o["unbalanced"] == [reb, 0, True], (r, o)
        assert o["unbalanced_steady"] == [0, True], (r, o)


def test_rccl_is_loaded_lazily():
    """libbvh_mi355x.so has no link-time dependency on librccl: single-GPU consumers load and run without it, and
    a process without any RCCL gets BVHGPU_RCCL_ERROR from the communicator calls instead of a loader failure."""
    so = os.path.join(ROOT, "bvh_amd", "libbvh_mi355x.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['BVHGPU_RCCL_LIB'] = '/nonexistent/librccl.so'; os.environ['BVH_AMD_NO_TORCH'] = '1';"
            "import numpy as np; from bvh_amd import Bvh, Context, dist, testbase as tb; from bvh_amd._lib import BvhGpuError, RCCL_ERROR;"
            "ctx = Context(0); _, a = tb.create_n_cubes(50); t = Bvh.from_aabbs(a, ctx); t.flatten_in_place();\n"
            "try:\n    dist.Communicator(ctx, 1, 0, bytes(128)); print('NO ERROR')\n"
            "except BvhGpuError as e:\n    print('status', e.status == RCCL_ERROR)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "status True" in p.stdout, (p.stdout, p.stderr[-2000:])
