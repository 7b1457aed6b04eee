"""CPU-only checks (no compute calls): the C-ABI library loads and exports every symbol
include/bvh_mi355x.h declares, POD layouts match the header, errors are reported (never a silent CPU
fallback), the input generators equal the oracle's, and the N>1 path works over gloo (world_size 2).
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bvh_mi355x.h")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()  # hipcc cross-compiles gfx950 without a GPU
    from bvh_amd import _lib
    return _lib


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bvhgpu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    lib = built.load()
    declared = _declared_functions()
    assert len(declared) >= 30
    bound = {name for name, _, _ in built.SYMBOLS}
    assert set(declared) == bound, set(declared) ^ bound
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libbvh_mi355x.so"
    nm = subprocess.run(["nm", "-D", "--defined-only", built.SO_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (bvhgpu_[a-z0-9_]+)", nm))
    assert set(declared) <= exported
    assert lib.bvhgpu_abi_version() == built.ABI_VERSION == 7


def test_library_contains_gfx950_code_object(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={built.SO_PATH}"], capture_output=True, text=True)
    blob = open(built.SO_PATH, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"k_traverse", b"k_traverse_lds", b"k_traverse_ordered", b"k_traverse_heap", b"k_prep", b"k_bin", b"k_split", b"k_mid",
                 b"k_small", b"k_flatten", b"k_nearest", b"k_gen_rays"):
        assert kern in blob, kern


def test_pod_layouts_match_header(built):
    assert built.NODE_F32.itemsize == 64 and built.NODE_F64.itemsize == 112
    assert built.FLAT_F32.itemsize == 36 and built.FLAT_F64.itemsize == 64
    assert built.RAY_F32.itemsize == 36 and built.RAY_F64.itemsize == 72
    # the same layouts compile to the same sizes with a C compiler
    prog = r'''
#include <stdio.h>
#include "bvh_mi355x.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(bvhgpu_node_f32), sizeof(bvhgpu_node_f64),
 sizeof(bvhgpu_flat_f32), sizeof(bvhgpu_flat_f64), sizeof(bvhgpu_ray_f32), sizeof(bvhgpu_ray_f64),
 sizeof(bvhgpu_traverse_stats), sizeof(bvhgpu_timings)); return 0;}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = subprocess.check_output([exe], text=True).split()
    assert sizes == ["64", "112", "36", "64", "36", "72", "40", "16"]
    assert C.sizeof(built.TraverseStats) == 40 and C.sizeof(built.Timings) == 16
    # oracle layouts are the same PODs (tests byte-compare across the two)
    from oracle import orc
    assert orc.NODE_F32 == built.NODE_F32 and orc.FLAT_F32 == built.FLAT_F32 and orc.RAY_F32 == built.RAY_F32
    assert orc.NODE_F64 == built.NODE_F64 and orc.FLAT_F64 == built.FLAT_F64 and orc.RAY_F64 == built.RAY_F64


def test_no_device_is_an_error_not_a_fallback(built, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is visible here")
    import bvh_amd
    assert bvh_amd.device_count() == 0
    h = C.c_void_p()
    rc = built.load().bvhgpu_create(0, None, C.byref(h))
    assert rc == built.NO_DEVICE and not h.value
    assert b"device" in built.load().bvhgpu_status_string(rc)
    with pytest.raises(bvh_amd.BvhGpuError):
        bvh_amd.Context(0)
    with pytest.raises(bvh_amd.BvhGpuError):
        bvh_amd.Bvh.from_aabbs(np.zeros((4, 6), np.float32))
    # NULL handles are rejected, not dereferenced
    assert built.load().bvhgpu_flatten(None) == built.INVALID_ARG
    assert built.load().bvhgpu_tree_info(None, None, None, None, None) == built.INVALID_ARG


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); import bvh_amd._lib as L; L.SO_PATH = %r; L._lib = None\n"
            "try:\n    L.load()\nexcept ImportError as e:\n    print('IMPORTERROR', 'no CPU fallback' in str(e).lower() or 'fallback' in str(e))\n"
            % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, BVH_AMD_NO_TORCH="1"))
    assert "IMPORTERROR True" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "bvh_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"liboracle", r"\borc_[a-z]", r"#include[^\n]*oracle", r"\borc\."):
                    assert not re.search(pat, text, flags=re.M), (os.path.join(dirpath, f), pat)
    assert not re.search(r"liboracle|orc_", open(HEADER).read())
    # and the engine does not link it
    ldd = subprocess.run(["ldd", os.path.join(pkg, "libbvh_mi355x.so")], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_generators_equal_oracle():
    from bvh_amd import testbase as tb
    from oracle import orc
    t, a = tb.create_n_cubes(1000)
    t2, a2 = orc.create_n_cubes(1000)
    assert t.tobytes() == t2.tobytes() and a.tobytes() == a2.tobytes()
    assert tb.generate_aligned_boxes_aabbs().tobytes() == orc.aligned_boxes().tobytes()
    boxes = tb.generate_aligned_boxes()
    assert [b.id for b in boxes] == list(range(-10, 11))
    assert np.array_equal(np.stack([b.aabb().as6() for b in boxes]), orc.aligned_boxes())
    tri = tb.Triangle(t[5, 0], t[5, 1], t[5, 2])
    assert np.array_equal(tri.aabb().as6(), a[5])
    # next_point3_at seeks: draw j of the seed-0 stream
    pts = tb.next_point3_at(np.array([1, 2, 3, 2_000_001], dtype=np.uint64), tb.default_bounds())
    r = orc.create_rays(0, 1)
    assert np.array_equal(pts[0], r[0]["o"])
    r2 = orc.create_rays(1_000_000, 1)
    assert np.array_equal(pts[3], r2[0]["o"])


def test_shard_ranges():
    from bvh_amd import dist as bdist
    R = 1000
    spans = [bdist.shard_range(r, 8, R) for r in range(8)]
    assert spans[0] == (0, R) and spans[7] == (7 * R, R)
    assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(7))
    with pytest.raises(ValueError):
        bdist.shard_range(8, 8, R)
    # strong scaling (configs[3]: 100 M rays over 8 GPUs): the shards tile the stream exactly for any total / world
    for total, world in ((100_000_000, 8), (1_000_003, 7), (5, 8), (0, 3)):
        shards = [bdist.strong_shard(r, world, total) for r in range(world)]
        assert shards[0][0] == 0 and sum(c for _, c in shards) == total
        assert all(shards[r][0] + shards[r][1] == shards[r + 1][0] for r in range(world - 1))


_GLOO_WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, %(root)r)
os.environ["BVH_AMD_NO_TORCH"] = "0"
import torch
import torch.distributed as dist
from bvh_amd import dist as bdist, testbase as tb
from oracle import orc
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
R = 1500
_, aabbs = tb.create_n_cubes(60)
# rank 0 owns the scene (stands in for build+flatten+scene_export); the peers get it by ONE broadcast
if rank == 0:
    flat = orc.flatten(orc.build(aabbs).nodes)
    payload = np.frombuffer(flat.tobytes() + aabbs.tobytes(), dtype=np.uint8).copy()
else:
    payload = None
n = bdist.broadcast_nbytes(len(payload) if rank == 0 else 0, "cpu", 0)
blob = torch.from_numpy(payload) if rank == 0 else torch.empty(n, dtype=torch.uint8)
bdist.broadcast_scene(blob, 0)
raw = blob.numpy().tobytes()
nflat = 3 * len(aabbs) - 2
flat = np.frombuffer(raw[:nflat * 36], dtype=orc.FLAT_F32)
sa = np.frombuffer(raw[nflat * 36:], dtype=np.float32).reshape(-1, 6)
first, cnt = bdist.shard_range(rank, world, R)
rays = orc.create_rays(first, cnt)
off, idx, _, st = orc.traverse_flat(flat, sa, rays)
pickle.dump((first, off, idx), open(%(out)r + str(rank), "wb"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    """world_size 2 over gloo on CPU: one scene broadcast, rays sharded, per-rank hit lists concatenate
    to exactly the single-process result (the oracle stands in for the GPU traversal here)."""
    import pickle
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "res")
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % dict(root=ROOT, port=port, out=out))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se.decode()[-2000:]
    from bvh_amd import testbase as tb
    from oracle import orc
    _, aabbs = tb.create_n_cubes(60)
    flat = orc.flatten(orc.build(aabbs).nodes)
    off, idx, _, _ = orc.traverse_flat(flat, aabbs, orc.create_rays(0, 3000))
    parts = [pickle.load(open(out + str(r), "rb")) for r in range(2)]
    assert parts[0][0] == 0 and parts[1][0] == 1500
    cat_idx = np.concatenate([parts[0][2], parts[1][2]])
    cat_off = np.concatenate([parts[0][1][:-1], parts[1][1] + parts[0][1][-1]])
    assert np.array_equal(cat_idx, idx) and np.array_equal(cat_off, off)


def test_bench_gpus_flag_can_never_be_downgraded():
    """`python bench.py --gpus 8` without torchrun used to fall through to ONE rank and print n_gpus = 1.  Now the flag
    and the launcher must agree, and without a launcher bench.py starts the ranks itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.resolve_launch(1, {}) == ("run", 1)
    assert bench.resolve_launch(1, {"WORLD_SIZE": "1"}) == ("run", 1)
    for n in (2, 4, 8):
        assert bench.resolve_launch(n, {}) == ("spawn", n)                       # no launcher: N ranks are started, never 1
        assert bench.resolve_launch(n, {"WORLD_SIZE": ""}) == ("spawn", n)
        assert bench.resolve_launch(n, {"WORLD_SIZE": str(n), "RANK": "0", "LOCAL_RANK": "0"}) == ("run", n)
        for ws in ("1", str(n + 1), "x"):                                        # the launcher disagrees with the flag: hard error
            with pytest.raises(SystemExit):
                bench.resolve_launch(n, {"WORLD_SIZE": ws, "RANK": "0", "LOCAL_RANK": "0"})
        with pytest.raises(SystemExit):
            bench.resolve_launch(n, {"WORLD_SIZE": str(n)})                      # WORLD_SIZE without RANK: not a launched rank
    with pytest.raises(SystemExit):
        bench.resolve_launch(1, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"})   # torchrun with 8 ranks but --gpus 1
    with pytest.raises(SystemExit):
        bench.resolve_launch(0, {})
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "5"]


def test_bench_self_launch_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the dry run shows the torch.distributed.run command it
    becomes; the real thing starts two ranks — on this GPU-less box both stop at "needs an MI355X", and no JSON line claiming
    n_gpus = 1 is ever printed."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=dict(env, BVH_BENCH_LAUNCH_DRY_RUN="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["launch"]
    assert "--nproc-per-node=2" in cmd and cmd[-6:] == ["--gpus", "2", "--steps", "1", "--warmup", "0"]
    import torch
    if torch.cuda.is_available():
        return                                   # (on a GPU box tests/test_gpu_dist.py runs the real thing to the end)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--one-device"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # the ranks were really started (torchrun ends the others as soon as one has failed, so not every rank gets to say it)
    assert p.stderr.count("bench.py needs an MI355X") >= 1 and "torch.distributed.run" in p.stderr, p.stderr[-3000:]


def test_rccl_is_shared_with_the_process_not_loaded_twice():
    """comm.hip dlopen()s RCCL; when the process already holds a copy (PyTorch bundles librccl.so) that copy must be the
    one used — a second RCCL in the process would open the devices again.  bvhgpu_rccl_info names what was resolved (no GPU needed)."""
    code = ("import os, json, torch\n"
            "from bvh_amd import dist\n"
            "i = dist.rccl_info()\n"
            "m = sorted({os.path.realpath(l.split()[-1]) for l in open('/proc/self/maps') if 'librccl' in l})\n"
            "print(json.dumps({'info': i, 'mapped': m}))\n")
    import json
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(r["mapped"]) == 1, r
    assert os.path.realpath(r["info"]["library"]) == r["mapped"][0] and r["info"]["shared_with_process"] is True, r
    assert r["info"]["version_code"] >= 20000


# ---- the Rust declarations against the header, signature by signature (no cargo here, so this is the check) ----------
_C_SCALARS = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "float": "f32",
              "double": "f64", "char": "c_char", "void": "c_void"}


def _c_type_to_rust(decl):
    """`const float *aabbs` / `bvhgpu_tree **out` / `bvhgpu_ctx *const *ctxs` / `const float bounds[6]` → the Rust spelling ffi.rs must use"""
    decl = " ".join(decl.replace("*", " * ").split())
    is_array = bool(re.search(r"\[\d*\]$", decl))
    decl = re.sub(r"\s*\[\d*\]$", "", decl)
    toks = decl.split()
    if toks and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-1] not in _C_SCALARS and not toks[-1].startswith("bvhgpu_") and toks[-1] != "const":
        toks = toks[:-1]                                         # the parameter's name
    elif len(toks) >= 2 and re.fullmatch(r"[A-Za-z_]\w*", toks[-1]) and toks[-2] not in ("const",) and toks[-1] not in _C_SCALARS and toks[-1] != "*":
        toks = toks[:-1]
    # base type with its own const
    base_const = False
    i = 0
    if toks[i] == "const":
        base_const, i = True, i + 1
    base = toks[i]
    i += 1
    if i < len(toks) and toks[i] == "const":                     # `T const`
        base_const, i = True, i + 1
    rust = _C_SCALARS.get(base, base)
    pointee_const = base_const
    for t in toks[i:]:
        if t == "*":
            rust = ("*const " if pointee_const else "*mut ") + rust
            pointee_const = False
        elif t == "const":
            pointee_const = True                                 # qualifies the pointer just made: matters for the NEXT level
        else:
            raise AssertionError(f"cannot parse C declarator {decl!r}")
    if is_array:                                                 # `T name[N]` as a parameter is `T *`
        rust = ("*const " if base_const else "*mut ") + rust
    return rust


def _header_prototypes():
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    h = re.sub(r"//[^\n]*", "", h)
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[\w]+(?:\s*\*+)?)\s*(bvhgpu_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.M):
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [_c_type_to_rust(a) for a in args.split(",")]
        ret = " ".join(ret.replace("*", " * ").split())
        out[name] = (params, None if ret == "void" else _c_type_to_rust(ret + " x") if not ret.endswith("*") else _c_type_to_rust(ret))
    return out


def _rust_prototypes():
    src = open(os.path.join(ROOT, "rust", "bvh-mi355x", "src", "ffi.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for name, args, ret in re.findall(r"pub fn (bvhgpu_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", src):
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            params.append(" ".join(a.split(":", 1)[1].split()))
        out[name] = (params, " ".join(ret.split()) if ret else None)
    return out


def test_rust_ffi_signatures_match_the_header():
    c, r = _header_prototypes(), _rust_prototypes()
    assert len(c) >= 76 and set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in sorted(c):
        cp, cr = c[name]
        rp, rr = r[name]
        assert len(cp) == len(rp), f"{name}: {len(cp)} parameters in the header, {len(rp)} in ffi.rs"
        for k, (a, b) in enumerate(zip(cp, rp)):
            assert a == b, f"{name}, parameter {k}: header says {a}, ffi.rs says {b}"
        assert cr == rr, f"{name}: returns {cr} in the header, {rr} in ffi.rs"
    # the ctypes table has the same arity for every symbol (types are looser there: void* for every pointer)
    from bvh_amd import _lib
    table = {n: (len(a), res) for n, res, a in _lib.SYMBOLS}
    assert set(table) == set(c)
    for name, (n_args, res) in table.items():
        assert n_args == len(c[name][0]), f"{name}: {n_args} argtypes in _lib.py, {len(c[name][0])} parameters in the header"
        assert (res is None) == (c[name][1] is None), name


def test_rust_ffi_structs_and_constants_match_the_header():
    """field order / types of the #[repr(C)] mirrors and every #define / enum value ffi.rs restates"""
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    src = open(os.path.join(ROOT, "rust", "bvh-mi355x", "src", "ffi.rs")).read()
    lib_rs = open(os.path.join(ROOT, "rust", "bvh-mi355x", "src", "lib.rs")).read()
    ty = {"float": "f32", "double": "f64", "uint32_t": "u32", "uint64_t": "u64"}
    for name, body in re.findall(r"typedef struct\s*\{([^}]*)\}\s*(bvhgpu_\w+);", h):
        name, body = body, name                                  # (regex groups: body first)
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            t, rest = decl.split(None, 1)
            for f in [x.strip() for x in rest.split(",")]:
                m = re.fullmatch(r"(\w+)(?:\[(\d+)\])?", f)
                fields.append((m.group(1), f"[{ty[t]}; {m.group(2)}]" if m.group(2) else ty[t]))
        m = re.search(r"pub struct %s\s*\{([^}]*)\}" % name, src)
        assert m, f"ffi.rs has no #[repr(C)] struct {name}"
        rust_fields = [(a.strip().replace("pub ", "").split(":")[0].strip(), a.split(":")[1].strip()) for a in m.group(1).split(",") if a.strip()]
        assert rust_fields == fields, (name, rust_fields, fields)
        assert re.search(r"#\[repr\(C\)\][^\n]*\n?\s*pub struct %s\b" % name, src) or re.search(r"#\[repr\(C\)\]\s*#\[derive[^\]]*\]\s*\n?pub struct %s\b" % name, src), name
    consts = dict(re.findall(r"pub const (BVHGPU_\w+): \w+ = ([^;]+);", src))
    defines = dict(re.findall(r"#define (BVHGPU_\w+) ([0-9]+)u?\b", h))
    enums = {}
    for body in re.findall(r"enum\s+\w*\s*\{([^}]*)\}", h) + re.findall(r"typedef enum\s*\{([^}]*)\}", h):
        nxt = 0
        for item in [x.strip() for x in body.split(",") if x.strip()]:
            m = re.fullmatch(r"(BVHGPU_\w+)(?:\s*=\s*([^,]+))?", item)
            if not m:
                continue
            if m.group(2) is not None:
                nxt = int(eval(m.group(2).replace("u", ""), {}, {}))
            enums[m.group(1)] = nxt
            nxt += 1
    known = {**{k: int(v) for k, v in defines.items()}, **enums}
    checked = 0
    for k, v in consts.items():
        if k in known:
            rv = {"u32::MAX": 0xFFFFFFFF}.get(v.strip(), None)
            rv = int(v) if rv is None else rv
            assert rv == known[k], f"{k}: {rv} in ffi.rs, {known[k]} in the header"
            checked += 1
    assert checked >= 25, checked
    # the shim implements the trait for both scalar types and knows its device
    assert "impl<T: GpuScalar> BoundingHierarchy<T, 3> for GpuBvh<T>" in lib_rs
    assert "pub type GpuBvh64 = GpuBvh<f64>" in lib_rs and "pub fn traverse_closest" in lib_rs and "pub fn device(&self)" in lib_rs
    for sym in re.findall(r"ffi::(bvhgpu_\w+)\b", lib_rs):       # every entry point / type lib.rs names exists in ffi.rs
        assert re.search(r"\b%s\b" % sym, src), sym


def test_bench_watchdog_prints_what_was_measured():
    """bench.py's guard around the never-yet-run N > 1 exchange (RCCL communicator + broadcast steps): a section that blocks inside a
    foreign call is abandoned after the timeout by a helper thread that prints the line measured so far and ends the process."""
    code = ("import os, sys, time, json, importlib.util\n"
            "spec = importlib.util.spec_from_file_location('bench_mod', %r)\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "line = {'value': 123.0, 'scene_dist': 'replicate'}\n"
            "def fire():\n"
            "    line['collective_watchdog'] = 'timed out'\n"
            "    os.write(1, (json.dumps(line) + '\\n').encode()); os._exit(0)\n"
            "with b.Watchdog(0.2, fire):\n"
            "    pass\n"                              # a section that finishes in time: nothing fires
            "time.sleep(0.5)\n"
            "with b.Watchdog(0.5, fire):\n"
            "    time.sleep(60)\n"                   # (blocks with the GIL released, like a hung ncclGroupEnd under ctypes)
            "print('NOT REACHED')\n") % os.path.join(ROOT, "bench.py")
    import json
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and time.time() - t0 < 30, p.stderr[-1000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1 and "NOT REACHED" not in p.stdout
    assert json.loads(lines[0]) == {"value": 123.0, "scene_dist": "replicate", "collective_watchdog": "timed out"}


def test_bench_timed_out_exchange_plan_is_named_on_the_line():
    """when the watchdog abandons the exchange plan (the RCCL broadcast nobody has run on more than one GPU), the line it
    prints must say so where a reader looks for the plans — scene_dist_plans carries the plan that did not come back with
    "timed_out": true, the stage it was in and the workload, beside the replicate result; `rccl` says how far the communicator got.  The same
    for an exchange that hangs inside the extra_configs section (configs[3] strong): the entry's own scene_dist_plans names it."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod_wd", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # headline: replicate measured and published, the exchange plan pending in the communicator
    res = {"value": 20000.0, "ms_per_step": 0.4, "scene_dist_plans": {"replicate": {"value": 20000.0, "ms_per_step": 0.4, "hits_all_ranks": 80000}}}
    line = {"value": res["value"], "n_gpus": 8, "workload_name": "cubes120k", "scene_dist_plans": res["scene_dist_plans"], "rccl": None}
    pending = {"res": res, "plan": "bcast", "stage": "forming the RCCL communicator", "workload": "cubes120k"}
    text = b.timed_out_line(line, pending, "the exchange plan of cubes120k", 60.0, {"nranks": None, "formed": False, "error": None})
    out = json.loads(text)
    assert out["scene_dist_plans"]["bcast"] == {"timed_out": True, "after_s": 60.0, "stage": "forming the RCCL communicator", "workload": "cubes120k"}
    assert out["scene_dist_plans"]["replicate"]["value"] == 20000.0 and out["value"] == 20000.0
    assert out["rccl"] == {"nranks": None, "formed": False, "error": None} and "did not finish within 60 s" in out["collective_watchdog"]
    # extras: the headline finished with both plans; configs[3] strong hangs in its exchange steps after its replicate result was published
    res3 = {"workload": "standin-incoherent", "value": 30000.0, "scene_dist_plans": {"replicate": {"value": 30000.0, "hits_all_ranks": 457389170}}}
    line = {"value": 21000.0, "n_gpus": 8, "workload_name": "cubes120k",
            "scene_dist_plans": {"replicate": {"value": 20000.0}, "bcast": {"value": 21000.0}}, "extra_configs": [res3]}
    pending = {"res": res3, "plan": "bcast", "stage": "the exchange plan's steps (communicator formed)", "workload": "standin-incoherent"}
    out = json.loads(b.timed_out_line(line, pending, "the exchange plan of standin-incoherent", 60.0, {"nranks": 8, "version": "2.26.6"}))
    e = out["extra_configs"][0]
    assert e["scene_dist_plans"]["bcast"]["timed_out"] is True and e["scene_dist_plans"]["replicate"]["hits_all_ranks"] == 457389170
    assert set(out["scene_dist_plans"]) == {"replicate", "bcast"} and "timed_out" not in out["scene_dist_plans"]["bcast"]
    assert out["rccl"]["nranks"] == 8
    # a hang outside any exchange plan (pending None): the note and rccl only
    out = json.loads(b.timed_out_line({"value": 1.0}, None, "the extra_configs section", 300.0, None))
    assert "did not finish within 300 s" in out["collective_watchdog"] and "scene_dist_plans" not in out
