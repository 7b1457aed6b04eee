"""Build libbvh_mi355x.so (the C-ABI engine, include/bvh_mi355x.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.  Flags that matter for bit-exact parity with the Rust reference:
  -ffp-contract=off   no FMA contraction (every IEEE op is rounded separately, like rustc)
  (no -ffast-math; hipcc's default keeps f32 divide / sqrt correctly rounded)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libbvh_mi355x.so")
SOURCES = ["capi.hip", "build.hip", "flatten.hip", "traverse.hip", "refit.hip", "comm.hip", "obj.cpp"]
HEADERS = ["common.hpp", "engine.hpp", "obj.cpp", os.path.join("..", "..", "include", "bvh_mi355x.h")]
# -fno-slp-vectorize: ROCm 7.2's SLP vectoriser + gfx950 instruction selection crash (SIGSEGV in
# constrainRegClass) on the integer-key min/max folds of sah_select(); packed v_pk_* VALU ops are no
# gain for this code anyway (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_variant(out: str, defines, verbose: bool = True) -> str:
    """developer builds with extra -D flags (instrumentation, tuning sweeps) into another file; load with BVH_AMD_SO=<out>"""
    cmd = [hipcc()] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return OUT
    # RCCL (the multi-GPU broadcast of the C ABI, comm.hip) is NOT linked: comm.hip dlopen()s librccl.so.1 on first use — a copy the
    # process already holds (PyTorch-ROCm bundles one) is shared, single-GPU consumers need none.  <rccl/rccl.h> is used for types only.
    cmd = [hipcc()] + FLAGS + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libbvh_mi355x.so")
    if verbose and res.stderr.strip():
        sys.stderr.write(res.stderr)
    return OUT


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python build_ext.py --variant <out.so> DEFINE[=v] ...
        k = sys.argv.index("--variant")
        print(build_variant(sys.argv[k + 1], sys.argv[k + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
