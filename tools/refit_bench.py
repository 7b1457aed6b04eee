#!/usr/bin/env python3
"""refit vs rebuild on BASELINE configs[1]'s scene (120 000 triangles): HIP-event time of bvhgpu_refit (+ re-flatten)
against bvhgpu_rebuild_flat, inputs resident in HBM."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, testbase as tb

dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
_, a = tb.create_n_cubes(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000)
a0 = torch.from_numpy(a).to(dev)
a1 = (a0.view(-1, 2, 3) + torch.rand(len(a), 1, 3, device=dev) * 4 - 2).reshape(-1, 6).contiguous()
bvh = Bvh.from_aabbs(a0, ctx)
bvh.flatten_in_place()
for name, fn in (("rebuild_flat", lambda x: bvh.rebuild(x, flatten=True)), ("refit+flatten", lambda x: bvh.refit(x))):
    for _ in range(5):
        fn(a1); fn(a0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for i in range(K):
        fn(a1 if i & 1 else a0)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / K * 1e3:.4f} ms per call ({len(a)} shapes)")
