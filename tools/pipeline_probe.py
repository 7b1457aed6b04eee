#!/usr/bin/env python3
"""Probe: S independent step loops (build + flatten + traverse of configs[1]) on S HIP streams from S host threads.
Does the latency-bound build of one step overlap the traversal of another?  Prints whole-GPU Mrays/s per S."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from bvh_amd._lib import RAY_F32

R, K = 1_000_000, 100
dev = torch.device("cuda", 0)
bounds = tb.default_bounds()
_, aabbs_np = tb.create_n_cubes(10000, bounds)
aabbs = torch.from_numpy(aabbs_np).to(dev)


class Lane:
    def __init__(self, first):
        self.ctx = Context(0)                       # its own non-blocking stream
        self.buf = torch.empty(R * RAY_F32.itemsize, dtype=torch.uint8, device=dev)
        self.rays = RayBatch.generate(first, R, bounds, self.buf, np.float32, self.ctx)
        self.bvh = Bvh.from_aabbs(aabbs, self.ctx)
        self.bvh.flatten_in_place()

    def run(self, k):
        for _ in range(k):
            self.bvh.rebuild(aabbs, flatten=True)
            self.bvh.traverse_batch(self.rays, fetch=False)


for S in (1, 2, 3, 4):
    lanes = [Lane(i * R) for i in range(S)]
    for ln in lanes:
        ln.run(5)
    torch.cuda.synchronize()
    th = [threading.Thread(target=ln.run, args=(K,)) for ln in lanes]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S}: {S * K * R / dt / 1e6:.0f} Mrays/s  ({dt / K * 1e3:.3f} ms per round of {S} steps)")
