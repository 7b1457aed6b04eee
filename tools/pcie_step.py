import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from bvh_amd import Bvh, Context, RayBatch, testbase as tb
from oracle import orc
dev = torch.device("cuda", 0)
ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
_, aabbs = tb.create_n_cubes(10000)
rays = orc.create_rays(0, 1_000_000)
pin_r = torch.from_numpy(rays.view(np.uint8)).pin_memory()
pin_a = torch.from_numpy(aabbs).pin_memory()
bvh = Bvh.from_aabbs(aabbs, ctx); bvh.flatten_in_place()
rb_page = RayBatch(len(rays), np.float32, host=rays)
rb_pin = RayBatch(len(rays), np.float32, host=pin_r.numpy().view(rays.dtype))
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("step from PAGEABLE host buffers (aabbs + rays uploaded each step): %.3f ms" % t(lambda: (bvh.rebuild(aabbs, flatten=True), bvh.traverse_batch(rb_page, fetch=False))))
print("step from PINNED host buffers: %.3f ms" % t(lambda: (bvh.rebuild(pin_a.numpy(), flatten=True), bvh.traverse_batch(rb_pin, fetch=False))))
d_a = torch.from_numpy(aabbs).to(dev)
buf = torch.empty(len(rays) * 36, dtype=torch.uint8, device=dev)
rd = RayBatch.generate(0, len(rays), tb.default_bounds(), buf, np.float32, ctx)
print("step HBM-resident: %.3f ms" % t(lambda: (bvh.rebuild(d_a, flatten=True), bvh.traverse_batch(rd, fetch=False))))
