import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import bvh_amd as eng
from bvh_amd import testbase as tb
from oracle import orc
n_cubes = int(sys.argv[1])
t0 = time.time(); _, aabbs = tb.create_n_cubes(n_cubes); print("gen", time.time() - t0, len(aabbs))
t0 = time.time(); bvh = eng.Bvh.from_aabbs(aabbs); bvh.ctx.synchronize(); print("gpu build (incl upload)", time.time() - t0, "levels", bvh.build_levels)
t0 = time.time(); bvh.rebuild(aabbs); print("gpu rebuild (incl upload)", time.time() - t0)
t0 = time.time(); ot = orc.build(aabbs, threads=orc.max_threads(), schedule="fast"); print("oracle build", time.time() - t0)
print("nodes identical:", bvh.nodes.tobytes() == ot.nodes.tobytes())
flat = bvh.flatten(); oflat = orc.flatten(ot.nodes)
print("flat identical:", flat.nodes.tobytes() == oflat.tobytes())
rays = orc.create_rays(0, 200000)
off, idx, _, st = flat.traverse_batch(eng.RayBatch(len(rays), np.float32, host=rays), stats=True)
ooff, oidx, _, ost = orc.traverse_flat(oflat, aabbs, rays, threads=orc.max_threads())
print("csr identical:", np.array_equal(off, ooff) and np.array_equal(idx, oidx), st["visited"] / len(rays))
