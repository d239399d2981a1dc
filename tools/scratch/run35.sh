timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tests/gpu_perf.py --reps 5 --tag binfast 2>&1 | grep PERF | cut -c1-120
timeout 300 python tests/gpu_perf.py --reps 5 --tag robust --robust 2>&1 | grep PERF | cut -c1-330
timeout 300 python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0,'.')
from embree_amd import api, workloads as W
dev = api.Device("")
meshes = W.synthetic_crown()
s = api.make_scene(dev, meshes, device_resident=True)
prim = W.crown_camera_rays(meshes, 1024, 1024)
tr = prim.copy(); s.intersect1M(tr)
rays = W.diffuse_bounce_rays(tr, meshes)
for rep in range(4):
    r = rays.copy(); t0 = time.perf_counter(); s.intersect1M(r); dt = time.perf_counter() - t0
    print("HOSTPATH rtcIntersect1M (96 MB up + down through pageable memory): %.2f ms -> %.1f Mrays/s" % (1e3*dt, rays.shape[0]/dt/1e6), flush=True)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof35 -o r35 -- python /root/repo/tests/gpu_perf.py --reps 2 > /dev/null 2>&1
cd /root/repo; python tools/kstats.py gpurun_out/prof35 | head -9
