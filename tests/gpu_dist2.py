"""Worker of test_gpu_multi (one process per rank; 2 ranks may share one GPU): configs[3] in miniature.
Every rank commits the same scene, traces its contiguous shard of the shadow rays with rtcOccluded1MDevice, packs the 4-byte results on the GPU
(mi355_pack_occluded) and the shards are gathered -- over RCCL when every rank has its own GPU or the world is 1, else (RCCL refuses two ranks on one
device) through the host with gloo, which exercises the same sharding / packing / ordering logic.  Rank 0 compares with the single-rank answer."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, shard, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE, RAY_DTYPE

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
transport = sys.argv[1] if len(sys.argv) > 1 else "rccl"
dist = None
if os.environ.get("MI355_IMPORT_TORCH"):                         # the situation bench.py is in at N > 1: PyTorch (with its private HIP runtime and RCCL) is loaded before
    import torch.distributed as _td                              # the library's first RCCL call; RCCL must still bind to the runtime that owns our allocations
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
L = api.load()
ngpu = L.mi355_device_count()
gpu = int(os.environ.get("LOCAL_RANK", "0")) % ngpu
dev = api.Device("gpu=%d" % gpu)
meshes = W.synthetic_crown(num_phi=24)
scene = api.make_scene(dev, meshes)
info = scene.info()
prim = W.crown_camera_rays(meshes, 64, 64)
scene.intersect1M(prim)
bounce = W.diffuse_bounce_rays(prim, meshes, seed=1)
scene.intersect1M(bounce)
total = 16 * bounce.shape[0]
lo, hi = shard.shard_range(total, rank, world)
rays = W.shadow_rays(bounce[lo // 16: hi // 16], meshes, samples=16, first=lo)
M = rays.shape[0]
d = api.DeviceArray.from_numpy(rays, gpu)
scene.occluded1M_device(d.ptr, M)
packed = api.DeviceArray(4 * M, gpu)
assert L.mi355_pack_occluded(d.ptr, M, 48, packed.ptr, None) == 0
L.mi355_device_synchronize(gpu)
assert scene.trace_status() == 0
if transport == "rccl":
    comm = shard.Communicator(gpu, rank, world, dist)
    allr = api.DeviceArray(4 * M * world, gpu)
    comm.allgather(packed.ptr, allr.ptr, 4 * M)
    L.mi355_device_synchronize(gpu)
    gathered = allr.download(np.uint32)
    comm.close()
else:
    gathered = shard.gather_host(packed.download(np.uint32), dist)
if rank == 0:
    # the single-rank answer: all rays in one launch
    full = W.shadow_rays(bounce, meshes, samples=16)
    assert full.shape[0] == total and full[lo:hi].tobytes() == rays.tobytes()       # the shard IS the slice of the single-rank ray set
    scene.occluded1M(full)
    want = full["tfar"].view(np.uint32)
    assert gathered.shape[0] == total and (gathered == want).all(), "gathered occlusion results differ from the single-rank answer"
    # closest-hit records through mi355_pack_hits
    dh = api.DeviceArray.from_numpy(bounce, gpu)
    ph = api.DeviceArray(32 * bounce.shape[0], gpu)
    assert L.mi355_pack_hits(dh.ptr, bounce.shape[0], 96, ph.ptr, None) == 0
    L.mi355_device_synchronize(gpu)
    h = ph.download(np.uint32).reshape(-1, 8)
    b32 = bounce.view(np.uint32).reshape(-1, 24)
    assert (h[:, 0] == b32[:, 8]).all() and (h[:, 1] == b32[:, 15]).all() and (h[:, 2] == b32[:, 16]).all() and (h[:, 3] == b32[:, 17]).all()   # tfar, u, v, primID
    assert (h[:, 4] == b32[:, 18]).all() and (h[:, 5:8] == b32[:, 12:15]).all()                                                                      # geomID, Ng
    print("DIST OK world=%d transport=%s rays=%d occluded=%d nodes=%d" % (world, transport, total, int((want == 0xFF800000).sum()), info["num_nodes"]), flush=True)
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
