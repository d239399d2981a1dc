"""CPU tests of the host-side logic: workload generators, RNG restatement, sharding, gloo world_size 2."""
import os
import subprocess
import sys

import numpy as np
import pytest

from embree_amd import shard, workloads as W
from embree_amd.rtypes import INVALID_ID, make_rayhits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_sampler_matches_reference_definition():
    """RandomSampler (tutorials/common/math/random_sampler.h): scalar restatement vs the vectorised one."""
    def scalar(idx, n):
        M = 0xFFFFFFFF
        k = (idx * 0xCC9E2D51) & M
        k = ((k << 15) | (k >> 17)) & M
        k = (k * 0x1B873593) & M
        h = k
        h = ((((h << 13) | (h >> 19)) & M) * 5 + 0xE6546B64) & M
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & M
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M
        h ^= h >> 16
        out = []
        for _ in range(n):
            h = (h * 1664525 + 1013904223) & M
            out.append(np.float32(h >> 1) * np.float32(4.656612873077392578125e-10))
        return out
    rs = W.RandomSampler(np.arange(50, dtype=np.uint32))
    a, b = rs.get_float(), rs.get_float()
    for i in (0, 1, 7, 49):
        s = scalar(i, 2)
        assert a[i] == s[0] and b[i] == s[1]
    assert (a >= 0).all() and (a < 1).all()


def test_scene_generators_are_deterministic_and_sized():
    c = W.cube_and_plane()
    assert W.num_triangles(c) == 14 and len(c) == 2
    cb = W.cornell_box()
    assert W.num_triangles(cb) == 34
    v, t = W.triangle_sphere([0, 0, 0], 1.0, 10)
    assert v.shape[0] == 20 * 11 and t.shape[0] == 2 * 20 * 9 and t.max() < v.shape[0]
    m1, m2 = W.synthetic_crown(num_phi=8), W.synthetic_crown(num_phi=8)
    assert all((a[0] == b[0]).all() and (a[1] == b[1]).all() for a, b in zip(m1, m2))
    assert len(m1) == 49 and W.num_triangles(m1) == 48 * 2 * 16 * 7 + 12
    # the full-size stand-in is 48 * 2*316*157 + 12 triangles (count only, no allocation)
    assert 48 * 2 * 316 * 157 + 12 == 4762764
    pp = W.synthetic_powerplant(target_tris=60000)
    assert abs(W.num_triangles(pp) - 60000) < 700 and pp[0][1].max() < pp[0][0].shape[0]


def test_ray_generators():
    r = W.cube_camera_rays()
    assert r.shape[0] == 1024 and (r["geomID"] == INVALID_ID).all() and (r["mask"] == 0xFFFFFFFF).all()
    d = np.stack([r["dir_x"], r["dir_y"], r["dir_z"]], -1)
    assert np.allclose((d * d).sum(-1), 1, atol=1e-5)
    meshes = W.synthetic_crown(num_phi=6)
    prim = W.crown_camera_rays(meshes, 16, 16)
    fake = prim.copy()
    fake["geomID"][::2] = 0                                   # pretend every other ray hit something at t=1 with Ng=+y
    fake["tfar"][::2] = 1.0
    fake["Ng_y"][::2] = 1.0
    b = W.diffuse_bounce_rays(fake, meshes)
    bd = np.stack([b["dir_x"], b["dir_y"], b["dir_z"]], -1)
    nn = np.where((d_ := np.stack([prim["dir_x"], prim["dir_y"], prim["dir_z"]], -1))[:, 1:2] > 0, -1.0, 1.0)
    assert (bd[::2, 1] * nn[::2, 0] >= -1e-6).all()           # bounce goes into the hemisphere of the face-forwarded normal
    assert (b["tnear"] > 0).all() and np.isinf(b["tfar"]).all()
    sh = W.shadow_rays(fake[:32], meshes, samples=16)
    assert sh.shape[0] == 512 and sh.dtype.itemsize == 48 and (sh["tfar"] > 0).all()
    assert W.diffuse_bounce_rays(fake, meshes).tobytes() == b.tobytes()


def test_shard_ranges_tile_exactly():
    for total in (0, 1, 7, 1 << 20, 16777216, 1000003):
        for world in (1, 2, 3, 4, 8):
            r = [shard.shard_range(total, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)
    assert shard.aggregate_throughput([10, 30], 2.0) == 20.0


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
from embree_amd import shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lo, hi = shard.shard_range(16777216, rank, world)             # config 4: 16M shadow rays sharded over the ranks
counts = shard.gather_counts(hi - lo, dist)
import numpy as np
mine = (np.arange(lo, lo + 1000, dtype=np.uint32) * 2654435761 %% 4294967291).astype(np.uint32)   # stand-in for a rank's packed results
allr = shard.gather_host(mine, dist)                           # what the RCCL all-gather does on the GPUs: rank order, every rank gets everything
want = np.concatenate([(np.arange(shard.shard_range(16777216, r, world)[0], shard.shard_range(16777216, r, world)[0] + 1000, dtype=np.uint32) * 2654435761 %% 4294967291).astype(np.uint32) for r in range(world)])
assert allr.dtype == np.uint32 and (allr == want).all()
elapsed = shard.max_over_ranks(0.5 + rank, dist)               # slowest rank defines the step time
dist.barrier()
if rank == 0:
    print("RESULT", sum(counts), elapsed, shard.aggregate_throughput(counts, elapsed))
dist.destroy_process_group()
'''


def test_world_size_2_gloo(tmp_path):
    """The N>1 host path (rendezvous, barrier, MAX over ranks, aggregate) with 2 CPU processes over gloo."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-800:] for o in outs]
    line = [l for l in outs[0][0].decode().splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 16777216 and float(line[2]) == 1.5 and abs(float(line[3]) - 16777216 / 1.5) < 1e-3


def test_shadow_ray_shards_are_slices_of_the_single_rank_set():
    """configs[3]: a rank generates only its contiguous shard of the 16 x hit-points shadow rays; the shards of N ranks must be exactly the rays one
    rank would generate (same RandomSampler indices), or the gathered result would not be the single-rank answer."""
    meshes = W.synthetic_crown(num_phi=8)
    rng = np.random.default_rng(3)
    n = 96
    b = make_rayhits(rng.random((n, 3), dtype=np.float32) * 3 + 1, rng.random((n, 3), dtype=np.float32) - 0.5)
    b["tfar"] = rng.random(n, dtype=np.float32) * 2
    b["geomID"] = np.where(rng.random(n) < 0.8, 0, INVALID_ID).astype(np.uint32)
    full = W.shadow_rays(b, meshes, samples=16)
    assert full.shape[0] == 16 * n
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            lo, hi = shard.shard_range(16 * n, r, world)
            assert lo % 16 == 0 and hi % 16 == 0
            parts.append(W.shadow_rays(b[lo // 16: hi // 16], meshes, samples=16, first=lo))
        assert np.concatenate(parts).tobytes() == full.tobytes()


_LAUNCHED = r'''
import os, sys
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
if len(sys.argv) > 2 and sys.argv[2] == "die" and rank == 1:
    sys.exit(3)                                               # a rank that fails before the rendezvous
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
open(os.path.join(sys.argv[1], "rank%d.txt" % rank), "w").write("%d %d" % (rank, world))
dist.barrier()
dist.destroy_process_group()
'''


def test_bench_launcher_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher around it must start N ranks itself (bench.spawn_ranks): rendezvous on 127.0.0.1, RANK / LOCAL_RANK /
    WORLD_SIZE set, exit code 0 only if every rank succeeded -- and a rank that dies must not leave the others waiting for ever."""
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "launched.py"
    script.write_text(_LAUNCHED)
    assert bench.spawn_ranks(2, script=str(script), script_args=[str(tmp_path)]) == 0
    assert (tmp_path / "rank0.txt").read_text() == "0 2" and (tmp_path / "rank1.txt").read_text() == "1 2"
    t0 = __import__("time").time()
    assert bench.spawn_ranks(2, script=str(script), script_args=[str(tmp_path), "die"]) != 0
    assert __import__("time").time() - t0 < 120
