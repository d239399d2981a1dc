#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ray-tracing core (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] = "crown, 2^20 incoherent diffuse-bounce rays,
closest-hit, 1 x MI355X".  crown.ecs is not shipped with the reference, so the scene is the seeded
synthetic stand-in of embree_amd/workloads.py (4,762,764 triangles, 49 geometries); rays are the
cosine-weighted bounce rays of a 1024x1024 camera image, generated with the reference's RandomSampler.
A "step" = one closest-hit pass over one batch of 2^20 rays through rtcIntersect1MDevice, rays already
resident in HBM (every step has its own pristine copy of the batch, staged before the timed region).
Steps are issued round-robin on --streams HIP streams (default 4), the way a wavefront renderer keeps several
ray batches in flight: the persistent traversal kernel fills the chip, so the next batch's blocks start as the
blocks of the previous one retire and the tail of a batch (lanes that have run out of rays, 22 % of the
lane-iterations of a lone 2^20-ray launch) overlaps with useful work.  --streams 1 gives the lone-launch number;
it is measured in every run as well and reported under roofline.serial.

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU): the BVH is replicated
(every rank builds it from the same inputs), each rank traces its own 2^20-ray batch (weak scaling);
the ray path has no exchange step, so there is no data-path collective; the barrier / max-over-ranks
uses torch.distributed (gloo) on the host.

Printed JSON (rank 0, one line): metric/value/... as the driver contract, plus
  roofline      achieved = ALGORITHMIC bytes per launch / (average kernel time / launches in flight), kernel time
                measured with HIP events on the stream each launch is issued on (kernel_ms_avg: agrees with rocprofv3
                --kernel-trace of this command); launches in flight = sum of kernel times / wall time of the timed
                region (concurrency).  roofline.serial = the same kernel launched alone, back to back on one stream.
                bytes = rays*(48 read + 52 written on hit) + visited nodes*80 + fetched
                triangle records*48 (visit counts from the counting build of the same kernel, same rays).
  cpu_baseline  the REAL reference (oracle/_ref, Embree 4.4.1 AVX2) looping rtcIntersect1 over the same
                rays on all host threads (kind "reference"), or the scalar C restatement on a sample (kind "port").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes its streams onto 4 hardware queues by default, one of them the null stream's: streams that share a queue
# run one after the other.  Ask for 8 so that every batch stream gets its own queue (must be set before the runtime starts).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, loaders, workloads as W           # noqa: E402  (loads the HIP library before anything else)
from embree_amd.rtypes import RAYHIT_DTYPE, INVALID_ID         # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy rate)
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_bench_latest.json")   # written by tools/pmc_summary.py from rocprofv3 --pmc passes of THIS command


def pmc_traffic():
    """HBM bytes per launch of the traversal kernel from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), or None."""
    try:
        d = json.load(open(PMC_JSON))
        return int(d["hbm_traffic_bytes_per_launch"])
    except Exception:
        return None



def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(meshes, rays, budget_s=25.0):
    """Reference leg, rank 0 / N=1 only.  Never touches the GPU path."""
    from oracle import refembree, restate
    if refembree.available():
        threads = refembree.hw_threads()
        s = refembree.RefScene("threads=%d" % threads)
        for v, t in meshes:
            s.add_mesh(v, t)
        build_s = s.commit()
        n = rays.shape[0]
        best, reps, spent = None, 0, 0.0
        warm = rays.copy()
        s.intersect1(warm, threads)
        while reps < 5 and spent < budget_s:
            r = rays.copy()
            dt = s.intersect1(r, threads)
            best = dt if best is None else min(best, dt)
            spent += dt
            reps += 1
        out = dict(value=n / best / 1e6, unit="Mrays/s", cores=threads, kind="reference",
                   sample="all %d rays of the step, rtcIntersect1 in 1024-ray blocks on %d threads, best of %d; "
                          "Embree 4.4.1 AVX2 single-ISA build (oracle/ref.mk); CPU build %.2f s = %.1f Mprims/s"
                          % (n, threads, reps, build_s, W.num_triangles(meshes) / build_s / 1e6))
        s.close()
        return out, warm
    if not restate.available():
        return None, None
    s = restate.OracleScene()
    for v, t in meshes:
        s.add_mesh(v, t)
    s.commit()
    n = min(rays.shape[0], 65536)
    r = rays[:n].copy()
    t0 = time.time()
    s.intersect1(r)
    dt = time.time() - t0
    return dict(value=n / dt / 1e6, unit="Mrays/s", cores=1, kind="port",
                sample="first %d rays of the step, scalar C restatement (oracle/restate.c)" % n), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--rays", type=int, default=1 << 20)
    ap.add_argument("--streams", type=int, default=4, help="HIP streams the steps are issued on round-robin (ray batches in flight)")
    ap.add_argument("--phi", type=int, default=158, help="sphere tessellation of the synthetic crown (158 -> 4.76M triangles)")
    ap.add_argument("--config", default="", help="extra rtcNewDevice config, e.g. max_leaf=2,int_cost=0.5")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--scene", default="", help=".ecs / .xml / .obj scene file; default: $EMBREE_MODEL_DIR/crown/crown.ecs if it exists, "
                                                 "else the synthetic crown stand-in")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        log("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    L = api.load()
    ngpu = L.mi355_device_count()
    if ngpu <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    gpu = local % ngpu
    dev = api.Device(("gpu=%d," % gpu) + args.config)

    dist = None
    if world > 1:
        import torch.distributed as dist_mod              # host-side rendezvous only (gloo); the HIP library is already bound
        import torch
        # Gloo announces its connections on the C-level stdout: keep stdout for the one JSON line (fd 1 -> stderr while the group forms)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
            dist_mod.barrier()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        dist = (dist_mod, torch)

    # ---- scene: replicated BVH, geometry resident on the device
    t0 = time.time()
    scene_path = args.scene or loaders.find_model("crown")
    if scene_path:                                         # the real asset when the box has it (the reference does not ship it)
        loaded = loaders.load_scene(scene_path)
        meshes, camera = loaded.meshes, loaded.camera
        scene_name = "%s (%s)" % (os.path.basename(scene_path), "loaded with embree_amd/loaders.py")
    else:
        meshes, camera = W.synthetic_crown(num_phi=args.phi), None
        scene_name = "synthetic-crown (crown.ecs is not shipped)"
    ntri = W.num_triangles(meshes)
    gen_s = time.time() - t0
    scene = api.Scene(dev)
    for v, t in meshes:
        scene.add_triangle_mesh(v, t, device_resident=True)      # rtcSetSharedGeometryBufferHostDevice: no upload at commit
    build_ms, commit_wall = [], []
    for rep in range(3):                                   # rtcCommitScene timed like buildbench (buildbench_device.cpp:385-387)
        t0 = time.time()
        scene.commit()
        commit_wall.append(time.time() - t0)
        build_ms.append(scene.info()["build_ms"])
    info = scene.info()
    # the same scene with RTC_BUILD_QUALITY_LOW (Morton build), reported next to the SAH build; the timed rays use the SAH tree
    low_ms = []
    L.rtcSetSceneBuildQuality(scene.h, api.RTC_BUILD_QUALITY_LOW)
    for rep in range(3):
        scene.commit()
        low_ms.append(scene.info()["build_ms"])
    low_info = scene.info()
    L.rtcSetSceneBuildQuality(scene.h, api.RTC_BUILD_QUALITY_MEDIUM)
    scene.commit()
    assert scene.info()["num_nodes"] == info["num_nodes"]

    # ---- rays: primary image traced on the GPU -> diffuse bounce rays (this rank's own seed)
    side = int(round(args.rays ** 0.5))
    prim = (W.camera_rays(camera["vp"], camera["vi"], camera["vu"], camera["fov"], side, side) if camera
            else W.crown_camera_rays(meshes, side, side))
    M = prim.shape[0]
    dprim = api.DeviceArray.from_numpy(prim, gpu)
    scene.intersect1M_device(dprim.ptr, M)
    L.mi355_device_synchronize(gpu)
    traced = dprim.download(RAYHIT_DTYPE)
    dprim.free()
    rays = W.diffuse_bounce_rays(traced, meshes, seed=1 + rank)

    streams = []
    for _ in range(max(1, args.streams)):
        st_ = C.c_void_p()
        L.mi355_stream_create(gpu, C.byref(st_))
        streams.append(st_)
    stream = streams[0]
    nserial = min(args.steps, 10)                             # lone-launch leg (roofline.serial), after the timed region
    nbuf = args.steps + args.warmup + nserial
    pristine = api.DeviceArray.from_numpy(rays, gpu)
    bufs = [api.DeviceArray(rays.nbytes, gpu) for _ in range(nbuf)]
    for b in bufs:
        L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, stream)
    L.mi355_synchronize(stream)

    # ---- visit counts for the algorithmic-bytes figure (counting build of the same kernel, same rays)
    dstat = api.DeviceArray.from_numpy(rays, gpu)
    st = scene.trace_stats(dstat.ptr, M, 96)
    result = dstat.download(RAYHIT_DTYPE)
    dstat.free()
    nhit = int((result["geomID"] != INVALID_ID).sum())
    alg_bytes = M * 48 + nhit * 52 + st["nodes"] * 80 + st["tris"] * 48

    def barrier():
        if dist:
            dist[0].barrier()

    for st_ in streams:                                     # per-stream traversal scratch exists before anything is timed (also when --warmup 0)
        assert L.mi355_trace_prepare(scene.bvh(), st_) == 0, L.mi355_last_error()
    for i in range(args.warmup):
        scene.intersect1M_device(bufs[i].ptr, M, 96, streams[i % len(streams)])
    L.mi355_device_synchronize(gpu)

    ev = [C.c_void_p() for _ in range(2 * args.steps)]
    for e in ev:
        L.mi355_event_create(C.byref(e))
    bvh = scene.bvh()
    barrier()
    L.mi355_device_synchronize(gpu)
    t0 = time.perf_counter()
    for k in range(args.steps):
        # rtcIntersect1MDevice's launch (mi355_trace_closest) with a HIP event on either side of the kernel
        rc = L.mi355_trace_timed(bvh, bufs[args.warmup + k].ptr, M, 96, 0, streams[k % len(streams)], ev[2 * k], ev[2 * k + 1])
        assert rc == 0, L.mi355_last_error()
    L.mi355_device_synchronize(gpu)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = dist[1].tensor([elapsed], dtype=dist[1].float64)
        dist[0].all_reduce(tt, op=dist[0].ReduceOp.MAX)
        elapsed = float(tt[0])
    kernel_ms = []
    for k in range(args.steps):
        ms = C.c_float()
        L.mi355_event_elapsed_ms(ev[2 * k], ev[2 * k + 1], C.byref(ms))
        kernel_ms.append(ms.value)
    # lone launches, back to back on one stream (outside the timed region): the kernel's own duration
    evs = [C.c_void_p() for _ in range(2 * nserial)]
    for e in evs:
        L.mi355_event_create(C.byref(e))
    t1 = time.perf_counter()
    for k in range(nserial):
        rc = L.mi355_trace_timed(bvh, bufs[args.warmup + args.steps + k].ptr, M, 96, 0, stream, evs[2 * k], evs[2 * k + 1])
        assert rc == 0, L.mi355_last_error()
    L.mi355_device_synchronize(gpu)
    serial_elapsed = time.perf_counter() - t1
    serial_ms = []
    for k in range(nserial):
        ms = C.c_float()
        L.mi355_event_elapsed_ms(evs[2 * k], evs[2 * k + 1], C.byref(ms))
        serial_ms.append(ms.value)
    # every timed buffer must hold the same answer as the counting run (same rays, same tree)
    for b in (bufs[args.warmup], bufs[args.warmup + args.steps - 1], bufs[-1]):
        assert b.download(RAYHIT_DTYPE).tobytes() == result.tobytes(), "timed kernel and counting kernel disagree"

    if rank == 0:
        avg_ms = float(np.mean(kernel_ms))
        conc = float(np.sum(kernel_ms)) * 1e-3 / elapsed       # launches in flight, averaged over the timed region
        eff_ms = avg_ms / max(conc, 1.0)                       # a launch that shares the chip with c-1 others gets 1/c of it
        ser_ms = float(np.mean(serial_ms)) if serial_ms else avg_ms
        value = world * M * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/s (incoherent diffuse, closest-hit) on crown", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not scene_path else "file",
            "config": {"workload": "configs[2]: %s, %d triangles, %d geometries, "
                                   "%d incoherent diffuse-bounce rays per GPU, closest-hit, rays + BVH resident in HBM"
                                   % (scene_name, ntri, len(meshes), M),
                       "rays_per_gpu": M, "triangles": ntri, "batches_in_flight": len(streams),
                       "parallelism": "rays sharded x%d, BVH replicated, no collective" % world,
                       "device_config": args.config},
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (eff_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes / (eff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(),
                         "kernel": "trace_kernel_q<closest>", "kernel_ms_avg": round(avg_ms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                         "concurrency": round(conc, 3),
                         "note": "achieved = algorithmic bytes (SURVEY 8d) per launch / (kernel time / launches in flight); most of these bytes are served by L1/L2/Infinity "
                                 "Cache (traffic = HBM bytes per launch from the PMC passes), so the fraction of the HBM peak can exceed 1; the kernel is VALU-issue bound",
                         "serial": {"kernel_ms_avg": round(ser_ms, 4), "kernel_ms_min": round(float(np.min(serial_ms)), 4) if serial_ms else None,
                                    "achieved": round(alg_bytes / (ser_ms * 1e-3) / 1e9, 1), "frac": round(alg_bytes / (ser_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "mrays_per_s": round(M * nserial / serial_elapsed / 1e6, 1) if nserial else None, "launches": nserial},
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "per_ray": {"nodes": round(st["nodes"] / M, 2), "node_step_simd_util": round(st["nodes"] / max(1, 64 * st["node_blocks"]), 3),
                                     "tri_step_simd_util": round(st["tris"] / max(1, 64 * st["tri_blocks"]), 3),
                                     "triangles": round(st["tris"] / M, 2), "bytes": round(alg_bytes / M, 1)}},
            "build": {"metric": "BVH build Mprims/s", "gpu_build_ms": round(float(np.min(build_ms)), 3),
                      "mprims_per_s_gpu": round(ntri / (float(np.min(build_ms)) * 1e-3) / 1e6, 1),
                      "commit_wall_ms": round(1e3 * float(np.min(commit_wall)), 3),
                      "mprims_per_s_commit": round(ntri / float(np.min(commit_wall)) / 1e6, 1),
                      "nodes": info["num_nodes"], "leaves": info["num_leaves"], "sah": round(info["sah"], 3),
                      "bvh_bytes": info["bytes_nodes"] + info["bytes_triangles"],
                      "low_quality": {"what": "RTC_BUILD_QUALITY_LOW: Morton-code build, same node/leaf layout", "gpu_build_ms": round(float(np.min(low_ms)), 3),
                                      "mprims_per_s_gpu": round(ntri / (float(np.min(low_ms)) * 1e-3) / 1e6, 1), "sah": round(low_info["sah"], 3)}},
            "hit_fraction": round(nhit / M, 4),
        }
        if world == 1 and not args.no_cpu:
            try:
                cb, ref_traced = cpu_baseline(meshes, rays)
                if cb:
                    out["cpu_baseline"] = cb
                if ref_traced is not None:                    # free parity check at full size against the real reference
                    same = (ref_traced["primID"] == result["primID"]) & (ref_traced["geomID"] == result["geomID"])
                    out["parity_vs_reference"] = {"rays": M, "id_mismatch": int((~same).sum()),
                                                  "max_rel_t_err": float(np.max(np.abs(ref_traced["tfar"][same] - result["tfar"][same]) /
                                                                                np.maximum(np.abs(ref_traced["tfar"][same]), 1e-30))) if same.any() else None}
            except Exception as e:                            # the baseline leg must never take the GPU number down
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    barrier()
    if dist:
        dist[0].destroy_process_group()


if __name__ == "__main__":
    main()
