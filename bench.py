#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ray-tracing core (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                       # configs[2], the headline
    python bench.py --gpus N --workload shadow16m --gather rccl         # configs[3]: 16 Mi shadow rays, strong scaling, results gathered over RCCL

Workload `crown` (default; config.workload): BASELINE.json configs[2] = "crown, 2^20 incoherent diffuse-bounce rays, closest-hit, 1 x MI355X".
crown.ecs is not shipped with the reference, so the scene is the seeded synthetic stand-in of embree_amd/workloads.py (4,762,764 triangles,
49 geometries) unless $EMBREE_MODEL_DIR/crown/crown.ecs exists; rays are the cosine-weighted bounce rays of a 1024 x 1024 camera image, generated
with the reference's RandomSampler.  A "step" = one closest-hit pass over one batch of 2^20 rays through rtcIntersect1MDevice, rays already
resident in HBM (every step has its own pristine copy of the batch, staged before the timed region).  The timed region issues the K steps back to
back on ONE HIP stream (--streams 1, the default since round 4): one 2^20-ray batch at a time, which is the metric as BASELINE.json words it
("1M incoherent diffuse rays"); `value` is that.  Several batches in flight (--pipeline-streams 4: the way a wavefront renderer drives
rtcIntersect1MDevice -- the persistent traversal kernel fills the chip, so the next batch's workgroups start as those of the previous one retire
and the tail of a batch overlaps useful work) is measured right after the timed region and reported as the named extra "pipelined".

Workload `shadow16m`: configs[3] = 16 Mi shadow rays (16 per hit point of the configs[2] rays) through rtcOccluded1MDevice, STRONG scaling: the
16,777,216 rays are sharded contiguously over the N ranks (embree_amd/shard.py), each rank packs its 4-byte results and the shards are gathered on
every GPU with one ncclAllGather over xGMI (--gather rccl; 64 MB in total); a step = the whole 16 Mi-ray job incl. pack + gather.

Multi-GPU (--gpus N, launched by torch.distributed.run, one process per GPU): the BVH is replicated (every rank builds it from the same inputs:
the build is deterministic) and nothing of the rays crosses xGMI before they are traced.  For N > 1 the north star's "hits gathered over RCCL/xGMI" is
INSIDE the timed step of both workloads: `crown` (weak scaling, 2^20 rays per rank and step) packs the fields a hit writes (32 of the 96 bytes) and
ncclGather-s them to rank 0; `shadow16m` (strong scaling) all-gathers the 4-byte results.  The collective of batch k runs on a communication stream,
ordered behind the pack by an event, and overlaps the trace of batch k + 1 (the trace kernels still run one at a time); the timed region ends when
every gather has landed; the root checks every rank's block against that rank's own checksums.  If RCCL cannot form a communicator the line is still
printed, with "rccl_ranks": 0, the reason under "gather" and the metric string saying that nothing was gathered.  Rendezvous, barrier and
MAX-over-ranks use torch.distributed (gloo) on the host; the data path uses RCCL through the library's own C ABI (mi355_comm_*).

Printed JSON (rank 0, one line): metric/value/... as the driver contract, plus
  roofline      achieved / peak / frac as SURVEY 8(d) prescribes: ALGORITHMIC bytes per launch / kernel duration of the TIMED launches (HIP events on the
                launch stream; agrees with rocprofv3 --kernel-trace of this command) / 8 TB/s; bytes = rays*(48 read + 52 written on hit) + visited
                nodes*80 + fetched triangle records*48, visit counts from the counting build of the same kernel on the same rays.  Most of those bytes
                are L1/L2/Infinity-Cache hits ("frac_is": cache-served), so next to it: `hbm_counter_frac` / `hbm_counter_from_profile` = what reaches
                the memory side (PMC FETCH_SIZE x2 + WRITE_SIZE per launch, from profiles/pmc_bench_latest.json, only if that file was collected for
                THIS kernel source -- else null), and `bound`: what the kernel actually follows -- VALU issue (`valu_from_profile`; round 4 cut the
                (lane, load) pairs by 16 % at +16 % VALU work: -8 %, profiles/r04_pair_records.md), with `address_rate` (scattered lane-addresses per
                second against 256 CUs x 1 per clock) beside it.
  pipelined     the same kernel with --pipeline-streams (4) batches in flight (with --streams > 1 the roles swap: `serial` = one batch at a time)
  end_to_end    rtcIntersect1M on a pageable host array: H2D + kernel + D2H (PCIe-inclusive; never `value`)
  cpu_baseline  the REAL reference (oracle/_ref, Embree 4.4.1, AVX2 and AVX-512 single-ISA builds) looping rtcIntersect1 over 16 x the step's rays on a
                persistent pinned pool of all host threads (kind "reference"; value_1Mi, rtcIntersect8/16 and rtcCommitScene beside it), or the scalar C
                restatement on a sample (kind "port")
  parity_vs_reference  the timed kernel's output against the reference on all rays, exact-t ties classified (tests/helpers.py); bench.py FAILS on a mismatch
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP multiplexes its streams onto 4 hardware queues by default, one of them the null stream's: streams that share a queue
# run one after the other.  Ask for 8 so that every batch stream gets its own queue (must be set before the runtime starts).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # (RCCL between processes: the host driver only supports dmabuf IPC)
from embree_amd import api, loaders, shard, workloads as W           # noqa: E402  (loads the HIP library before anything else)
from embree_amd.rtypes import RAYHIT_DTYPE, RAY_DTYPE, INVALID_ID, rays_of   # noqa: E402
from tools.kernel_hash import trace_kernel_hash                        # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy rate)
NUM_SIMDS = 256 * 4     # CUs x SIMDs
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_bench_latest.json")   # written by tools/pmc_summary.py from rocprofv3 --pmc passes of THIS command


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_pmc():
    """PMC counters per launch of the closest-hit kernel, but only if they were collected for the kernel source that is being run."""
    try:
        d = json.load(open(PMC_JSON))
    except Exception:
        return None, "no PMC file"
    if d.get("source_hash") != trace_kernel_hash():
        return None, "profiles/pmc_bench_latest.json was collected for another build of the kernel (source hash %s, now %s)" % (d.get("source_hash"), trace_kernel_hash())
    return d, None


# ----------------------------------------------------------------------------------------------------------------- CPU leg (rank 0, N = 1)
def cpu_baseline(meshes, rays, any_hit=False, budget_s=25.0):
    """Reference leg.  Never touches the GPU path.
    What is timed, and how (the round-3 harness created its 255 threads inside the clock of a 10 ms job):
      * rays: rtcIntersect1 / rtcOccluded1 in 1024-ray blocks on a PERSISTENT pinned pool (oracle/ref_driver.cpp), on 16 Mi rays -- the job size of the reference's
        own ParallelIntersectBenchmark (tutorials/verify/verify.cpp:5923-5983: 16 tiles of the step's rays) -- and on the step's own 2^20 rays beside it;
        `value` is the 16 Mi figure of the fastest ISA build that runs on this host (AVX-512 next to AVX2, oracle/ref.mk ISA=avx512);
      * packets: rtcIntersect8 / rtcIntersect16 over the same rays in ray order, for context (the reference's packet code path on incoherent rays);
      * build: rtcCommitScene wall time, new scene per repetition (buildbench_device.cpp:385-387), internal tasking started up front and pinned
        ("start_threads=1,set_affinity=1"), at the thread count that is fastest on this host."""
    from oracle import refembree, restate
    ntri = W.num_triangles(meshes)
    if refembree.available():
        hw = refembree.hw_threads()
        # A container may see every hardware thread and still be allowed only a few CPUs' worth of time (cgroup cpu.max: quota per period): a 5 ms job on 256
        # threads then runs at full speed and a 1 s job at the quota -- measured on the round-4 box: 200 Mrays/s for 2^20 rays, 19 Mrays/s for 16 x 2^20, the
        # same 19-23 with 64, 128 or 256 threads (tools/cpu_probe.py, profiles/r04_cpu_baseline.md).  Both figures are reported and the quota is named.
        quota = None
        for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                            ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
            try:
                quota = parse(open(path).read().strip())
                break
            except Exception:                                  # noqa: BLE001
                continue
        isas = [i for i in ("avx2", "avx512") if refembree.available(i)]
        t_start = time.time()
        # ---- build
        build = {}
        for th in sorted({hw, min(hw, 128), min(hw, 64), min(hw, 32), min(hw, 16)}, reverse=True):
            times = []
            for rep in range(4):
                sc = refembree.RefScene("threads=%d,start_threads=1,set_affinity=1" % th, isa=isas[-1])
                for v, t in meshes:
                    sc.add_mesh(v, t)
                dt = sc.commit()
                sc.close()
                if rep:
                    times.append(dt)
                if time.time() - t_start > 40.0 and times:
                    break
            build[th] = times
        best_th = min(build, key=lambda k: min(build[k]))
        bt = build[best_th]
        # ---- rays
        n = rays.shape[0]
        src = rays_of(rays) if (any_hit and rays.dtype.itemsize == 96) else rays
        per_isa, warm = {}, None
        for isa in isas:
            s = refembree.RefScene("threads=%d" % hw, isa=isa)
            for v, t in meshes:
                s.add_mesh(v, t)
            s.commit()
            if warm is None:
                warm = rays.copy()                             # (the AVX2 library's answers: what the parity block compares with)
                (s.occluded1 if any_hit else s.intersect1)(warm, hw)
            s.run_tiled(src, 1, hw, any_hit)                   # warm-up: pool started, tree paged in
            d1 = []
            for _ in range(7):
                if quota is not None:
                    time.sleep(0.25)                           # (a fresh cgroup period: the burst must not start on a quota the job before it used up)
                d1.append(s.run_tiled(src, 1, hw, any_hit))
            d16, spent = [], 0.0
            while len(d16) < 5 and spent < budget_s / len(isas):
                dt = s.run_tiled(src, 16, hw, any_hit)        # 16 Mi records (verify.cpp's job size), filled by the pool's own threads
                d16.append(dt)
                spent += dt + 0.3
            rec = dict(mrays_16Mi=16 * n / min(d16) / 1e6, mrays_16Mi_median=16 * n / float(np.median(d16)) / 1e6,
                       mrays_1Mi=n / min(d1) / 1e6, mrays_1Mi_median=n / float(np.median(d1)) / 1e6, native_ray16=s.native16())
            if not any_hit:                                    # packet entry points on the same (incoherent) rays, in ray order
                for K in (8, 16):
                    dk = []
                    for _ in range(3):
                        r = rays.copy()
                        if quota is not None:
                            time.sleep(0.25)
                        dk.append(s.packet(K, r, threads=hw))
                    rec["rtcIntersect%d_mrays_1Mi" % K] = n / min(dk) / 1e6
            per_isa[isa] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in rec.items()}
            s.close()
        best_isa = max(per_isa, key=lambda k: max(per_isa[k]["mrays_16Mi"], per_isa[k]["mrays_1Mi"]))
        sustained, burst = per_isa[best_isa]["mrays_16Mi"], per_isa[best_isa]["mrays_1Mi"]
        throttled = quota is not None and quota < 0.9 * hw and burst > 2.0 * sustained
        out = dict(value=burst if throttled else sustained, unit="Mrays/s", cores=hw, kind="reference", isa=best_isa, cpu_quota_cores=quota,
                   value_is=("the 2^20-ray job (a ~5 ms burst on all %d hardware threads): this container's cgroup allows %.1f CPUs' worth of time per period, which is what the 16 Mi-ray "
                             "job (value_16Mi, ~1 s) runs at whatever the thread count -- the burst figure is the one that says what the host's cores can do" % (hw, quota)) if throttled else
                            "the 16 Mi-ray job (value_1Mi beside it)",
                   value_16Mi=sustained, median=per_isa[best_isa]["mrays_16Mi_median"], value_1Mi=burst, per_isa=per_isa,
                   sample="16 x the %d rays of the step = %d rays in a buffer the pool's own threads fill (first touch where it is traced), %s in 1024-ray blocks on a persistent pool of %d pinned threads started before the clock (FTZ/DAZ), best of <= 5; "
                          "value_1Mi: the step's own rays, best of 7; Embree 4.4.1 single-ISA builds (oracle/ref.mk): %s"
                          % (n, 16 * n, "rtcOccluded1" if any_hit else "rtcIntersect1", hw, ", ".join(isas)),
                   build=dict(mprims_per_s=ntri / min(bt) / 1e6, best_s=min(bt), median_s=float(np.median(bt)), threads=best_th, reps=len(bt), isa=isas[-1],
                              what="rtcCommitScene wall time, new scene per repetition, tasking threads started and pinned up front (start_threads=1,set_affinity=1), 1 warm-up + %d timed; "
                                   "thread counts tried: %s" % (len(bt), {k: round(min(v), 3) for k, v in build.items()})))
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
    (s.occluded1 if any_hit else s.intersect1)(r)
    dt = time.time() - t0
    return dict(value=n / dt / 1e6, unit="Mrays/s", cores=1, kind="port",
                sample="first %d rays of the step, scalar C restatement (oracle/restate.c)" % n), None


def reference_visits(meshes, rays, sample=1 << 16):
    """Visit counters of the REFERENCE-STRUCTURE tree (oracle/restate.c: the reference's BVH8 of Triangle4 blocks, ordered descent; the counters of
    kernels/common/stat.h:9-19) on a sample of the timed rays, to stand beside roofline.per_ray.  CPU, test infrastructure."""
    from oracle import restate
    if not restate.available():
        return None
    o = restate.OracleScene()
    for v, t in meshes:
        o.add_mesh(v, t)
    o.commit()
    n = min(sample, rays.shape[0])
    idx = np.linspace(0, rays.shape[0] - 1, n).astype(np.int64)          # spread over the whole batch
    r = np.ascontiguousarray(rays[idx])
    o.visit_stats(reset=True)
    o.intersect1(r)
    vs = o.visit_stats()
    c = o.counts()
    o.close()
    return dict(rays_sampled=int(n), nodes_per_ray=round(vs["nodes"] / n, 2), leaves_per_ray=round(vs["leaves"] / n, 2), triangle4_blocks_per_ray=round(vs["blocks"] / n, 2),
                tree=dict(nodes=c["nodes"], triangle4_blocks=c["blocks"]),
                what="oracle/restate.c: the reference's tree (BVH8 AABB nodes, leaves of <= 7 Triangle4 blocks, bvh_builder_sah.h:214-308) and its ordered single-ray descent "
                     "(bvh_traverser1.h:311-433) on an evenly spaced sample of the timed rays; a block is four triangles tested by one SIMD instruction")


def classify_parity(got, want, rays_in, meshes):
    """IDs bit-exact except classified exact-t ties (the checker's t of the GPU's triangle within 4 ulp of the reference's t, at most 1e-4 of the rays), t within 1e-4:
    the same check the GPU tests use; raises AssertionError on a real difference."""
    from oracle import restate
    from tests.helpers import compare_closest
    o = restate.OracleScene()
    for v, t in meshes:
        o.add_mesh(v, t)
    st = compare_closest(got, want, rays_in, o.triangle_t, max_tie_frac=1e-4, label="bench vs reference")
    same = (got["primID"] == want["primID"]) & (got["geomID"] == want["geomID"])
    hit = same & (want["geomID"] != INVALID_ID)
    rel = float(np.max(np.abs(got["tfar"][hit] - want["tfar"][hit]) / np.maximum(np.abs(want["tfar"][hit]), 1e-30))) if hit.any() else None
    return dict(rays=int(got.shape[0]), id_mismatch=int((~same).sum()), classified_exact_t_ties=st["ties"], unexplained=0, max_rel_t_err=rel)


# ----------------------------------------------------------------------------------------------------------------- helpers
class Events:
    def __init__(self, L, n):
        self.L, self.ev = L, [C.c_void_p() for _ in range(2 * n)]
        for e in self.ev:
            L.mi355_event_create(C.byref(e))

    def ms(self, k):
        v = C.c_float()
        self.L.mi355_event_elapsed_ms(self.ev[2 * k], self.ev[2 * k + 1], C.byref(v))
        return v.value

    def free(self):
        for e in self.ev:
            self.L.mi355_event_destroy(e)


STUCK = []            # helper threads that never came back from a collective


def run_guarded(fn, timeout_s):
    """Runs fn() on a helper thread and gives up waiting after timeout_s (a collective that never completes must not take the result line down)."""
    box = {}

    def body():
        try:
            box["value"] = fn()
        except Exception as e:                                 # noqa: BLE001
            box["error"] = repr(e)
    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        STUCK.append(th)
        return dict(error="timed out after %d s" % timeout_s)
    return box.get("value") if "value" in box else dict(error=box.get("error", "unknown"))


def spawn_ranks(n, script=None, script_args=None):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) and pass rank 0's
    line through.  Under `python -m torch.distributed.run` (WORLD_SIZE set) this is not used: the launcher has started the ranks already."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MI355_BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if script_args is None else list(script_args)), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:                                            # a rank that dies must not leave the others waiting at a barrier for ever
        time.sleep(0.2)
        for p_ in list(alive):
            r_ = p_.poll()
            if r_ is None:
                continue
            alive.remove(p_)
            if r_ != 0:
                rc = rc or r_
                deadline = time.time() + 20
                for q in alive:
                    while q.poll() is None and time.time() < deadline:
                        time.sleep(0.2)
                    if q.poll() is None:
                        q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="crown", choices=["crown", "shadow16m"])
    ap.add_argument("--rays", type=int, default=1 << 20, help="crown: rays per batch and GPU; shadow16m: hit points (x16 shadow rays), all GPUs together")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the TIMED steps are issued on round-robin = ray batches in flight (1 = one batch at a time: the metric as BASELINE.json words it, the default)")
    ap.add_argument("--pipeline-streams", type=int, default=4, help="batches in flight of the extra `pipelined` leg that is measured when --streams is 1 (0 = skip it)")
    ap.add_argument("--gather", default="auto", choices=["auto", "rccl", "none"], help="results gathered on the GPUs over RCCL (auto: rccl when more than one rank)")
    ap.add_argument("--phi", type=int, default=158, help="sphere tessellation of the synthetic crown (158 -> 4.76M triangles)")
    ap.add_argument("--config", default="", help="extra rtcNewDevice config, e.g. max_leaf=2,int_cost=0.5")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dump", default="", help="shadow16m, rank 0: save the first --dump-rays rays and their gathered result words to this .npz (tests/test_gpu_round3.py checks them against the reference)")
    ap.add_argument("--dump-rays", type=int, default=1 << 22)
    ap.add_argument("--inprocess-gpus", type=int, default=0, help="extra leg at N = 1: rtcIntersect1M through ONE RTCDevice over this many GPUs (0 = all GPUs of the node, 1 = skip; "
                                                                    "more than the node has = replicas share GPUs)")
    ap.add_argument("--scaling", default="both", choices=["weak", "both"], help="crown at N > 1: `both` (default) measures the strong-scaling leg (ONE 2^20-ray batch sharded over the N GPUs, gathered to "
                                                                               "rank 0: the literal metric) and makes it `value`, the weak figure (2^20 rays per GPU and step) stands beside it as `weak`; "
                                                                               "`weak` measures the weak leg only")
    ap.add_argument("--spin-up", type=float, default=0.5, help="seconds of the same launches between the warm-up steps and the timed region: the timed region of 20 steps is 13 ms, inside the "
                                                             "ramp of the GPU's clocks (round 5: the first four kernels 0.71 .. 0.67 ms, the last four 0.65 .. 0.63); the reference's own benchmark protocol "
                                                             "skips its start-up frames too (scripts/run-benchmark.sh:16-18).  0 = off")
    ap.add_argument("--sustain", type=float, default=6.0, help="seconds of back-to-back batches after the timed region (rank 0, N = 1): long enough for a 5-second utilisation sampler to see the GPU busy; 0 = skip")
    ap.add_argument("--scene", default="", help=".ecs / .xml / .obj scene file; default: $EMBREE_MODEL_DIR/crown/crown.ecs if it exists, "
                                                 "else the synthetic crown stand-in")
    args = ap.parse_args()
    shadow = args.workload == "shadow16m"
    if args.steps is None:
        args.steps = 10 if shadow else 60
    if args.warmup is None:
        args.warmup = 2 if shadow else 12

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # no launcher: this process becomes the launcher of N ranks (SCALE runs must never measure N = 1 N times)
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries ONE JSON line and nothing else: native libraries print there too (RCCL's version banner, Gloo's connection messages), so fd 1 points at
    # stderr for the whole run and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but %d ranks were started (WORLD_SIZE): refusing to print a line whose n_gpus is not what ran" % (args.gpus, world))
    use_rccl = args.gather == "rccl" or (args.gather == "auto" and world > 1)
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

    def barrier():
        if dist:
            dist[0].barrier()

    def max_over_ranks(x):
        if not dist:
            return float(x)
        tt = dist[1].tensor([x], dtype=dist[1].float64)
        dist[0].all_reduce(tt, op=dist[0].ReduceOp.MAX)
        return float(tt[0])

    # ---- scene: replicated BVH, geometry resident on the device
    scene_path = args.scene or loaders.find_model("crown")
    if scene_path:                                         # the real asset when the box has it (the reference does not ship it)
        loaded = loaders.load_scene(scene_path)
        meshes, camera = loaded.meshes, loaded.camera
        scene_name = "%s (%s)" % (os.path.basename(scene_path), "loaded with embree_amd/loaders.py")
    else:
        meshes, camera = W.synthetic_crown(num_phi=args.phi), None
        scene_name = "synthetic-crown (crown.ecs is not shipped)"
    ntri = W.num_triangles(meshes)
    scene = api.Scene(dev)
    for v, t in meshes:
        scene.add_triangle_mesh(v, t, device_resident=True)      # rtcSetSharedGeometryBufferHostDevice: no upload at commit
    build_ms, commit_wall = [], []
    for rep in range(1 + 5):                               # rtcCommitScene timed like buildbench (buildbench_device.cpp:385-387): 1 warm-up + 5
        scene.touch()                                      # a commit of an unmodified scene returns at once (scene.cpp:831): mark it modified
        t0 = time.perf_counter()
        scene.commit()
        if rep:
            commit_wall.append(time.perf_counter() - t0)
            build_ms.append(scene.info()["build_ms"])
    info = scene.info()
    # the same scene with RTC_BUILD_QUALITY_LOW (Morton build), reported next to the SAH build; the timed rays use the SAH tree
    low_ms = []
    L.rtcSetSceneBuildQuality(scene.h, api.RTC_BUILD_QUALITY_LOW)
    for rep in range(4):
        scene.touch()
        scene.commit()
        if rep:
            low_ms.append(scene.info()["build_ms"])
    low_info = scene.info()
    high_ms = []                                           # RTC_BUILD_QUALITY_HIGH: spatial splits inside the recursion (one launch sequence, one host round trip since round 4)
    L.rtcSetSceneBuildQuality(scene.h, api.RTC_BUILD_QUALITY_HIGH)
    for rep in range(4):
        scene.touch()
        scene.commit()
        if rep:
            high_ms.append(scene.info()["build_ms"])
    high_info = scene.info()
    L.rtcSetSceneBuildQuality(scene.h, api.RTC_BUILD_QUALITY_MEDIUM)
    scene.commit()
    assert scene.info()["num_nodes"] == info["num_nodes"]
    bvh = scene.bvh()

    # ---- rays: primary image traced on the GPU -> diffuse bounce rays
    side = int(round(args.rays ** 0.5))
    prim = (W.camera_rays(camera["vp"], camera["vi"], camera["vu"], camera["fov"], side, side) if camera
            else W.crown_camera_rays(meshes, side, side))
    dprim = api.DeviceArray.from_numpy(prim, gpu)
    scene.intersect1M_device(dprim.ptr, prim.shape[0])
    L.mi355_device_synchronize(gpu)
    traced = dprim.download(RAYHIT_DTYPE)
    dprim.free()
    if shadow:                                             # configs[3]: 16 shadow rays per hit point of the bounce rays; this rank's contiguous shard of the 16 * rays
        bounce = W.diffuse_bounce_rays(traced, meshes, seed=1)
        db = api.DeviceArray.from_numpy(bounce, gpu)
        scene.intersect1M_device(db.ptr, bounce.shape[0])
        L.mi355_device_synchronize(gpu)
        bounce = db.download(RAYHIT_DTYPE)
        db.free()
        total_rays = 16 * bounce.shape[0]
        lo, hi = shard.shard_range(total_rays, rank, world)
        assert lo % 16 == 0 and hi % 16 == 0 and (hi - lo) * world == total_rays, "the ray count must split evenly over the ranks"
        rays = W.shadow_rays(bounce[lo // 16: hi // 16], meshes, samples=16, first=lo)
        rec, dtype, any_hit = 48, RAY_DTYPE, 1
    else:
        rays = W.diffuse_bounce_rays(traced, meshes, seed=1 + rank)   # this rank's own batch (weak scaling)
        total_rays = rays.shape[0] * world
        rec, dtype, any_hit = 96, RAYHIT_DTYPE, 0
    M = rays.shape[0]

    # the leg that is NOT the timed region is measured right after it: lone launches (`serial`) when the timed steps are pipelined, and the other way round
    npipe = (1 if args.streams > 1 else max(0, args.pipeline_streams)) if not shadow else 0
    streams = []
    for _ in range(max(1, args.streams, npipe)):
        st_ = C.c_void_p()
        L.mi355_stream_create(gpu, C.byref(st_))
        streams.append(st_)
    tstreams = streams[:max(1, args.streams)]
    nbuf = args.steps + args.warmup
    pristine = api.DeviceArray.from_numpy(rays, gpu)
    bufs = [api.DeviceArray(rays.nbytes, gpu) for _ in range(nbuf)]

    def restore(which):
        for b in which:
            L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, streams[0])
        L.mi355_synchronize(streams[0])
    restore(bufs)

    # ---- visit counts for the algorithmic-bytes figure (counting build of the same kernel, same rays)
    dstat = api.DeviceArray.from_numpy(rays, gpu)
    st = scene.trace_stats(dstat.ptr, M, rec, any_hit=bool(any_hit))
    result = dstat.download(dtype)
    dstat.free()
    if shadow:
        nhit = int(np.isneginf(result["tfar"]).sum())
        alg_bytes = M * 48 + M * 4 + st["nodes"] * 80 + st["tris"] * 48
    else:
        nhit = int((result["geomID"] != INVALID_ID).sum())
        alg_bytes = M * 48 + nhit * 52 + st["nodes"] * 80 + st["tris"] * 48

    # ---- RCCL: communicator + result buffers (shadow16m: used inside the timed region; crown: exercised after it)
    comm, comm_err = None, None
    pack_bytes = M * (4 if shadow else 32)
    if use_rccl:
        res = run_guarded(lambda: shard.Communicator(gpu, rank, world, dist[0] if dist else None), 120)
        if isinstance(res, dict):
            comm_err = res["error"]
            log("rank %d: RCCL communicator unavailable (%s): results stay in the per-rank buffers" % (rank, comm_err))
        else:
            comm = res
        ok = 0.0 if comm is None else 1.0
        if dist:                                           # every rank must agree before a collective is issued
            tt = dist[1].tensor([ok], dtype=dist[1].float64)
            dist[0].all_reduce(tt, op=dist[0].ReduceOp.MIN)
            ok = float(tt[0])
        if ok < 1.0 and comm is not None:
            comm.close()
            comm, comm_err = None, comm_err or "another rank has no communicator"
    gather_in_step = comm is not None and (shadow or world > 1 or args.gather == "rccl")
    comm_stream = None
    if gather_in_step:
        cs = C.c_void_p()
        assert L.mi355_stream_create(gpu, C.byref(cs)) == 0, L.mi355_last_error()
        comm_stream = cs

    class Stepper:
        """One pass of the hot path over one batch of `m` rays: trace; with a communicator also pack the written fields and gather them over RCCL -- the gather of THIS
        batch runs on the communication stream and overlaps the trace of the NEXT one (the trace kernels themselves still run one at a time on `stream`).  Two result
        buffers: the pack of batch k + 2 waits for the gather of batch k.  Every gather is bracketed by events on the communication stream (gather_ms)."""

        def __init__(self, m, max_steps):
            self.m, self.pack_bytes, self.step_no, self.gather_ev = m, m * (4 if shadow else 32), 0, []
            self.packed = self.gathered = self.ev_packed = self.ev_gathered = None
            if gather_in_step:
                self.packed = [api.DeviceArray(self.pack_bytes, gpu) for _ in range(2)]
                self.gathered = [api.DeviceArray(self.pack_bytes * world if (shadow or rank == 0) else 16, gpu) for _ in range(2)]
                self.ev_packed, self.ev_gathered = [C.c_void_p() for _ in range(2)], [C.c_void_p() for _ in range(2)]
                for e in self.ev_packed + self.ev_gathered:
                    assert L.mi355_event_create(C.byref(e)) == 0
                self.pool = [C.c_void_p() for _ in range(2 * max_steps)]
                for e in self.pool:
                    assert L.mi355_event_create(C.byref(e)) == 0

        def collective(self, k):
            if shadow:
                comm.allgather(self.packed[k].ptr, self.gathered[k].ptr, self.pack_bytes, comm_stream)
            else:
                comm.gather(self.packed[k].ptr, self.gathered[k].ptr, self.pack_bytes, 0, comm_stream)

        def __call__(self, buf, stream, ev_a=None, ev_b=None, timed=False):
            rc = L.mi355_trace_timed(bvh, buf.ptr, self.m, rec, any_hit, stream, ev_a, ev_b)
            assert rc == 0, L.mi355_last_error()
            if gather_in_step:
                k = self.step_no & 1
                if self.step_no >= 2:                            # the result buffer of batch k - 2 must have left before it is packed over
                    assert L.mi355_stream_wait_event(stream, self.ev_gathered[k]) == 0
                pack = L.mi355_pack_occluded if shadow else L.mi355_pack_hits
                assert pack(buf.ptr, self.m, rec, self.packed[k].ptr, stream) == 0, L.mi355_last_error()
                assert L.mi355_event_record(self.ev_packed[k], stream) == 0
                assert L.mi355_stream_wait_event(comm_stream, self.ev_packed[k]) == 0
                g0 = g1 = None
                if timed and 2 * len(self.gather_ev) + 1 < len(self.pool):
                    g0, g1 = self.pool[2 * len(self.gather_ev)], self.pool[2 * len(self.gather_ev) + 1]
                    self.gather_ev.append((g0, g1))
                    assert L.mi355_event_record(g0, comm_stream) == 0
                self.collective(k)
                if g1 is not None:
                    assert L.mi355_event_record(g1, comm_stream) == 0
                assert L.mi355_event_record(self.ev_gathered[k], comm_stream) == 0
                self.step_no += 1

        def gather_ms(self):
            """milliseconds each timed collective spent on the communication stream (from the moment its packed block was ready: includes waiting for the slowest peer)"""
            out_ = []
            for g0, g1 in self.gather_ev:
                v = C.c_float()
                L.mi355_event_elapsed_ms(g0, g1, C.byref(v))
                out_.append(v.value)
            self.gather_ev = []
            return out_

    step = Stepper(M, 2 * (args.steps + args.warmup) + 64)
    if gather_in_step:
        # one collective on trial before anything is timed: a gather that never completes (a link that does not come up) must cost the line its gather, not the line
        def trial():
            step.collective(0)
            deadline = time.time() + 90
            while L.mi355_stream_query(comm_stream) == 1:
                if time.time() > deadline:
                    return dict(error="the trial collective did not complete within 90 s")
                time.sleep(0.002)
            return True
        res = run_guarded(trial, 120)
        ok = 1.0 if res is True else 0.0
        if dist:
            tt = dist[1].tensor([ok], dtype=dist[1].float64)
            dist[0].all_reduce(tt, op=dist[0].ReduceOp.MIN)
            ok = float(tt[0])
        if ok < 1.0:
            comm_err = (res.get("error") if isinstance(res, dict) else None) or "the trial collective failed on another rank"
            log("rank %d: %s: results stay in the per-rank buffers" % (rank, comm_err))
            gather_in_step, comm = False, None                # (the communicator is left alone: destroying one with a collective in flight can hang as well)

    for st_ in streams:                                     # per-stream traversal scratch exists before anything is timed (also when --warmup 0)
        assert L.mi355_trace_prepare(bvh, st_) == 0, L.mi355_last_error()
    for i in range(args.warmup):
        step(bufs[i], tstreams[i % len(tstreams)])
    L.mi355_device_synchronize(gpu)
    # the clocks: the same launches (restore + trace of a batch of their own) for --spin-up seconds, so that the K timed steps run at the clock the GPU sustains
    spin_launches = 0
    if args.spin_up > 0:
        spin_buf = api.DeviceArray(rays.nbytes, gpu)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spin_up:
            for _ in range(8):
                L.mi355_memcpy_d2d_async(spin_buf.ptr, pristine.ptr, rays.nbytes, tstreams[0])
                rc_ = L.mi355_trace_timed(bvh, spin_buf.ptr, M, rec, any_hit, tstreams[0], None, None)
                assert rc_ == 0, L.mi355_last_error()
            L.mi355_synchronize(tstreams[0])
            spin_launches += 8
        spin_buf.free()

    # ================================================================================================ the timed region
    ev = Events(L, args.steps)
    barrier()
    L.mi355_device_synchronize(gpu)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(bufs[args.warmup + k], tstreams[k % len(tstreams)], ev.ev[2 * k], ev.ev[2 * k + 1], timed=True)
    L.mi355_device_synchronize(gpu)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    # ================================================================================================
    kernel_ms = [ev.ms(k) for k in range(args.steps)]
    ev.free()
    gather_ms = step.gather_ms() if gather_in_step else []
    for st_ in tstreams:
        assert scene.trace_status(st_) == 0, "a traversal safety net dropped work"
    # every timed buffer must hold the same answer as the counting run (same rays, same tree)
    for b in (bufs[args.warmup], bufs[-1]):
        assert b.download(dtype).tobytes() == result.tobytes(), "timed kernel and counting kernel disagree"
    gather_check = None
    if gather_in_step:
        L.mi355_synchronize(comm_stream)
        last = (step.step_no - 1) & 1                      # the result buffer the last timed batch was gathered into
        gathered = step.gathered
        if shadow:                                         # every rank holds all shards: rank r's part must be what rank r computed
            g = gathered[last].download(np.uint32).reshape(world, M)
            assert (g[rank] == result["tfar"].view(np.uint32)).all(), "gathered shard differs from the local result"
            occl = np.array([int((g[r] == 0xFF800000).sum()) for r in range(world)], np.int64)
            mine = dist[1].tensor([int(np.isneginf(result["tfar"]).sum())], dtype=dist[1].int64) if dist else None
            if dist:
                allc = [dist[1].zeros(1, dtype=dist[1].int64) for _ in range(world)]
                dist[0].all_gather(allc, mine)
                assert [int(a[0]) for a in allc] == occl.tolist(), "gathered occlusion counts differ from what the ranks computed"
            gather_check = dict(transport="RCCL ncclAllGather over xGMI, on a communication stream: the gather of batch k overlaps the trace of batch k + 1", inside_timed_region=True,
                                bytes_total=pack_bytes * world, occluded_per_rank=occl.tolist())
            if args.dump and rank == 0:
                k = min(args.dump_rays, M)
                np.savez(args.dump, rays=rays[:k], gathered=g.reshape(-1)[:k])
        else:                                              # the root holds every rank's packed hit records: rank r's block must be what rank r computed
            sums = np.array([int(result["primID"].astype(np.uint64).sum()), int(result["tfar"].view(np.uint32).astype(np.uint64).sum()),
                             int((result["geomID"] != INVALID_ID).sum())], np.int64)
            allsums = [sums]
            if dist:
                tl = [dist[1].zeros(3, dtype=dist[1].int64) for _ in range(world)]
                dist[0].all_gather(tl, dist[1].from_numpy(sums.copy()))
                allsums = [t.numpy() for t in tl]
            if rank == 0:
                g = gathered[last].download(np.uint32).reshape(world, M, 8)          # { tfar, u, v, primID | geomID, Ng }
                assert (g[0][:, 0] == result["tfar"].view(np.uint32)).all() and (g[0][:, 3] == result["primID"]).all() and (g[0][:, 4] == result["geomID"]).all(), "root's own block differs from its result"
                for r in range(world):
                    got = np.array([int(g[r][:, 3].astype(np.uint64).sum()), int(g[r][:, 0].astype(np.uint64).sum()), int((g[r][:, 4] != INVALID_ID).sum())], np.int64)
                    assert (got == allsums[r]).all(), "gathered block of rank %d differs from what that rank computed" % r
            gather_check = dict(transport="RCCL ncclGather of the packed hit records (32 B per ray) to rank 0 over xGMI, on a communication stream: the gather of batch k overlaps the trace "
                                          "of batch k + 1", inside_timed_region=True, bytes_per_rank=pack_bytes, bytes_into_root_per_step=pack_bytes * (world - 1),
                                checked="every rank's block on the root against that rank's own checksums (primID sum, tfar-bits sum, hit count)")

    # ---- N > 1, crown: the other reading of "1M incoherent diffuse rays at 1/2/4/8 GPUs" -- ONE 2^20-ray batch sharded N ways (strong scaling), gather to rank 0 in the step
    strong = None
    if not shadow and world > 1 and args.scaling == "both":
        all_rays = W.diffuse_bounce_rays(traced, meshes, seed=1)          # rank 0's batch, the same on every rank (seeded)
        Ms = all_rays.shape[0] // world                                  # (equal shards: RCCL's gather takes one size; a remainder below N rays is dropped and named)
        shard_rays = all_rays[rank * Ms:(rank + 1) * Ms].copy()
        sp_ = api.DeviceArray.from_numpy(shard_rays, gpu)
        nst = args.steps
        sbufs = [api.DeviceArray(shard_rays.nbytes, gpu) for _ in range(nst + 4)]
        for b in sbufs:
            L.mi355_memcpy_d2d_async(b.ptr, sp_.ptr, shard_rays.nbytes, streams[0])
        L.mi355_synchronize(streams[0])
        sstep = Stepper(Ms, 2 * (nst + 4) + 8)
        for i in range(4):
            sstep(sbufs[nst + i], tstreams[0])
        L.mi355_device_synchronize(gpu)
        sev = Events(L, nst)
        barrier()
        L.mi355_device_synchronize(gpu)
        t1 = time.perf_counter()
        for k in range(nst):
            sstep(sbufs[k], tstreams[0], sev.ev[2 * k], sev.ev[2 * k + 1], timed=True)
        L.mi355_device_synchronize(gpu)
        barrier()
        sel = max_over_ranks(time.perf_counter() - t1)
        sk = [sev.ms(k) for k in range(nst)]
        sev.free()
        sg = sstep.gather_ms() if gather_in_step else []
        want_ = shard_rays.copy()                                        # this rank's shard must be what a lone trace of the same rays gives (same tree on every rank)
        chk = api.DeviceArray.from_numpy(want_, gpu)
        scene.intersect1M_device(chk.ptr, Ms)
        L.mi355_device_synchronize(gpu)
        assert sbufs[0].download(dtype).tobytes() == chk.download(dtype).tobytes(), "strong-scaling shard differs from a lone trace of the same rays"
        chk.free()
        strong = dict(value=round(Ms * world * nst / sel / 1e6, 2), unit="Mrays/s", scaling="strong", rays_total=Ms * world, rays_per_gpu=Ms, steps=nst, ms_per_step=round(1e3 * sel / nst, 4),
                      trace_ms=round(float(np.mean(sk)), 4), gather_ms=round(float(np.mean(sg)), 4) if sg else None,
                      gather_bound=bool(sg and float(np.mean(sg)) > 0.9 * (1e3 * sel / nst)) if sg else None,
                      what="ONE batch of %d incoherent diffuse rays cut into %d contiguous shards, every shard traced on its own GPU, the packed hit records gathered to rank 0 over RCCL inside "
                           "the step (the gather of batch k overlaps the trace of batch k + 1); whole-job rays per second, max over ranks" % (Ms * world, world))
        for b in sbufs:
            b.free()
        sp_.free()

    # ---- a leg long enough for an outside sampler (the driver polls utilisation every few seconds; the timed region is 15 ms): the same batches back to back for --sustain seconds
    sustained = None
    if not shadow and world == 1 and rank == 0 and args.sustain > 0:
        nsu = min(len(bufs), 32)
        t1 = time.perf_counter()
        launches_ = 0
        while time.perf_counter() - t1 < args.sustain:
            for b in bufs[:nsu]:
                L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, tstreams[0])
                assert L.mi355_trace_closest(bvh, b.ptr, M, rec, tstreams[0]) == 0, L.mi355_last_error()
            L.mi355_synchronize(tstreams[0])
            launches_ += nsu
        sdt = time.perf_counter() - t1
        assert bufs[0].download(dtype).tobytes() == result.tobytes()
        sustained = dict(seconds=round(sdt, 2), launches=launches_, value=round(M * launches_ / sdt / 1e6, 1), unit="Mrays/s",
                         what="the timed batches again, back to back on one stream for %.0f s, each behind a device-to-device copy that restores its rays (the copies are inside this figure): "
                              "not the metric -- a leg long enough for a utilisation sampler to see the GPU at work" % args.sustain)

    # ---- small batches (one 2^20-ray batch over 8 GPUs is 2^17 rays per launch): lone launches over the first 2^17 / 2^15 rays of the batch
    small_batch = None
    if not shadow and rank == 0 and world == 1 and M >= (1 << 17):
        small_batch = []
        for nsm in (1 << 17, 1 << 15):
            evs = Events(L, 24)
            sb_ = api.DeviceArray(nsm * rec, gpu)
            for k in range(24 + 4):
                L.mi355_memcpy_d2d_async(sb_.ptr, pristine.ptr, nsm * rec, tstreams[0])
                e0_, e1_ = (evs.ev[2 * (k - 4)], evs.ev[2 * (k - 4) + 1]) if k >= 4 else (None, None)
                assert L.mi355_trace_timed(bvh, sb_.ptr, nsm, rec, any_hit, tstreams[0], e0_, e1_) == 0, L.mi355_last_error()
            L.mi355_synchronize(tstreams[0])
            sms = np.array([evs.ms(k) for k in range(24)])
            evs.free()
            assert sb_.download(dtype, nsm).tobytes() == result[:nsm].tobytes(), "a small batch and the full batch disagree on the same rays"
            sb_.free()
            small_batch.append(dict(rays=nsm, us=round(1e3 * float(np.median(sms)), 1), us_min=round(1e3 * float(sms.min()), 1), mrays=round(nsm / float(np.median(sms)) / 1e3, 1)))
    # ---- extra legs, outside the timed region --------------------------------------------------------------------------------------
    pipelined = None
    if npipe >= 1 and npipe != len(tstreams) and not shadow:   # the other mode: several batches in flight as a wavefront renderer keeps them / one batch at a time
        nps = min(args.steps, 40)
        restore(bufs[:nps])
        evp = Events(L, nps)
        for i in range(min(8, nps)):                       # warm the extra streams
            step(bufs[i], streams[i % npipe])
        L.mi355_device_synchronize(gpu)
        restore(bufs[:nps])
        t1 = time.perf_counter()
        for k in range(nps):
            step(bufs[k], streams[k % npipe], evp.ev[2 * k], evp.ev[2 * k + 1])
        L.mi355_device_synchronize(gpu)
        pel = time.perf_counter() - t1
        pms = [evp.ms(k) for k in range(nps)]
        evp.free()
        assert bufs[0].download(dtype).tobytes() == result.tobytes()
        pavg = float(np.mean(pms))
        pipelined = dict(value=round(M * nps / pel / 1e6, 1), unit="Mrays/s", batches_in_flight=npipe, steps=nps, ms_per_step=round(1e3 * pel / nps, 4),
                         kernel_ms_avg=round(pavg, 4), kernel_ms_min=round(float(np.min(pms)), 4), concurrency=round(float(np.sum(pms)) * 1e-3 / pel, 3),
                         hbm_algorithmic=({"achieved": round(alg_bytes / (pavg * 1e-3) / 1e9, 1), "frac": round(alg_bytes / (pavg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "how": "algorithmic bytes per launch / average duration of the lone launches"} if npipe == 1 else
                                          {"achieved": round(alg_bytes * nps / pel / 1e9, 1), "frac": round(alg_bytes * nps / pel / 1e9 / HBM_PEAK_GBS, 4), "how": "algorithmic bytes x launches / elapsed"}),
                         what=("the same launches issued round-robin on %d HIP streams: the tail of one batch overlaps the next; per-GPU figure" % npipe) if npipe > 1 else
                              "the same launches back to back on ONE stream: one 2^20-ray batch at a time, the kernel's own duration is the step time; per-GPU figure")
    e2e = None
    if not shadow and rank == 0:                           # PCIe-inclusive: the blocking host-array entry point on a pageable numpy array
        times = []
        for _ in range(9):                                 # (the box's share of its host CPUs moves a call by +- 20 % from one moment to the next: best AND median)
            h = rays.copy()
            t1 = time.perf_counter()
            scene.intersect1M(h)
            times.append(time.perf_counter() - t1)
        assert h.tobytes() == result.tobytes()
        link = (C.c_double * 3)()
        link_ok = L.mi355_measure_host_link(gpu, rays.nbytes, 3, link) == 0 and link[2] > 0
        e2e = dict(value=round(M / min(times) / 1e6, 1), unit="Mrays/s", ms=round(1e3 * min(times), 3), ms_median=round(1e3 * float(np.median(times)), 3),
                   link_floor_ms=round(link[2], 3) if link_ok else None, frac_of_link_floor=round(link[2] / (1e3 * min(times)), 3) if link_ok else None,
                   link_GBs={"upload": round(link[0], 1), "download": round(link[1], 1)} if link_ok else None,
                   what="rtcIntersect1M on a pageable host array of %d RTCRayHit, pipelined in chunks, best of 9.  What crosses the link is what the kernels read and write (round 6, config key packed_link, default): the CPU packs the 48-byte ray part of every record into the library's pinned staging, 48 MB go up, the records are rebuilt on the GPU, the traversal runs, "
                        "the fields it wrote are packed on the GPU (32 bytes per ray), 32 MB come down, and the CPU writes them into the caller's records of the rays that hit (a miss leaves the caller's record alone, as the reference does) -- 80 MB instead of the 192 MB of whole records both ways (packed_link=0).  The GPU never maps the "
                        "caller's pages (profiles/r06_host_memory_fault.md; host_register=1 restores round 5's registration of the caller's array).  link_floor_ms = the bytes of the WHOLE array up from and down to PINNED host memory, both directions at once and nothing else (mi355_measure_host_link): what the host link "
                        "allows a copy of the records" % M)
    latency = None
    if not shadow and rank == 0:                           # the per-ray entry points are "not the measured path and the report must say so": what one call costs
        one = rays[:1].copy()
        for _ in range(20):
            scene.intersect1(one.copy())
        ts = []
        for i in range(200):
            r1 = rays[i:i + 1].copy()
            t1 = time.perf_counter()
            scene.intersect1(r1)
            ts.append(time.perf_counter() - t1)
        latency = dict(rtcIntersect1_us_median=round(1e6 * float(np.median(ts)), 1), rtcIntersect1_us_min=round(1e6 * float(np.min(ts)), 1),
                       calls=200, what="one blocking rtcIntersect1 call on a host RTCRayHit (H2D copy + one-wave launch + D2H copy + status read), "
                                       "ctypes call overhead included: the Embree 4 per-ray API works but is not how a GPU is fed -- the batched calls are the measured path")
    multi = None
    if not shadow and rank == 0 and world == 1 and args.inprocess_gpus != 1:
        # ONE process, one RTCDevice over K GPUs (rtcNewDevice("gpus=K")): the BVH committed on every GPU, the host ray array sharded over them by rtcIntersect1M
        k_gpus = args.inprocess_gpus if args.inprocess_gpus > 0 else ngpu
        if k_gpus > 1:
            def multi_leg():                                  # (guarded: a device that does not answer on a node never seen before must cost the line this leg, not the line)
                mdev = api.Device(("gpu=%d,gpus=%d,%s" % (gpu, k_gpus, "gpu_oversubscribe=1," if k_gpus > ngpu else "")) + args.config)
                msc = api.Scene(mdev)
                for v, t in meshes:
                    msc.add_triangle_mesh(v, t)
                t1 = time.perf_counter()
                msc.commit()
                commit_s = time.perf_counter() - t1
                times = []
                for _ in range(3):
                    h = rays.copy()
                    t1 = time.perf_counter()
                    msc.intersect1M(h)
                    times.append(time.perf_counter() - t1)
                assert h.tobytes() == result.tobytes(), "the sharded in-process query disagrees with the single-GPU answer"
                out_ = dict(gpus=k_gpus, distinct_gpus=min(k_gpus, ngpu), value=round(M / min(times) / 1e6, 1), unit="Mrays/s", ms=round(1e3 * min(times), 3),
                             commit_all_replicas_ms=round(1e3 * commit_s, 2),
                             what="rtcNewDevice(\"gpus=%d\"): one process, the tree committed on every GPU, rtcIntersect1M on a pageable host array of %d RTCRayHit split contiguously over the "
                                  "replicas (one host thread per GPU, results copied straight into the caller's array); PCIe-inclusive; result identical to the single-GPU answer" % (k_gpus, M))
                msc.release()
                mdev.release()
                return out_
            multi = run_guarded(multi_leg, 240)
    gather = None
    if rank == 0:
        avg_ms = float(np.mean(kernel_ms))
        conc = float(np.sum(kernel_ms)) * 1e-3 / elapsed
        value = total_rays * args.steps / elapsed / 1e6 if shadow else world * M * args.steps / elapsed / 1e6
        # achieved = algorithmic bytes / kernel time.  With several launches in flight the launches OVERLAP, so "kernel time" is the timed
        # region itself: bytes of all timed launches / elapsed (chip level, this rank's GPU).  Dividing by the per-launch duration of overlapping launches
        # would count every moment of the region launches_in_flight times; the lone-launch figure (duration = kernel time) stands beside it under `serial`.
        achieved = alg_bytes * args.steps / elapsed / 1e9
        bw = (C.c_double * 2)()
        bw_ok = L.mi355_measure_bandwidth(gpu, 2 << 30, 5, bw) == 0          # what a streaming copy / read kernel reaches on THIS box ("also measure")
        pmc, pmc_note = load_pmc() if not shadow else (None, "PMC passes are collected for the closest-hit kernel only")
        hbm_alg = {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                   "is": "SURVEY 8(d)'s figure: algorithmic bytes x launches / elapsed against the HBM peak.  NOT a bound and not HBM traffic: it charges every re-visit of a node, and L1 / L2 / the "
                         "Infinity Cache serve most of them (a value above 1 means exactly that)",
                   "peak_measured": {"copy_GBs": round(bw[0], 1), "read_GBs": round(bw[1], 1), "frac_of_copy": round(achieved / bw[0], 4) if bw_ok and bw[0] > 0 else None,
                                     "what": "mi355_measure_bandwidth on this GPU: device-to-device copy (bytes read + written) and read-only streaming kernels over 2 GiB, best of 5"} if bw_ok else None}
        # the bytes a launch cannot avoid: every DISTINCT node and triangle record it fetches once (a bit per record, set by the counting kernel), its rays in, its hit records out
        compulsory = int(st["unique_nodes"] * 80 + st["unique_tris"] * 48 + M * 48 + (M * 4 if shadow else nhit * 52)) if st.get("unique_nodes") else None
        roof = {"bound": "valu", "bound_is": "VALU instruction issue binds this kernel, not bytes (VERDICT r05): frac = the share of the timed region in which the 1024 SIMDs' vector pipes issue",
                "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None,
                "frac_algorithmic_cache_served": hbm_alg["frac"], "hbm_algorithmic": hbm_alg,
                "traffic": int(pmc["hbm_traffic_bytes_per_launch"]) if pmc and "hbm_traffic_bytes_per_launch" in pmc else None,
                "traffic_source": "profiles/pmc_bench_latest.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per lone launch of this kernel source)" if pmc else None,
                "traffic_is": "an UPPER bound: FETCH_SIZE counts 64 bytes per memory-side request whatever its size -- x 2 is right for streaming reads (128-byte requests), x 1 for lone 16-byte loads, "
                              "x 1.33 for 80-byte records at random addresses (tools/fetch_calib.hip, profiles/r05_fetch_calibration.md); this kernel's node / triangle reads are of the last two kinds" if pmc else None,
                "compulsory_bytes": compulsory,
                "compulsory_is": "distinct nodes fetched x 80 + distinct triangle records fetched x 48 (one bit per record, set by the counting kernel on the same rays) + rays x 48 read + hit records written: "
                                 "what a perfect cache in front of the memory would still move once per launch",
                "unique_nodes": int(st.get("unique_nodes", 0)), "unique_triangle_records": int(st.get("unique_tris", 0)),
                "traffic_over_compulsory": round(pmc["hbm_traffic_bytes_per_launch"] / compulsory, 2) if (pmc and compulsory and "hbm_traffic_bytes_per_launch" in pmc) else None,
                "kernel": "trace_kernel_q<%s>" % ("any" if shadow else "closest"),
                "how": "algorithmic bytes per launch x launches timed / elapsed of the timed region / peak (launches overlap: %.2f in flight); lone launches: see `serial`" % conc,
                "launches_timed": args.steps, "launches_in_flight": round(conc, 3),
                "spin_up": {"seconds": args.spin_up, "launches": spin_launches, "what": "the same launches between the warm-up steps and the timed region (not timed): the GPU's clocks are up when the K steps start"},
                "kernel_ms_avg_overlapping": round(avg_ms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "kernel_ms_first4": [round(float(x), 4) for x in kernel_ms[:4]], "kernel_ms_last4": [round(float(x), 4) for x in kernel_ms[-4:]],
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "per_ray": {"nodes": round(st["nodes"] / M, 2), "node_step_simd_util": round(st["nodes"] / max(1, 64 * st["node_blocks"]), 3),
                            "tri_step_simd_util": round(st["tris"] / max(1, 64 * st["tri_blocks"]), 3),
                            "triangles": round(st["tris"] / M, 2), "empty_node_visits": round(st["empty_nodes"] / M, 2), "stale_node_visits": round(st["culled_groups"] / M, 2),
                            "bytes": round(alg_bytes / M, 1),
                            "wave_iterations": int(st["wave_iters"]), "node_step_blocks": int(st["node_blocks"]), "tri_step_blocks": int(st["tri_blocks"]),
                            "handout_events": int(st["refill_events"]), "handout_clock_share": round(st["refill_clocks"] / max(1, st["loop_clocks"]), 4),
                            "node_step_clock_share": round(st["node_step_clocks"] / max(1, st["loop_clocks"]), 4)},
                "note": "algorithmic bytes (SURVEY 8d) = rays x (48 read + 52 written on a hit) + node visits x 80 + triangle records x 48, visit counts from the counting build of the same "
                        "kernel on the same rays. Most of these bytes are served by L1 / L2 / Infinity Cache: hbm_counter_from_profile is what reaches the memory side."}
        # what binds (profiles/r02_trace_history.md): every lane that fetches a 16-byte piece of a node / triangle / ray costs the CU's address path one slot, whatever
        # the width and whatever the cache level that answers: 5 per node visit, 3 per triangle test, 3 per ray read + the hit record stores
        acc = 5 * st["nodes"] + 3 * st["tris"] + M * 3 + (nhit * 4 if not shadow else nhit)
        roof["address_rate"] = {"lane_accesses_per_launch": int(acc), "per_ray": round(acc / M, 1), "achieved_G_per_s": round(acc * args.steps / elapsed / 1e9, 1),
                                "peak_G_per_s": round(256 * 2.4, 1), "frac": round(acc * args.steps / elapsed / (256 * 2.4e9), 4),
                                "what": "scattered lane-addresses ((lane, 16-byte load) pairs) per second over all launches in flight against 256 CUs x 1 address per clock x 2.4 GHz. "
                                        "Round 2 read this as the binding resource (an extra load per triangle test cost 11-12 %); round 4 removed 16 % of the pairs at +16 % VALU work "
                                        "(two triangles per leaf record) and lost 8 %: the kernel follows VALU issue first (profiles/r04_pair_records.md)"}
        if pmc:
            # PMC counters are per LAUNCH (lone launches of the PMC run, from profiles/); the rates below are for the whole chip over the timed region:
            # counter x launches timed / elapsed
            c = pmc["counters"]
            lone_ms = (pipelined or {}).get("kernel_ms_avg") if (len(tstreams) > 1 and pipelined) else avg_ms      # a lone launch's duration (the `serial` leg), for the clock
            clock_hz = pmc.get("kernel_cycles", 0.0) / (lone_ms * 1e-3) if (pmc.get("kernel_cycles") and lone_ms) else 2.4e9
            clock_hz = min(max(clock_hz, 1.0e9), 2.4e9)
            hbm_rate = pmc["hbm_traffic_bytes_per_launch"] * args.steps / elapsed / 1e9
            roof["hbm_counter_frac"] = round(hbm_rate / HBM_PEAK_GBS, 4)
            roof["hbm_counter_from_profile"] = {"bytes_per_launch": int(pmc["hbm_traffic_bytes_per_launch"]), "achieved": round(hbm_rate, 1), "frac": round(hbm_rate / HBM_PEAK_GBS, 4),
                                                "frac_of_copy": round(hbm_rate / bw[0], 4) if bw_ok and bw[0] > 0 else None,
                                                "what": "rocprofv3 --pmc FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE per launch of this kernel source (profiles/pmc_bench_latest.json, hash %s) x launches timed / elapsed; includes Infinity-Cache hits" % pmc["source_hash"]}
            if "SQ_INSTS_VALU" in c:
                valu_s = c["SQ_INSTS_VALU"] * 4.0 / (NUM_SIMDS * clock_hz)
                roof["achieved"] = round(c["SQ_INSTS_VALU"] * args.steps / elapsed / 1e9, 2)
                roof["peak"] = round(NUM_SIMDS * clock_hz / 4.0 / 1e9, 2)
                roof["frac"] = round(valu_s * args.steps / elapsed, 4)
                roof["frac_is"] = ("wave instructions on the vector pipes per second (SQ_INSTS_VALU per launch, profiles/pmc_bench_latest.json, x launches timed / elapsed) against 1024 SIMDs x clock / 4 cycles "
                                   "per instruction: the BINDING resource of this kernel.  The SURVEY 8(d) bytes-over-HBM-peak figure is frac_algorithmic_cache_served")
                roof["valu_from_profile"] = {"wave_instructions_per_launch": int(c["SQ_INSTS_VALU"]), "cycles_per_instruction": 4, "clock_ghz": round(clock_hz / 1e9, 3),
                                             "issue_ms": round(valu_s * 1e3, 4), "frac": round(valu_s * args.steps / elapsed, 4),
                                             "what": "SQ_INSTS_VALU (profiles/pmc_bench_latest.json) x 4 cycles / (1024 SIMDs x clock) per launch x launches timed / elapsed: the share of the timed region in which the VALU pipes issue "
                                                     "(~4 cycles per wave instruction: tools/valu_bench.hip; clock = kernel cycles of the PMC run / duration of a lone launch, capped at 2.4 GHz)"}
        else:
            roof["pmc_note"] = pmc_note
        if roof["frac"] is None:                              # no counters for this build of the kernel: the only live figure is the algorithmic one -- say so instead of leaving the field empty
            roof.update({"bound": "hbm", "achieved": hbm_alg["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_alg["frac"],
                         "frac_is": "NO PMC profile for this kernel source (%s): this is SURVEY 8(d)'s algorithmic-bytes figure against the HBM peak -- cache-served bytes, not the binding resource "
                                    "(VALU issue: ~0.71 - 0.76 of the region in the profiles of rounds 5 and 6)" % (pmc_note or "none")})
        # rtcCommitScene against ITS roofline: bytes the build algorithm has to move -- vertices + indices in and references out (primref_gen), every level of
        # the binary binned-SAH build reads the references once to bin them and once to partition them and writes them once (32 B each way), the wide nodes and the
        # leaf records are written once, the leaf records gather their vertices again -- over the GPU time of the commit.  Levels = log2(leaves): what a balanced tree needs.
        b_levels = float(np.log2(max(2, info["num_leaves"])))
        b_bytes = ntri * (48 + 32) + b_levels * ntri * 96.0 + info["num_nodes"] * 80 + ntri * (48 + 48)
        b_s = float(np.min(build_ms)) * 1e-3
        try:                                                   # the commit's HBM-side bytes as the counters see them (tools/commit_traffic.py over PMC passes of tests/gpu_build_only.py)
            ct = json.load(open(os.path.join(ROOT, "profiles", "r06_commit_traffic.json")))
        except Exception:
            ct = None
        build_roof = {"bound": "hbm on paper; measured: LDS atomics and wave-instruction issue of the many small sets (profiles/r06_pmc_small_build.md: VALU pipes 0.61 busy, LDS bank-conflict ratio 0.51)", "algorithmic_bytes": int(b_bytes),
                      "achieved": round(b_bytes / b_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b_bytes / b_s / 1e9 / HBM_PEAK_GBS, 4),
                      "frac_of_copy": round(b_bytes / b_s / 1e9 / bw[0], 4) if bw_ok and bw[0] > 0 else None, "dominant_kernel": "small_build",
                      "how": "triangles x (48 read + 32 written) + log2(leaves) = %.1f levels x triangles x (32 binned + 32 read + 32 written) + nodes x 80 + triangles x (48 gathered + 48 written), "
                             "over the fastest of the timed commits" % b_levels,
                      "traffic": int(ct["traffic_bytes_per_commit"]) if ct else None,
                      "traffic_frac": round(ct["traffic_bytes_per_commit"] / b_s / 1e9 / HBM_PEAK_GBS, 4) if ct else None,
                      "traffic_is": ("FETCH_SIZE x 2 + WRITE_SIZE summed over every kernel of a commit of this scene (profiles/r06_commit_traffic.md: %.2f GB read, %.2f GB written; top_partition + top_bin "
                                     "are 5.9 GB of it, at 3.4 - 5.3 TB/s each) over this run's commit time: the formula and the counters agree to 2 %% -- what keeps the commit from the HBM roof is not "
                                     "re-read bytes but the kernels that do not stream (small_build: 1.46 GB in 1.47 ms)" % (ct["read_bytes_per_commit"] / 1e9, ct["write_bytes_per_commit"] / 1e9)) if ct else None}
        weak_value, weak_ms = value, 1e3 * elapsed / args.steps
        strong_is_value = strong is not None                     # N > 1, crown: the LITERAL metric is one 2^20-ray batch over the N GPUs
        out = {
            "metric": ("Mrays/s (shadow rays, any-hit) on crown, 16 Mi rays sharded" if shadow else
                       "Mrays/s (incoherent diffuse, closest-hit) on crown" + (", %d batches of 2^20 rays in flight (one batch at a time: serial.value)" % len(tstreams) if len(tstreams) > 1 else "")
                       + ((", ONE batch of 2^20 rays cut over the %d GPUs (strong scaling; every GPU its own 2^20 rays: `weak`)" % world) if strong_is_value else "")
                       + ((", hits packed and gathered to rank 0 over RCCL inside the step" if gather_in_step else ", NO gather (RCCL communicator unavailable)") if world > 1 else "")),
            "value": strong["value"] if strong_is_value else round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "ranks": world, "distinct_gpus": min(world, ngpu), "rccl_ranks": (world if comm is not None else 0), "steps": args.steps, "warmup": args.warmup, "ms_per_step": strong["ms_per_step"] if strong_is_value else round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if (shadow or strong_is_value) else "weak",
            "scaling_is": ("strong: the 16 Mi shadow rays of configs[3] are one job cut over the ranks" if shadow else
                           "strong: BASELINE.json words the metric on 1M rays at 1/2/4/8 GPUs -- ONE 2^20-ray batch cut into N contiguous shards, gathered to rank 0 inside the step (K steps, barrier + "
                           "synchronize on both sides, max over ranks).  `weak` is the other reading: every GPU its own 2^20-ray batch per step; the roofline block describes THAT launch (2^20 rays "
                           "on one GPU)" if strong_is_value else
                           "weak: every GPU traces its own 2^20-ray batch per step (value = all of them per second); --scaling both makes the strong figure the value" if world > 1 else
                           "N = 1 (both readings coincide)"), "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not scene_path else "file",
            "config": {"workload": ("configs[3]: %s, %d triangles, %d shadow rays (16 per hit point) in total, %d per GPU, rtcOccluded1MDevice, rays + BVH resident in HBM, results %s"
                                    % (scene_name, ntri, total_rays, M, "packed and all-gathered over RCCL" if comm is not None else "left in the per-rank buffers")) if shadow else
                                   ("configs[2]: %s, %d triangles, %d geometries, %d incoherent diffuse-bounce rays per GPU and step, closest-hit, rays + BVH resident in HBM, %s"
                                    % (scene_name, ntri, len(meshes), M, "one batch at a time" if len(tstreams) == 1 else "%d batches in flight" % len(tstreams))),
                       "rays_per_gpu": M, "triangles": ntri, "batches_in_flight": len(tstreams),
                       "parallelism": "rays sharded x%d, BVH replicated (deterministic build on every rank), %s" % (world, ("RCCL all-gather of the 4-byte results" if shadow else "RCCL gather of the packed 32-byte hit records to rank 0") + " inside the step, on a communication stream (overlaps the next batch's trace)" if gather_in_step else "no collective inside the step"),
                       "device_config": args.config},
            "roofline": roof,
            "build": {"metric": "BVH build Mprims/s", "roofline": build_roof, "gpu_build_ms": round(float(np.min(build_ms)), 3), "gpu_build_ms_median": round(float(np.median(build_ms)), 3),
                      "mprims_per_s_gpu": round(ntri / (float(np.min(build_ms)) * 1e-3) / 1e6, 1),
                      "commit_wall_ms": round(1e3 * float(np.min(commit_wall)), 3), "commit_wall_ms_median": round(1e3 * float(np.median(commit_wall)), 3),
                      "mprims_per_s_commit": round(ntri / float(np.min(commit_wall)) / 1e6, 1), "reps": "1 warm-up + %d timed (buildbench_device.cpp:385-387)" % len(build_ms),
                      "kernel_launches": info.get("num_launches"), "host_syncs": info.get("num_host_syncs"),
                      "nodes": info["num_nodes"], "leaves": info["num_leaves"], "sah": round(info["sah"], 3),
                      "bvh_bytes": info["bytes_nodes"] + info["bytes_triangles"],
                      "low_quality": {"what": "RTC_BUILD_QUALITY_LOW: Morton-code build, same node/leaf layout", "gpu_build_ms": round(float(np.min(low_ms)), 3),
                                      "mprims_per_s_gpu": round(ntri / (float(np.min(low_ms)) * 1e-3) / 1e6, 1), "sah": round(low_info["sah"], 3)},
                      "high_quality": {"what": "RTC_BUILD_QUALITY_HIGH: spatial splits inside the recursion (the reference's default form of HIGH), enqueued as one launch sequence like the default quality",
                                       "gpu_build_ms": round(float(np.min(high_ms)), 3), "mprims_per_s_gpu": round(ntri / (float(np.min(high_ms)) * 1e-3) / 1e6, 1), "sah": round(high_info["sah"], 3),
                                       "references_added": int(high_info["num_presplit"]), "host_syncs": high_info.get("num_host_syncs")}},
            "hit_fraction": round(nhit / M, 4),
        }
        if pipelined:
            out["pipelined" if npipe > 1 else "serial"] = pipelined
        if strong:
            out["strong"] = strong
            out["weak"] = dict(value=round(weak_value, 2), unit="Mrays/s", scaling="weak", rays_per_gpu=M, steps=args.steps, ms_per_step=round(weak_ms, 4),
                               what="every GPU traces its own 2^20-ray batch per step, hits gathered to rank 0 inside the step; all GPUs' rays per second, max over ranks")
        if small_batch:
            out["small_batch"] = {"legs": small_batch, "rays": small_batch[0]["rays"], "us": small_batch[0]["us"], "mrays": small_batch[0]["mrays"],
                                  "what": "lone launches over the first 2^17 (and 2^15) rays of the timed batch, HIP events around each, median of 24: what ONE GPU of eight is handed when a 2^20-ray batch is "
                                          "cut eight ways.  <= 2^16 rays take the static launch shape (no cursor, helpers from the first iteration; profiles/r06_batch_sweep.md); hit records byte-identical to "
                                          "the full batch's"}
        if sustained:
            out["sustained"] = sustained
        if gather_in_step and gather_ms:
            out["step_timing"] = {"trace_ms": round(avg_ms, 4), "gather_ms": round(float(np.mean(gather_ms)), 4), "gather_ms_max": round(float(np.max(gather_ms)), 4),
                                  "ms_per_step": round(1e3 * elapsed / args.steps, 4), "gather_bound": bool(float(np.mean(gather_ms)) > 0.9 * (1e3 * elapsed / args.steps)),
                                  "what": "per timed step on rank 0: the trace kernel (HIP events on the launch stream) and the collective (HIP events on the communication stream, from the moment the "
                                          "packed block was ready: includes waiting for the slowest peer).  gather_bound: the collective takes more than 0.9 of a step -- the pack of batch k + 2 "
                                          "waits for the gather of batch k, so the line is then the link's, not the kernel's"}
        if e2e:
            out["end_to_end"] = e2e
        if latency:
            out["per_call_latency"] = latency
        if multi:
            out["in_process_multi_gpu"] = multi
        if gather is not None:
            out["gather"] = gather
        if gather_check is not None:
            out["gather"] = gather_check
        if use_rccl and comm is None:
            out["gather"] = {"error": comm_err}
        failed = None
        if world == 1 and not args.no_cpu:
            try:
                cb, ref_traced = cpu_baseline(meshes, rays, any_hit=shadow)
                if cb:
                    out["cpu_baseline"] = cb
                    if cb.get("value_1Mi") and cb.get("value_16Mi"):   # (a GPU / CPU ratio quoted from this line must say WHICH CPU figure it divides by -- both, named)
                        out["gpu_over_cpu"] = {"vs_burst_1Mi": round(out["value"] / cb["value_1Mi"], 1), "vs_sustained_16Mi": round(out["value"] / cb["value_16Mi"], 1),
                                               "what": "value / the reference's rtcIntersect1 on %d hardware threads: the 2^20-ray job (a ~5 ms burst at full speed) and the 16 Mi-ray job (which this "
                                                       "container's cgroup quota of %s CPUs throttles); a reported baseline, not a target -- the roofline block says how good the kernel is" % (cb["cores"], cb.get("cpu_quota_cores"))}
                if not shadow:
                    try:
                        out["reference_visits"] = reference_visits(meshes, rays)
                    except Exception as e:                    # noqa: BLE001
                        out["reference_visits"] = {"error": repr(e)}
                if ref_traced is not None and not shadow:        # parity at full size against the real reference, ties classified
                    try:
                        out["parity_vs_reference"] = classify_parity(result, ref_traced, rays, meshes)
                    except AssertionError as e:
                        out["parity_vs_reference"] = {"FAILED": str(e)}
                        failed = str(e)
                elif ref_traced is not None:
                    flips = int((np.isneginf(ref_traced["tfar"]) != np.isneginf(result["tfar"])).sum())
                    out["parity_vs_reference"] = {"rays": M, "occlusion_flips": flips}
                    if flips > 1e-5 * M:
                        failed = "%d occlusion results differ from the reference" % flips
            except Exception as e:                            # the baseline leg must never take the GPU number down
                out["cpu_baseline"] = {"error": repr(e)}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        if failed:
            log("PARITY FAILURE: " + failed)
            os._exit(1)
    barrier()
    if dist:
        dist[0].destroy_process_group()
    sys.stdout.flush()
    if STUCK:
        os._exit(0)                                          # a helper thread stuck in a collective must not keep the process alive (normal exit otherwise: profilers flush at exit)


if __name__ == "__main__":
    main()
