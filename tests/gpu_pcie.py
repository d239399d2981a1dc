"""What the host-array path of rtcIntersect1M is made of: pin, H2D, D2H, unpin of a 96 MB array (GPU box script)."""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
hip.hipHostRegister.argtypes = [vp, C.c_size_t, C.c_uint]; hip.hipHostUnregister.argtypes = [vp]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]; hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]; hip.hipStreamCreate.argtypes = [C.POINTER(vp)]; hip.hipStreamSynchronize.argtypes = [vp]
hip.hipMemcpy2DAsync.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, vp]
N = 1 << 20; B = N * 96
d = vp(); assert hip.hipMalloc(C.byref(d), B) == 0
s1, s2 = vp(), vp(); hip.hipStreamCreate(C.byref(s1)); hip.hipStreamCreate(C.byref(s2))
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for trial in range(2):
    a = np.zeros(B, np.uint8); a[:] = 1
    p = a.ctypes.data
    print("pageable H2D %.2f ms  D2H %.2f ms" % (t(lambda: hip.hipMemcpy(d, p, B, 1)), t(lambda: hip.hipMemcpy(p, d, B, 2))))
    treg = t(lambda: (hip.hipHostRegister(p, B, 0), hip.hipHostUnregister(p)), 3)
    t0 = time.perf_counter(); assert hip.hipHostRegister(p, B, 0) == 0; t1 = time.perf_counter()
    print("register+unregister %.2f ms (register alone %.2f ms)" % (treg, (t1 - t0) * 1e3))
    def h2d(): hip.hipMemcpyAsync(d, p, B, 1, s1); hip.hipStreamSynchronize(s1)
    def d2h(): hip.hipMemcpyAsync(p, d, B, 2, s1); hip.hipStreamSynchronize(s1)
    def both(): hip.hipMemcpyAsync(d, p, B // 2, 1, s1); hip.hipMemcpyAsync(vp(p + B // 2), vp(d.value + B // 2), B // 2, 2, s2); hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    def d2h2d(): hip.hipMemcpy2DAsync(vp(p + 32), 96, vp(d.value + 32), 96, 64, N, 2, s1); hip.hipStreamSynchronize(s1)
    def h2d2d(): hip.hipMemcpy2DAsync(d, 96, p, 96, 48, N, 1, s1); hip.hipStreamSynchronize(s1)
    print("pinned H2D 96 MB %.2f ms | D2H 96 MB %.2f ms | 48 MB each way at once %.2f ms | 2D D2H 64 of 96 B %.2f ms | 2D H2D 48 of 96 B %.2f ms" % (t(h2d), t(d2h), t(both), t(d2h2d), t(h2d2d)))
    hip.hipHostUnregister(p)
# ---- the pipeline's copy pattern without kernels: chunks go round k streams, each chunk H2D then D2H in its stream
a = np.zeros(B, np.uint8); a[:] = 1; p = a.ctypes.data
assert hip.hipHostRegister(p, B, 0) == 0
streams = []
for _ in range(8):
    s_ = vp(); hip.hipStreamCreate(C.byref(s_)); streams.append(s_)
for k, chunk in ((2, 1 << 17), (4, 1 << 17), (4, 1 << 16), (8, 1 << 16), (4, 1 << 18), (1, 1 << 20)):
    def pipe():
        c = 0
        for first in range(0, N, chunk):
            q = streams[c % k]; o = first * 96; nb = chunk * 96
            hip.hipMemcpyAsync(vp(d.value + o), vp(p + o), nb, 1, q)
            hip.hipMemcpyAsync(vp(p + o), vp(d.value + o), nb, 2, q)
            c += 1
        for q in streams[:k]: hip.hipStreamSynchronize(q)
    print("copy pipeline: %d streams, chunks of %d rays: %.2f ms" % (k, chunk, t(pipe)))
# ---- dedicated upload / download streams, an event per chunk (upload k+1 overlaps download k)
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]; hip.hipEventRecord.argtypes = [vp, vp]; hip.hipStreamWaitEvent.argtypes = [vp, vp, C.c_uint]
evs = []
for _ in range(64):
    e = vp(); assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0; evs.append(e)      # hipEventDisableTiming
up, down = streams[0], streams[1]
for chunk in (1 << 16, 1 << 17, 1 << 18):
    def pipe2():
        c = 0
        for first in range(0, N, chunk):
            o = first * 96; nb = chunk * 96
            hip.hipMemcpyAsync(vp(d.value + o), vp(p + o), nb, 1, up)
            hip.hipEventRecord(evs[c], up); hip.hipStreamWaitEvent(down, evs[c], 0)
            hip.hipMemcpyAsync(vp(p + o), vp(d.value + o), nb, 2, down)
            c += 1
        hip.hipStreamSynchronize(up); hip.hipStreamSynchronize(down)
    print("upload stream + download stream, chunks of %d rays: %.2f ms" % (chunk, t(pipe2)))
