#!/usr/bin/env python3
"""What the first HIP calls of a process cost on this box (ctypes on libamdhip64 + this library): the fixed price every
command-line run pays before its first chunk is through.  One JSON line.  Builder tool."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

t = {}


def timed(name, fn):
    t0 = time.perf_counter()
    r = fn()
    t[name] = round(time.perf_counter() - t0, 4)
    return r


hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
timed("hipInit", lambda: hip.hipInit(0))
timed("hipSetDevice", lambda: hip.hipSetDevice(0))
s1, s2 = vp(), vp()
timed("hipStreamCreate_1", lambda: hip.hipStreamCreateWithFlags(C.byref(s1), 1))
d1, d2 = vp(), vp()
timed("hipMalloc_1MB", lambda: hip.hipMalloc(C.byref(d1), C.c_size_t(1 << 20)))
timed("hipMalloc_256MB", lambda: hip.hipMalloc(C.byref(d2), C.c_size_t(256 << 20)))
timed("hipMemsetAsync_first+sync", lambda: (hip.hipMemsetAsync(d2, 0, C.c_size_t(1 << 20), s1), hip.hipStreamSynchronize(s1)))
timed("hipMemsetAsync_second+sync", lambda: (hip.hipMemsetAsync(d2, 0, C.c_size_t(1 << 20), s1), hip.hipStreamSynchronize(s1)))
h1 = vp()
timed("hipHostMalloc_32MB", lambda: hip.hipHostMalloc(C.byref(h1), C.c_size_t(32 << 20), 0))
h2 = vp()
timed("hipHostMalloc_32MB_second", lambda: hip.hipHostMalloc(C.byref(h2), C.c_size_t(32 << 20), 0))
timed("touch_32MB", lambda: C.memset(h1, 1, 32 << 20))
timed("H2D_32MB_first+sync", lambda: (hip.hipMemcpyAsync(d2, h1, C.c_size_t(32 << 20), 1, s1), hip.hipStreamSynchronize(s1)))
timed("H2D_32MB_second+sync", lambda: (hip.hipMemcpyAsync(d2, h1, C.c_size_t(32 << 20), 1, s1), hip.hipStreamSynchronize(s1)))
timed("hipStreamCreate_2", lambda: hip.hipStreamCreateWithFlags(C.byref(s2), 1))
timed("H2D_32MB_stream2_first+sync", lambda: (hip.hipMemcpyAsync(d2, h1, C.c_size_t(32 << 20), 1, s2), hip.hipStreamSynchronize(s2)))
timed("D2H_1MB_first+sync", lambda: (hip.hipMemcpyAsync(h2, d2, C.c_size_t(1 << 20), 2, s1), hip.hipStreamSynchronize(s1)))
h3 = vp()
hip.hipHostMalloc(C.byref(h3), C.c_size_t(128 << 20), 0)
C.memset(h3, 1, 128 << 20)
for mb in (4, 8, 16, 32, 64, 128):
    for rep in range(2):
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(d2, h3, C.c_size_t(mb << 20), 1, s1)
        t1 = time.perf_counter()
        hip.hipStreamSynchronize(s1)
        t2 = time.perf_counter()
    t["H2D_%dMB_enqueue_ms" % mb] = round((t1 - t0) * 1e3, 3)
    t["H2D_%dMB_done_ms" % mb] = round((t2 - t0) * 1e3, 3)
# two copies back to back on one stream / on two streams
for name, sa, sb in (("one_stream", s1, s1), ("two_streams", s1, s2)):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(d2, h3, C.c_size_t(32 << 20), 1, sa)
    hip.hipMemcpyAsync(vp(d2.value + (64 << 20)), vp(h3.value + (64 << 20)), C.c_size_t(32 << 20), 1, sb)
    t1 = time.perf_counter()
    hip.hipStreamSynchronize(sa); hip.hipStreamSynchronize(sb)
    t["2x32MB_%s_enqueue_ms" % name] = round((t1 - t0) * 1e3, 3)
    t["2x32MB_%s_done_ms" % name] = round((time.perf_counter() - t0) * 1e3, 3)
from badger_amd import _native  # noqa: E402
_native.PRELOAD_TORCH = False
timed("load_library", _native.load)
ctx = timed("bdg_init", lambda: _native.Context(0))
b = np.frombuffer(b"ACGT" * 300, dtype=np.uint8).copy()
off = np.array([0, 600, 1200], dtype=np.uint64)
timed("extract_batch_first", lambda: ctx.extract_batch(b, off, 12))
timed("extract_batch_second", lambda: ctx.extract_batch(b, off, 12))
wl = np.arange(1000, dtype=np.uint32) * 7919
timed("nearest16_first(other module)", lambda: ctx.nearest16(wl[:10], wl, 2))
timed("graph_edges_first(other module)", lambda: ctx.graph_edges(wl, 1, 5))
print(json.dumps(t))
os._exit(0)
