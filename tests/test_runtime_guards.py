"""GPU: runtime behaviours the engine works around, watched so that a ROCm change is noticed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_small_memset_nodes_in_a_replayed_graph():
    """hipMemsetAsync captured into a hipGraph wrote garbage into 16..160-byte buffers from the third
    replay of the same executable graph on ROCm 7.0 / HIP 7.0.5, which is why every zero-fill inside a captured run is a
    kernel node (k_zero_words).  This test replays such a graph: it passes when memset nodes behave (the workaround could
    then go), and reports the known defect as an expected failure otherwise."""
    hip = C.CDLL("libamdhip64.so")
    vp = C.c_void_p
    hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
    hip.hipMemsetAsync.argtypes = [vp, C.c_int, C.c_size_t, vp]
    hip.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]

    def ok(rc):
        assert rc == 0, "HIP error %d" % rc

    sizes = [16, 32, 64, 160]
    bufs = []
    for n in sizes:
        p = vp()
        ok(hip.hipMalloc(C.byref(p), n))
        bufs.append(p)
    s = vp()
    ok(hip.hipStreamCreateWithFlags(C.byref(s), 1))
    ok(hip.hipStreamBeginCapture(s, 2))  # hipStreamCaptureModeThreadLocal... relaxed = 2
    for p, n in zip(bufs, sizes):
        ok(hip.hipMemsetAsync(p, 0, n, s))
    g = vp()
    ok(hip.hipStreamEndCapture(s, C.byref(g)))
    ex = vp()
    ok(hip.hipGraphInstantiate(C.byref(ex), g, None, None, 0))
    bad = []
    for rep in range(6):
        for p, n in zip(bufs, sizes):
            ok(hip.hipMemset(p, 0xAB, n))
        ok(hip.hipGraphLaunch(ex, s))
        ok(hip.hipStreamSynchronize(s))
        for p, n in zip(bufs, sizes):
            host = (C.c_ubyte * n)()
            ok(hip.hipMemcpy(host, p, n, 2))
            if any(host):
                bad.append((rep, n, bytes(host)[:8].hex()))
    if bad:
        pytest.xfail("memset graph nodes still write garbage on replay (first: replay %d, %d bytes, %s...): k_zero_words stays" % bad[0])
