"""Error behaviour of the C ABI on a live context: bad arguments fail (0 / exception) with a message and
leave the context usable -- the reference's C ABI convention (int 1 = ok, 0 = failed; null-tolerant)."""
import ctypes as C

import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj
from tests import worlds

pytestmark = pytest.mark.gpu


def err(t):
    return (t.lib.scTickGetLastError(t.ctx) or b"").decode()


def test_argument_errors_are_reported_and_recoverable():
    w = worlds.random_world(300, seed=41)
    t = WorldTick.from_world(w, broadphase=False, capacity=512)
    with pytest.raises(capi.ScTickError, match="exceeds capacity"):
        t.set_count(513)
    with pytest.raises(capi.ScTickError, match="range exceeds"):
        t.upload_positions(299, np.zeros((2, 3), np.float32))
    with pytest.raises(capi.ScTickError, match="not affine"):
        m = np.tile(np.eye(4, dtype=np.float32).ravel(), (1, 1)); m[0, 3] = 0.5
        t.upload_world_matrices(0, m)
    with pytest.raises(capi.ScTickError, match="bits above 15"):
        t.upload_layers(0, np.array([1 << 20], np.uint32), np.array([1], np.uint32))
    with pytest.raises(capi.ScTickError, match="topology must cover"):
        t.set_topology(np.full(10, -1, np.int32))
    with pytest.raises(capi.ScTickError, match="out of range"):
        t.mark_dirty_indices(np.array([300], np.uint32))
    with pytest.raises(capi.ScTickError, match="no tile rectangle"):
        t.run(capi.FULL)
    t.run(capi.XFORM | capi.CULL)
    with pytest.raises(capi.ScTickError, match="CULLED_LIST"):
        t.culled()
    with pytest.raises(capi.ScTickError, match="BROADPHASE"):
        t.pairs()
    with pytest.raises(capi.ScTickError, match="DRAWS"):
        t.draws()
    with pytest.raises(capi.ScTickError, match="no movers"):
        t.advance_movers(0.016)
    # null pointers where data is required
    assert t.lib.scTickUploadPositions(t.ctx, 0, 1, None) == 0 and "null" in err(t)
    assert t.lib.scTickGetCounts(t.ctx, None) == 0
    cnt = C.c_uint32()
    assert t.lib.scTickGetKernelTimes(t.ctx, 99, None, 0, C.byref(cnt)) == 0
    # the context still works after all of that
    t.run(capi.XFORM | capi.CULL | capi.CULLED_LIST)
    assert len(t.visible()) + len(t.culled()) == int(w.has_mesh.sum())
    t.close()


def test_create_context_rejects_bad_descriptors():
    lib = capi.load()
    d = capi.ContextDesc()
    d.capacity = 0
    assert lib.scTickCreateContext(C.byref(d)) is None and b"capacity" in lib.scTickGetLastError(None)
    d.capacity = (1 << 24) + 1
    assert lib.scTickCreateContext(C.byref(d)) is None
    d.capacity = 16; d.device_ordinal = 99
    assert lib.scTickCreateContext(C.byref(d)) is None and b"device ordinal" in lib.scTickGetLastError(None)


def test_split_pairs_protocol_is_enforced():
    from sc_gameengine_amd import synth_world as sw
    w = sw.generate(4, 4, 15)
    t = WorldTick.from_world(w, broadphase=True)
    with pytest.raises(capi.ScTickError, match="without a preceding"):
        t.run_pairs()
    t.run(capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS)
    with pytest.raises(capi.ScTickError, match="scTickRunPairs has not been called"):
        t.pairs()
    t.run_pairs()
    got, total = t.pairs()
    assert total == len(got)
    t.close()


def test_profiling_samples_only_the_kernels_asked_for_and_learn_ticks_is_a_host_counter():
    """scTickSetProfiling(n) times every n-th tick's launches by events; scTickSetProfilingKernels narrows that to some kernels (what a
    run that is itself being timed uses: a timed launch costs its queue a few microseconds); scTickGetLearnTicks is scTickGetBinStats'
    learn-tick count without the read-back."""
    w = sw.generate(8, 8, 15)
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(camera_view_proj(w.camera))
    flags = capi.FULL
    t.set_profiling_kernels([capi.K_XFORM_CULL])
    t.set_profiling(2)
    for _ in range(6):
        t.run(flags)
    t.sync()
    assert len(t.kernel_times_ms(capi.K_XFORM_CULL)) == 3 and len(t.kernel_times_ms(capi.K_PAIRS)) == 0
    t.set_profiling_kernels(None)
    t.set_profiling(1)
    for _ in range(4):
        t.run(flags)
    t.sync()
    k1, kp = t.kernel_times_ms(capi.K_XFORM_CULL), t.kernel_times_ms(capi.K_PAIRS)
    assert len(k1) == 4 and len(kp) == 4 and all(x > 0 for x in list(k1) + list(kp))
    t.set_profiling(0)
    assert t.learn_ticks() == t.bin_stats()["learn_ticks"] >= 1
    t.close()
