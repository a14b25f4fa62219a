"""The renderer's draw order on the device (second half): SC_TICK_DRAWS | SC_TICK_SORT_DRAWS against
the oracle's restatement of VkRenderer's filter + sort (src/engine/src/sc_vk.cpp:1842-1864).  The oracle sorts
stably; so does the device, so the two lists must be identical item for item.  A second, order-free check asserts
what the reference itself guarantees: same multiset of draws, keys non-decreasing."""
import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj
from tests import worlds
from tests.test_gpu_parity import assert_mats_equal

pytestmark = pytest.mark.gpu
FLAGS = capi.XFORM | capi.CULL | capi.DRAWS | capi.SORT_DRAWS


def check_sorted(oracle, w, pipeline, mesh_count, max_draws=0, graph=False, expect_min=1):
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=False, max_draws=max_draws)
    vp = camera_view_proj(w.camera)
    t.set_view_proj(vp)
    t.set_draw_sort_table(pipeline, mesh_count)
    if graph:
        t.set_graph_mode(True)
    for _ in range(2 if graph else 1):
        ow.transform_system(); ow.culling_system(view_proj=vp)
        t.run(FLAGS)
    ent, mesh, mat, model, dropped = ow.draw_items(max_draws=max_draws)
    order = ow.renderer_draw_order(pipeline, mesh_count)
    idx, gmesh, gmat, gmodel = t.draws()
    c = t.counts()
    assert c.draws_emitted == len(ent) and c.draws_dropped == dropped and c.draws_sorted == len(order)
    assert len(order) >= expect_min
    assert np.array_equal(idx, ent[order])               # entity index == dense index in a bulk-built world
    assert np.array_equal(gmesh, mesh[order]) and np.array_equal(gmat, mat[order])
    assert_mats_equal(gmodel, model[order])
    # what the reference guarantees whatever its std::sort does with ties
    pipe = np.asarray(pipeline, np.uint8)
    keys = list(zip(pipe[gmat].tolist(), gmat.tolist(), gmesh.tolist()))
    assert keys == sorted(keys)
    t.close(); ow.close()
    return c


def test_sorted_draws_small_budget(oracle):
    w = worlds.random_world(6000, seed=51, spread=120.0, p_no_mesh=0.05)      # meshes 0..3, materials 0..5
    pipeline = np.array([1, 0, 1, 0xFF, 0, 1], np.uint8)                       # material 3 does not exist
    c = check_sorted(oracle, w, pipeline, mesh_count=3, max_draws=4096)        # mesh 3 is out of range
    assert c.draws_sorted < c.draws_emitted


def test_sorted_draws_handles_span_several_key_bytes(oracle):
    w = worlds.random_world(5000, seed=52, spread=100.0)
    rng = np.random.default_rng(3)
    w.mesh = rng.choice([0, 1, 255, 256, 70000, 2**24 - 1], w.n).astype(np.uint32)
    w.material = rng.choice([0, 7, 300, 65536, 65537], w.n).astype(np.uint32)
    pipeline = np.full(65538, 0xFF, np.uint8)
    pipeline[[0, 7, 300, 65536, 65537]] = [1, 0, 127, 0, 1]
    check_sorted(oracle, w, pipeline, mesh_count=2**24)


def test_sorted_draws_multi_workgroup_and_graph(oracle):
    w = worlds.random_world(60000, seed=55, spread=150.0, p_child=0.2)        # > 8192 visible, no budget: several workgroups per pass
    rng = np.random.default_rng(4)
    w.mesh = rng.integers(0, 40, w.n).astype(np.uint32)
    w.material = rng.integers(0, 300, w.n).astype(np.uint32)
    pipeline = (np.arange(300) % 2).astype(np.uint8)
    pipeline[::17] = 0xFF
    c = check_sorted(oracle, w, pipeline, mesh_count=37, expect_min=8193)
    check_sorted(oracle, w, pipeline, mesh_count=37, graph=True, expect_min=8193)


def test_nothing_survives_and_single_key(oracle):
    w = worlds.random_world(2000, seed=53, spread=80.0)
    ow = worlds.oracle_world(oracle, w, camera=False); ow.close()
    t = WorldTick.from_world(w, broadphase=False)
    t.set_camera(w.camera)
    t.set_draw_sort_table(np.full(6, 0xFF, np.uint8), 4)                        # no material exists
    t.run(FLAGS)
    assert t.counts().draws_sorted == 0 and len(t.draws()[0]) == 0
    t.set_draw_sort_table(np.zeros(0, np.uint8), 0)                             # empty tables
    t.run(FLAGS)
    assert t.counts().draws_sorted == 0
    w.mesh[:] = 2; w.material[:] = 1
    t.upload_render_meshes(0, w.has_mesh, w.mesh, w.material)
    t.set_draw_sort_table(np.array([0, 1], np.uint8), 3)
    t.run(FLAGS)
    idx = t.draws()[0]
    t.run(capi.XFORM | capi.CULL | capi.DRAWS)
    assert np.array_equal(idx, t.draws()[0])                                    # one key: the order is the visible list's
    t.close()


def test_sort_flag_without_table_is_an_error():
    w = worlds.random_world(100, seed=54)
    t = WorldTick.from_world(w, broadphase=False)
    t.set_camera(w.camera)
    assert t.lib.scTickRun(t.ctx, FLAGS) == 0
    assert b"scTickSetDrawSortTable" in t.lib.scTickGetLastError(t.ctx)
    bad = np.array([0, 200], np.uint8)
    assert t.lib.scTickSetDrawSortTable(t.ctx, bad.ctypes.data_as(capi.U8P), 2, 1) == 0
    assert t.lib.scTickSetDrawSortTable(None, None, 0, 0) == 0
    t.close()
