"""Sector residency on the device: scTickAppendEntities / scTickRemoveEntities against the
oracle's ECS doing what WorldPartition::pumpCompletedLoads / pumpUnloadQueue do to the reference's World
(create + add components; destroy = swap-remove in every pool).  The oracle's pool order is pinned against
the reference's own ComponentPool (tests/golden/sc_ecs_ref.json).  After every change a full tick must
match: dense order, world matrices (IEEE equality), dirty flags, ordered visible list, pair set."""
import os

import numpy as np
import pytest

from sc_gameengine_amd import capi, sectors, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj
from tests import worlds
from tests.test_gpu_broadphase import sorted_pairs
from tests.test_gpu_parity import assert_mats_equal

pytestmark = pytest.mark.gpu
FLAGS = capi.FULL | capi.DENSE_AABBS | capi.CULLED_LIST
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sectors")
UNIT_MIN, UNIT_MAX = np.float32([-0.5, -0.5, -0.5]), np.float32([0.5, 0.5, 0.5])


class Twin:
    """One world kept in the oracle's ECS and in a device context, plus the per-entity arrays the oracle's ECS
    does not hold (collision layers), in dense order."""

    def __init__(self, oracle, w, capacity, **kw):
        self.oracle = oracle
        self.ow = worlds.oracle_world(oracle, w, camera=False)
        self.t = WorldTick.from_world(w, broadphase=True, capacity=capacity, **kw)
        self.group, self.mask = w.group.copy(), w.mask.copy()
        self.vp = camera_view_proj(w.camera)
        self.t.set_view_proj(self.vp)

    def close(self):
        self.t.close(); self.ow.close()

    def remove(self, idx):
        idx = np.asarray(idx, np.uint32)
        ents_before = self.ow.dense_entities()
        for e in ents_before[idx]:
            assert self.ow.destroy(int(e))
        src, dst = self.t.remove_entities(idx)
        ents_after = self.ow.dense_entities()
        # the relocations reported are exactly how the reference's pool ended up
        n1 = len(ents_after)
        assert self.t.n == n1 == len(ents_before) - len(idx)
        expect = ents_before[:n1].copy()
        expect[dst] = ents_before[src]
        assert np.array_equal(expect, ents_after)
        assert (src >= n1).all() and (dst < n1).all() and len(np.unique(dst)) == len(dst)
        for name in ("group", "mask"):
            a = getattr(self, name)
            b = a[:n1].copy(); b[dst] = a[src]
            setattr(self, name, b)
        return src, dst

    def append(self, pos, rot, scale, mesh, material, parent=None, bmin=None, bmax=None, group=None, mask=None):
        k = len(pos)
        first_dense = self.ow.count()
        new_ents = []
        for i in range(k):
            e = self.ow.create()
            tr = self.ow.add_transform(e)
            for a in range(3):
                tr.localPos[a], tr.localRot[a], tr.localScale[a] = pos[i][a], rot[i][a], scale[i][a]
            tr.dirty = 1
            self.oracle_set_mesh_ids([e], [mesh[i]], [material[i]])
            self.ow.add_bounds(e, UNIT_MIN if bmin is None else bmin[i], UNIT_MAX if bmax is None else bmax[i])
            new_ents.append(e)
        if parent is not None:
            ents = self.ow.dense_entities()
            for i, p in enumerate(parent):
                if p >= 0:
                    self.ow.get_transform(new_ents[i]).parent = int(ents[p])
        first = self.t.append_entities(pos, rot, scale, bmin=bmin, bmax=bmax, mesh=mesh, material=material,
                                       group=group, mask=mask, parent=parent)
        assert first == first_dense
        self.group = np.concatenate([self.group, np.full(k, 0xFFFFFFFF, np.uint32) if group is None else np.asarray(group, np.uint32)])
        self.mask = np.concatenate([self.mask, np.full(k, 0xFFFFFFFF, np.uint32) if mask is None else np.asarray(mask, np.uint32)])
        return first

    def oracle_set_mesh_ids(self, ents, mesh, material):
        import ctypes as C
        L = self.ow.L
        for e, m, t in zip(ents, mesh, material):
            p = C.cast(L.orc_add_render_mesh(self.ow.w, e), C.POINTER(C.c_uint32))
            p[0], p[1] = int(m), int(t)

    def tick_and_compare(self, pairs=True):
        ow, t = self.ow, self.t
        ow.transform_system()
        ow.culling_system(view_proj=self.vp)
        t.run(FLAGS)
        n = ow.count()
        assert t.counts().entities == n
        assert_mats_equal(t.world_matrices(), ow.world_matrices())
        assert np.array_equal(t.dirty(), ow.dirty())
        ents = ow.dense_entities()
        dense_of = {int(e): i for i, e in enumerate(ents)}
        assert t.visible().tolist() == [dense_of[int(e)] for e in ow.visible()]
        assert t.culled().tolist() == [dense_of[int(e)] for e in ow.culled()]
        c = t.counts()
        assert c.renderables_total == len(ow.candidates())
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)
        if pairs:
            want = self.oracle.broadphase_bruteforce(mn, mx, self.group, self.mask)
            got, total = t.pairs()
            assert total == len(want)
            assert np.array_equal(sorted_pairs(got), want)
        # draw items follow the visible list
        ent, mesh, mat, model, _ = ow.draw_items()
        t.run(capi.DRAWS)
        d_idx, d_mesh, d_mat, d_model = t.draws()
        assert d_idx.tolist() == [dense_of[int(e)] for e in ent]
        assert np.array_equal(d_mesh, mesh) and np.array_equal(d_mat, mat)
        assert_mats_equal(d_model, model)


def test_stream_sectors_out_and_in(oracle):
    w = sw.config("config1")                               # 1024 roots, every one renderable
    tw = Twin(oracle, w, capacity=4096)
    tw.tick_and_compare()
    rng = np.random.default_rng(7)
    sixteen = sectors.read_sector_file(os.path.join(GOLD, "v4_sixteen.scsector"))
    full = sectors.read_sector_file(os.path.join(GOLD, "v4_full.scsector"))
    for rnd in range(6):
        n = tw.ow.count()
        # despawn one "sector" (a contiguous run) and a few strays, in a shuffled order
        start = int(rng.integers(0, n - 40))
        idx = np.concatenate([np.arange(start, start + 16), rng.choice(np.setdiff1d(np.arange(n), np.arange(start, start + 16)), 9, replace=False)])
        if rnd == 2:
            idx = np.unique(np.concatenate([idx, [n - 1, n - 2]]))      # tail entities too: holes that need no move
        rng.shuffle(idx)
        src, dst = tw.remove(idx)
        tw.tick_and_compare()
        # activate a sector read from a .scsector file: positions pulled into the grid so that pairs happen
        sec = sixteen if rnd % 2 == 0 else full
        pos = (np.abs(sec.pos) % 300.0 + 100.0).astype(np.float32)
        pos[:, 1] = 0.5
        scale = np.where(sec.scale == 0, sec.scale, np.minimum(sec.scale, 6.0)).astype(np.float32)      # keeps the all-zero scale the fixture carries
        first = tw.append(pos, sec.rot, scale, mesh=(sec.mesh_id % 7).astype(np.uint32), material=(sec.material_id % 5).astype(np.uint32))
        assert first == tw.ow.count() - sec.instances
        tw.tick_and_compare()
    c = tw.t.counts()
    assert c.max_depth == 0 and c.unreachable == 0
    tw.close()


def test_despawn_and_spawn_inside_hierarchies(oracle):
    w = worlds.random_world(1500, seed=41, spread=150.0, max_depth=3, zero_scales=5)
    tw = Twin(oracle, w, capacity=4096)
    tw.tick_and_compare()
    rng = np.random.default_rng(8)
    for rnd in range(5):
        n = tw.ow.count()
        idx = rng.choice(n, 60, replace=False)               # parents, children, leaves alike: orphans become dirty roots
        tw.remove(idx)
        tw.tick_and_compare()
        # new entities: some roots, some children of old entities, some children of entities of the same batch
        k = 40
        n = tw.ow.count()
        parent = np.full(k, -1, np.int32)
        parent[10:20] = rng.integers(0, n, 10)
        parent[20:30] = n + rng.integers(0, 10, 10)          # batch-mates (roots of this batch)
        parent[30:35] = n + np.arange(20, 25)                # grandchildren through the batch
        pos = rng.uniform(-100, 100, (k, 3)).astype(np.float32)
        pos[10:] = rng.uniform(-2, 2, (k - 10, 3))
        rot = rng.uniform(-3, 3, (k, 3)).astype(np.float32)
        scale = rng.uniform(0.5, 2, (k, 3)).astype(np.float32)
        tw.append(pos, rot, scale, mesh=rng.integers(0, 4, k).astype(np.uint32), material=rng.integers(0, 4, k).astype(np.uint32),
                  parent=parent, group=np.full(k, 1, np.uint32), mask=np.full(k, 0xFFFFFFFF, np.uint32))
        tw.tick_and_compare()
    tw.close()


def test_deep_chains_and_cycles_survive_removal(oracle):
    w = worlds.chain_world(10, branches=6, seed=42)          # depth 9: level kernels
    w.parent[3] = 57; w.parent[57] = 3                       # a cycle: both unreachable, with whatever hangs below
    tw = Twin(oracle, w, capacity=256)
    tw.tick_and_compare(pairs=False)
    assert tw.t.counts().unreachable > 0
    tw.remove([20, 3, 59, 0])
    tw.tick_and_compare(pairs=False)
    tw.remove([int(tw.ow.count()) - 1, 5])
    tw.tick_and_compare(pairs=False)
    tw.close()


def test_movers_move_with_their_entities(oracle):
    w = sw.generate_config5(4, 4)
    tw = Twin(oracle, w, capacity=w.n + 64)
    kind, vel, lo, hi = w.mover_kind.copy(), w.mover_vel.copy(), w.mover_lo.copy(), w.mover_hi.copy()
    rng = np.random.default_rng(9)
    for rnd in range(3):
        n = tw.ow.count()
        src, dst = tw.remove(rng.choice(n, 37, replace=False))
        n1 = tw.ow.count()
        moved = []
        for a in (kind, vel, lo, hi):
            b = a[:n1].copy(); b[dst] = a[src]; moved.append(b)
        kind, vel, lo, hi = moved
        for _ in range(2):
            tw.ow.advance_movers(kind, vel, lo, hi, 1.0 / 60.0)
            tw.t.advance_movers(1.0 / 60.0)
            tw.tick_and_compare()
        assert np.array_equal(tw.t.mover_velocities(), vel)
    tw.close()


def test_graph_mode_recaptures_after_residency_changes(oracle):
    w = sw.config("config1")
    tw = Twin(oracle, w, capacity=2048)
    tw.t.set_graph_mode(True)
    tw.tick_and_compare()
    tw.remove(np.arange(100, 164))
    tw.tick_and_compare()
    tw.tick_and_compare()
    sec = sectors.read_sector_file(os.path.join(GOLD, "v4_sixteen.scsector"))
    tw.t.activate_sector(sec)                                   # device only ...
    tw.t.remove_entities(np.arange(tw.t.n - 16, tw.t.n))        # ... and gone again: tail removal moves nothing
    tw.tick_and_compare()
    tw.close()


def test_argument_errors_leave_the_world_alone(oracle):
    w = sw.config("config1")
    tw = Twin(oracle, w, capacity=1030)
    tw.tick_and_compare()
    lib, ctx = tw.t.lib, tw.t.ctx
    bad = np.array([5, 5], np.uint32)
    assert lib.scTickRemoveEntities(ctx, bad.ctypes.data_as(capi.U32P), 2, None, None, None) == 0
    assert b"twice" in lib.scTickGetLastError(ctx)
    bad = np.array([5, 4000], np.uint32)
    assert lib.scTickRemoveEntities(ctx, bad.ctypes.data_as(capi.U32P), 2, None, None, None) == 0
    assert b"out of range" in lib.scTickGetLastError(ctx)
    z = np.zeros((8, 3), np.float32)
    assert lib.scTickAppendEntities(ctx, 8, z.ctypes.data_as(capi.F32P), z.ctypes.data_as(capi.F32P), z.ctypes.data_as(capi.F32P),
                                    None, None, None, None, None, None, None, None) == 0
    assert b"capacity" in lib.scTickGetLastError(ctx)
    par = np.array([2000], np.int32)
    assert lib.scTickAppendEntities(ctx, 1, z.ctypes.data_as(capi.F32P), z.ctypes.data_as(capi.F32P), z.ctypes.data_as(capi.F32P),
                                    None, None, None, None, None, None, par.ctypes.data_as(capi.I32P), None) == 0
    assert lib.scTickAppendEntities(None, 0, None, None, None, None, None, None, None, None, None, None, None) == 0
    assert lib.scTickRemoveEntities(None, None, 0, None, None, None) == 0
    assert tw.t.counts().entities == 1024
    tw.tick_and_compare()
    tw.close()


def test_large_batch_removal_1m():
    """Full-size property check: removing 100k of 1M entities keeps every survivor's data intact."""
    w = sw.config("config3")
    t = WorldTick.from_world(w, broadphase=True)
    t.set_camera(w.camera)
    t.run(capi.FULL)
    before = t.positions()
    rng = np.random.default_rng(10)
    idx = rng.choice(w.n, 100_000, replace=False).astype(np.uint32)
    src, dst = t.remove_entities(idx)
    n1 = w.n - len(idx)
    after = t.positions()
    expect = before[:n1].copy(); expect[dst] = before[src]
    assert np.array_equal(after, expect)
    gone = np.zeros(w.n, bool); gone[idx] = True
    survivors = np.sort(np.concatenate([np.flatnonzero(~gone[:n1]), src]))
    assert np.array_equal(survivors, np.flatnonzero(~gone))           # nobody lost, nobody duplicated
    t.run(capi.FULL)
    c = t.counts()
    assert c.entities == n1 and 0 < c.visible < n1 and c.pairs_truncated == 0
    t.close()


def test_whole_sectors_of_a_hierarchical_world_stream_without_relinking(oracle):
    """Despawning complete subtrees relocates tail entities that are parents and children themselves; their links are
    patched in place (no O(N) re-link) and every later tick still matches -- moving roots drag the right children."""
    w = sw.generate(8, 8, 15, hierarchy=True)              # per sector: ground + 15 props in runs root, child, grandchild, root
    tw = Twin(oracle, w, capacity=w.n)
    tw.tick_and_compare()
    relinks0 = tw.t.counts().relinks
    rng = np.random.default_rng(11)
    for rnd in range(8):
        # a sector's 16 entities (complete subtrees) wherever earlier relocations have put them: entity handles of a
        # bulk-built world are the original dense indices, 16 per sector
        ents = tw.ow.dense_entities()
        sector_of = (ents & 0xFFFFFF) // 16
        pick = rng.choice(np.unique(sector_of), 3, replace=False)
        idx = np.flatnonzero(np.isin(sector_of, pick))
        assert len(idx) == 48
        if rnd % 2:
            rng.shuffle(idx)
        tw.remove(idx)
        tw.ow.nudge_roots_x(0.25); tw.t.nudge_roots_x(0.25)
        tw.tick_and_compare()
    c = tw.t.counts()
    assert c.relinks == relinks0, "complete subtrees must not trigger a world re-link"
    assert c.max_depth == 2
    # a removal that orphans a child does re-link (and the orphan becomes a dirty root)
    parents = tw.ow.parents()
    ents = tw.ow.dense_entities()
    dense_of = {int(e): i for i, e in enumerate(ents)}
    child = next(i for i in range(len(parents)) if parents[i] != 0xFFFFFFFF)
    tw.remove([dense_of[int(parents[child])]])
    tw.tick_and_compare()
    assert tw.t.counts().relinks == relinks0 + 1
    tw.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_residency_sequences(oracle, seed):
    """Random mix of despawns (arbitrary entities: leaves, parents, whole or partial subtrees), activations (roots, children
    of survivors, children of batch-mates), moves and ticks; everything is compared after every tick."""
    rng = np.random.default_rng(500 + seed)
    w = worlds.random_world(int(rng.integers(600, 2500)), seed=200 + seed, spread=float(rng.choice([80.0, 300.0])),
                            max_depth=int(rng.integers(1, 6)), zero_scales=3)
    tw = Twin(oracle, w, capacity=8192, max_pairs=1 << 18)
    tw.tick_and_compare()
    for step in range(8):
        n = tw.ow.count()
        op = rng.integers(0, 4)
        if op in (0, 1) and n > 200:
            k = int(rng.integers(1, min(n // 3, 150)))
            if op == 0:
                idx = rng.choice(n, k, replace=False)
            else:                                                   # a tail-heavy batch: many removals need no move
                idx = np.unique(np.concatenate([np.arange(n - k // 2, n), rng.choice(n, k // 2 + 1, replace=False)]))
                rng.shuffle(idx)
            tw.remove(idx)
        if op in (1, 2, 3):
            k = int(rng.integers(1, 80))
            n = tw.ow.count()
            parent = None
            if op != 2:
                parent = np.full(k, -1, np.int32)
                sel = rng.random(k) < 0.5
                parent[sel] = rng.integers(0, n + k, int(sel.sum()))
                parent[parent == n + np.arange(k)] = -1             # no self parents
            pos = rng.uniform(-250, 250, (k, 3)).astype(np.float32)
            tw.append(pos, rng.uniform(-3, 3, (k, 3)).astype(np.float32), rng.uniform(0.3, 2.5, (k, 3)).astype(np.float32),
                      mesh=rng.integers(0, 5, k).astype(np.uint32), material=rng.integers(0, 5, k).astype(np.uint32), parent=parent,
                      group=rng.choice(np.array([1, 2], np.uint32), k), mask=rng.choice(np.array([0xFFFFFFFF, 1], np.uint32), k))
        # gameplay moves a few entities
        n = tw.ow.count()
        ids = np.sort(rng.choice(n, min(n, 40), replace=False)).astype(np.uint32)
        newp = rng.uniform(-250, 250, (len(ids), 3)).astype(np.float32)
        tw.ow.set_local_positions(tw.ow.dense_entities()[ids], newp)
        for j in range(len(ids)):
            tw.t.upload_positions(int(ids[j]), newp[j:j + 1])
        tw.tick_and_compare()
    tw.close()
