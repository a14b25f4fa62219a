"""GPU broadphase against the oracle's brute force / grid (own spec):
the pair SET must be identical; world AABBs must equal the oracle's as IEEE values."""
import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick
from tests import worlds

pytestmark = pytest.mark.gpu
FLAGS = capi.XFORM | capi.BROADPHASE | capi.DENSE_AABBS


def sorted_pairs(p):
    p = np.asarray(p, np.uint32).reshape(-1, 2)
    if len(p) == 0:
        return p
    k = p[:, 0].astype(np.uint64) << np.uint64(32) | p[:, 1].astype(np.uint64)
    return p[np.argsort(k)]


def gpu_vs_oracle(oracle, w, brute=True, cell=64.0, flags=FLAGS, ticks=1, nudge=None, max_pairs=0):
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=max_pairs)
    for k in range(ticks):
        if nudge is not None and k:
            ow.nudge_roots_x(nudge); t.nudge_roots_x(nudge)
        ow.transform_system()
        t.run(flags)
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)            # IEEE equality, inf for "no Bounds"
        want = oracle.broadphase_bruteforce(mn, mx, w.group, w.mask) if brute else oracle.broadphase_grid(mn, mx, w.group, w.mask, cell)
        got, total = t.pairs()
        assert total == len(want), f"pair count {total} != {len(want)}"
        assert np.array_equal(sorted_pairs(got), want)
        assert (got[:, 0] < got[:, 1]).all() if len(got) else True
    return t, ow


def test_random_world_mixed_layers(oracle):
    w = worlds.random_world(3000, seed=31, spread=150.0, max_depth=3)
    t, ow = gpu_vs_oracle(oracle, w, ticks=3, nudge=0.7)
    c = t.counts()
    assert c.pairs > 50 and c.pairs_truncated == 0
    t.close(); ow.close()


def test_big_boxes_and_outside_rect(oracle):
    w = worlds.random_world(2500, seed=32, spread=700.0, max_depth=2)      # rect is +-512 m: many boxes fall outside it
    w.bmin[:40] *= 30.0; w.bmax[:40] *= 30.0                               # larger than 2x2 sectors
    t, ow = gpu_vs_oracle(oracle, w)
    c = t.counts()
    assert c.big_boxes > 100 and c.pairs > 0
    t.close(); ow.close()


def test_bin_overflow_takes_the_slow_path(oracle):
    w = worlds.random_world(1500, seed=33, spread=300.0, p_child=0.0)
    w.pos[:300] = np.float32([10.0, 0.0, 10.0]) + np.random.default_rng(1).uniform(-8, 8, (300, 3)).astype(np.float32)  # 300 boxes in one sector
    t, ow = gpu_vs_oracle(oracle, w)
    c = t.counts()
    assert c.bin_overflow > 200 and c.pairs > 1000
    t.close(); ow.close()


def test_deep_chains_broadphase(oracle):
    w = worlds.chain_world(12, branches=20, seed=34)
    w.pos[:20] = np.random.default_rng(2).uniform(-100, 100, (20, 3)).astype(np.float32)
    t, ow = gpu_vs_oracle(oracle, w, ticks=2, nudge=1.5)
    t.close(); ow.close()


def test_synthworld_with_dynamic_layer_vs_cpu_grid(oracle):
    w = sw.generate(32, 32, 15)                                   # 16 384 entities
    dyn = (np.arange(w.n) % 5) == 2
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    t, ow = gpu_vs_oracle(oracle, w, brute=False, cell=16.0, ticks=3, nudge=0.5)
    assert t.counts().big_boxes == 0 and t.counts().pairs > 100        # edge slabs land in the ring, not the big list
    t.close(); ow.close()


def test_all_static_world_reports_no_pairs(oracle):
    w = sw.generate(16, 16, 15)                                   # static props: group 2 / mask 1 never pass the filter
    t, ow = gpu_vs_oracle(oracle, w, brute=False, cell=16.0)
    assert t.counts().pairs == 0
    t.close(); ow.close()


def test_pair_list_truncation_is_reported(oracle):
    w = worlds.random_world(2000, seed=35, spread=40.0, p_child=0.0)
    w.group[:] = 1; w.mask[:] = 0xFFFFFFFF
    ow = worlds.oracle_world(oracle, w, camera=False)
    ow.transform_system()
    mn, mx = ow.world_aabbs()
    want = oracle.broadphase_bruteforce(mn, mx, w.group, w.mask)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=100)
    t.run(capi.XFORM | capi.BROADPHASE)
    got, total = t.pairs()
    c = t.counts()
    # the list is kept in 64 shard segments of max_pairs/64 (rounded up): a full segment drops its surplus
    assert total == len(want) > 100 and 0 < len(got) <= 128 and c.pairs_truncated == 1
    wantset = set(map(tuple, want.tolist()))
    assert all(tuple(p) in wantset for p in got.tolist())
    t.close(); ow.close()


def test_full_tick_all_stages_together(oracle):
    from sc_gameengine_amd.tick import camera_view_proj
    w = sw.generate(24, 24, 15)
    dyn = (np.arange(w.n) % 7) == 3
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    ow = worlds.oracle_world(oracle, w)
    t = WorldTick.from_world(w, broadphase=True)
    vp = camera_view_proj(w.camera)
    t.set_view_proj(vp)
    for k in range(3):
        if k:
            ow.nudge_roots_x(0.3); t.nudge_roots_x(0.3)
        ow.transform_system(); ow.culling_system(view_proj=vp)
        t.run(capi.FULL)
        assert np.array_equal(t.visible(), ow.visible())
        assert np.array_equal(t.world_matrices(), ow.world_matrices()[:w.n])
        mn, mx = ow.world_aabbs()
        want = oracle.broadphase_grid(mn[:w.n], mx[:w.n], w.group, w.mask, 16.0)
        got, total = t.pairs()
        assert total == len(want) and np.array_equal(sorted_pairs(got), want)
    t.close(); ow.close()


def test_full_size_1m_pairs_match_cpu_grid(oracle):
    w = sw.config("config3")
    dyn = (np.arange(w.n) % 16) == 4                              # one prop per sector is dynamic
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    t = WorldTick.from_world(w, broadphase=True)
    t.run(FLAGS)
    gmn, gmx = t.world_aabbs()
    want = oracle.broadphase_grid(gmn, gmx, w.group, w.mask, 64.0)   # the oracle's search on the device's boxes
    got, total = t.pairs()
    assert total == len(want) and np.array_equal(sorted_pairs(got), want)
    # second tick: the self-cleaning bins and the parity-switched counters give the same answer again
    t.run(capi.XFORM | capi.BROADPHASE)
    got2, total2 = t.pairs()
    assert total2 == total and np.array_equal(sorted_pairs(got2), want)
    c = t.counts()
    assert c.big_boxes == 0 and c.bin_overflow == 0
    t.close()


def crowded_world(sx, sz, per_sector, seed):
    """per_sector small dynamic boxes in every sector of an sx x sz world (flat, all roots)."""
    rng = np.random.default_rng(seed)
    n = sx * sz * per_sector
    cx, cz = np.meshgrid(np.arange(sx), np.arange(sz))
    base = np.stack([cx.ravel() * 64.0, np.zeros(sx * sz), cz.ravel() * 64.0], axis=1)
    pos = (np.repeat(base, per_sector, axis=0) + rng.uniform([1, 0, 1], [63, 4, 63], (n, 3))).astype(np.float32)
    w = worlds.random_world(n, seed=seed, p_child=0.0, p_no_bounds=0.0, p_no_mesh=0.0)
    w.pos[:] = pos
    w.rot[:] = 0.0
    w.rot[:, 1] = rng.uniform(-3, 3, n).astype(np.float32)
    w.scale[:] = rng.uniform(0.5, 1.5, (n, 3)).astype(np.float32)
    w.bmin[:], w.bmax[:] = -0.5, 0.5
    w.group[:], w.mask[:] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    w.origin, w.sectors = (0, 0), (sx, sz)
    return w


def test_crowded_sectors_overflow_list_equals_cpu_grid_and_stays_cheap(oracle):
    """100k entities at 200 boxes per sector: every sector overflows its 64-record bin three times over.  The overflow records
    stay with their sector (no global list that every sector walks), so the pair set still equals the oracle's and the tick
    costs what (records per sector)^2 / 64 tests cost -- not sectors x list length as a global list would: well under a
    millisecond here, where the same entities spread 50 per sector (no overflow, 16x fewer candidate pairs per sector) take ~15 us."""
    import time
    dense = crowded_world(32, 16, 200, seed=41)                    # 102 400 entities, 512 sectors
    sparse = crowded_world(64, 32, 50, seed=41)                    # the same count over 2 048 sectors
    times = {}
    for name, w in (("dense", dense), ("sparse", sparse)):
        ow = worlds.oracle_world(oracle, w, camera=False)
        ow.transform_system()
        mn, mx = ow.world_aabbs()
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 21)
        t.run(capi.XFORM | capi.BROADPHASE)
        got, total = t.pairs()
        c = t.counts()
        assert total == len(want) and np.array_equal(sorted_pairs(got), want), f"{name}: {total} pairs vs {len(want)}"
        assert c.big_boxes == 0 and c.border_lost == 0 and c.pairs_truncated == 0
        if name == "dense":
            assert c.bin_overflow > 60000 and len(want) > 5000
        else:
            assert c.bin_overflow == 0
        t.mark_dirty(0, w.n)
        for _ in range(3):
            t.run(capi.XFORM | capi.BROADPHASE)
        t.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            t.run(capi.XFORM | capi.BROADPHASE)
        t.sync()
        times[name] = (time.perf_counter() - t0) / 20
        got2, total2 = t.pairs()                                   # the overflow list is consumed and refilled every tick
        assert total2 == len(want) and np.array_equal(sorted_pairs(got2), want)
        t.close(); ow.close()
    assert times["dense"] < 1e-3 and times["dense"] < 40.0 * times["sparse"], times


def test_one_sector_far_beyond_its_capacity_is_counted_not_silent(oracle):
    w = worlds.random_world(3000, seed=43, spread=300.0, p_child=0.0, p_no_bounds=0.0)
    w.pos[:1500] = np.float32([10.0, 0.0, 10.0]) + np.random.default_rng(3).uniform(-8, 8, (1500, 3)).astype(np.float32)   # 1500 boxes in one sector
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 22)
    t.run(capi.XFORM | capi.BROADPHASE)
    c = t.counts()
    assert c.bin_overflow >= 1400 and c.border_lost >= 1500 - 64 - 1024 - 40      # what the sector cannot hold is reported
    t.close()


def test_layers_that_only_pass_across_kinds(oracle):
    """Records that do not pass the filter against their own kind but do against another (4/8 vs 8/4) next to ordinary static
    and dynamic ones: the pair search's broadcast path cannot cover those, the general paths must (and do)."""
    w = worlds.random_world(4000, seed=51, spread=120.0, p_child=0.0, p_no_bounds=0.0)
    rng = np.random.default_rng(6)
    kind = rng.integers(0, 5, w.n)
    w.group[:] = np.choose(kind, [1, 2, 4, 8, 2]).astype(np.uint32)
    w.mask[:] = np.choose(kind, [0xFFFFFFFF, 1, 8, 4, 1]).astype(np.uint32)
    t, ow = gpu_vs_oracle(oracle, w, ticks=2, nudge=0.8)
    got, _ = t.pairs()
    k = {(int(w.group[a]), int(w.group[b])) for a, b in got[:2000]}
    assert (4, 8) in k or (8, 4) in k                      # such pairs really occur
    assert (2, 2) not in k
    t.close(); ow.close()


@pytest.mark.parametrize("dyn_per_sector", [1, 2, 3, 5, 9, 16, 17, 20, 21, 30])
def test_every_count_of_dynamic_records_per_sector(oracle, dyn_per_sector):
    """The pair search picks its path by the number D of records in a bin that pass the filter against their own kind
    (dynamic bodies): D <= 2 broadcast as they lie, D <= 20 lane (record, phase) over the re-ordered bin (G = 64 / D partner
    phases: every quotient is exercised here), above that the general paths.  Sectors hold D dynamic boxes among static
    ones, some sectors nothing but dynamic ones, crowded enough that cast records meet each other and the statics."""
    rng = np.random.default_rng(100 + dyn_per_sector)
    SX = 6
    pos, scl, grp, msk = [], [], [], []
    for sz in range(SX):
        for sx in range(SX):
            ox, oz = (sx - SX // 2) * 64.0, (sz - SX // 2) * 64.0
            d = dyn_per_sector if (sx + sz) % 5 else min(dyn_per_sector + 3, 40)        # a few sectors with a different count
            s = 0 if (sx * 7 + sz) % 6 == 0 else int(rng.integers(0, max(1, 50 - d)))    # some bins hold dynamic boxes only
            k = d + s
            # clustered so that boxes really overlap: a third of them around two hot spots, the rest anywhere in the sector
            hot = rng.uniform(8, 56, (2, 2))
            xz = np.where(rng.random((k, 1)) < 0.35, hot[rng.integers(0, 2, k)] + rng.normal(0, 2.0, (k, 2)), rng.uniform(1, 63, (k, 2)))
            pos.append(np.column_stack([ox + xz[:, 0], rng.uniform(0.3, 1.5, k), oz + xz[:, 1]]))
            scl.append(rng.uniform(0.5, 4.0, (k, 3)))
            grp.append(np.r_[np.full(d, sw.GROUP_DYNAMIC), np.full(s, sw.GROUP_STATIC)])
            msk.append(np.r_[np.full(d, sw.MASK_ALL), np.full(s, sw.MASK_STATIC)])
    w = worlds.random_world(sum(len(p) for p in pos), seed=7, spread=10.0, p_child=0.0, p_no_bounds=0.0)
    w.pos[:] = np.concatenate(pos).astype(np.float32); w.scale[:] = np.concatenate(scl).astype(np.float32); w.rot[:] = 0.0
    w.rot[:, 1] = rng.uniform(0, 6.28, w.n).astype(np.float32)
    w.group[:] = np.concatenate(grp).astype(np.uint32); w.mask[:] = np.concatenate(msk).astype(np.uint32)
    perm = rng.permutation(w.n)                      # dense order (= record order in a bin) unrelated to the kind
    for a in ("pos", "scale", "rot", "group", "mask"):
        getattr(w, a)[:] = getattr(w, a)[perm]
    t, ow = gpu_vs_oracle(oracle, w, ticks=2, nudge=0.9)
    c = t.counts()
    assert c.pairs > 20 * dyn_per_sector and c.pairs_truncated == 0 and c.border_lost == 0
    t.close(); ow.close()


def test_overflow_list_holds_every_copy_of_every_box(oracle):
    """Two thousand boxes around one corner where four sectors meet: most of them straddle an edge, so there are MORE overflow
    records than entities (up to four copies of a box, 64 per bin kept).  The list is sized for that (4 x capacity); it used to
    hold one record per entity, and what did not fit was dropped without a count (found by tools/stress_broadphase.py)."""
    w = worlds.random_world(1956, seed=54, spread=20.0, max_depth=0, p_child=0.0, p_no_bounds=0.05)
    w.group[:] = sw.GROUP_DYNAMIC; w.mask[:] = sw.MASK_ALL
    t, ow = gpu_vs_oracle(oracle, w, ticks=2, nudge=0.8, max_pairs=1 << 20)
    c = t.counts()
    assert c.bin_overflow > w.n and c.border_lost == 0 and c.pairs_truncated == 0 and c.pairs > 10000
    t.close(); ow.close()


@pytest.mark.parametrize("period,variant", [("3", "0"), ("64", "0"), ("64", "2")])
def test_remembered_bin_slots_over_many_ticks_with_boxes_changing_sectors(oracle, monkeypatch, period, variant):
    """Home slots: records go to the slot they reserved at the last learn tick while their box keeps its
    sector.  Forty ticks of a world whose roots move 2.3 m per tick -- boxes enter and leave sectors all the time, copies in
    neighbouring sectors appear and lapse, deep chains and big boxes take the reserving path throughout -- with a learn tick
    every 3 ticks, every 64 (so nearly every box ends up away from home), and with the slots switched off (SC_TICK_VARIANT
    bit 1): the pair set must be the oracle's on every tick either way."""
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", period)
    monkeypatch.setenv("SC_TICK_VARIANT", variant)
    w = worlds.random_world(4000, seed=77, spread=220.0, max_depth=5)
    w.bmin[:25] *= 40.0; w.bmax[:25] *= 40.0                                # a few big boxes (never binned)
    w.pos[100:260] = np.float32([40.0, 0.0, -30.0]) + np.random.default_rng(5).uniform(-9, 9, (160, 3)).astype(np.float32)   # one crowded sector: bin overflow
    t, ow = gpu_vs_oracle(oracle, w, brute=False, ticks=40, nudge=2.3, max_pairs=1 << 20)
    c = t.counts()
    assert c.pairs > 200 and c.pairs_truncated == 0 and c.bin_overflow > 50
    t.close(); ow.close()


def test_remembered_slots_survive_uploads_appends_and_removals(oracle, monkeypatch):
    """Whatever changes the world's shape -- positions uploaded, layers changed, entities appended and removed -- between
    ticks: the pair set stays the oracle's (layers, appends and removals force a learn tick; positions do not need one)."""
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", "1000")
    w = worlds.random_world(2500, seed=78, spread=180.0, max_depth=2)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 19)
    rng = np.random.default_rng(9)

    def check():
        ow.transform_system(); t.run(FLAGS)
        mn, mx = ow.world_aabbs()
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        got, total = t.pairs()
        assert total == len(want) and np.array_equal(sorted_pairs(got), want)
    check(); check()
    # positions of a third of the roots jump across the world: their records leave home, no learn tick
    roots = np.flatnonzero(w.parent < 0)
    mv = rng.choice(roots, len(roots) // 3, replace=False)
    w.pos[mv] = rng.uniform(-180, 180, (len(mv), 3)).astype(np.float32)
    ow.set_local_positions(np.arange(w.n, dtype=np.uint32), w.pos)       # (setLocalPosition marks dirty: every entity, on both sides)
    t.upload_positions(0, w.pos)
    check(); check()
    # layers flip for half of the world
    flip = rng.random(w.n) < 0.5
    w.group[flip], w.mask[flip] = 1, 0xFFFFFFFF
    t.upload_layers(0, w.group, w.mask)
    check(); check()
    t.close(); ow.close()


def test_a_world_that_cannot_pair_is_only_a_hint_for_the_launch_shape(oracle):
    """The host sizes the pair role's grid by whether ANY two uploaded layer words admit a pair (an all-static city does not: its
    pair role is a sweep over counters).  A hint only: the same context finds every pair once layers that can meet are uploaded,
    and none again when they are taken back."""
    w, d = _static_city_with_wanderers(6000, 0, seed=91)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 19)

    def check(expect_some):
        for _ in range(3):
            ow.nudge_roots_x(0.5); t.nudge_roots_x(0.5)
            ow.transform_system(); t.run(FLAGS)
            mn, mx = ow.world_aabbs()
            want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
            got, total = t.pairs()
            assert total == len(want) and np.array_equal(sorted_pairs(got), want)
            assert (len(want) > 0) == expect_some
    check(False)
    roots = np.flatnonzero(w.parent < 0)
    dyn = roots[:400]
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    t.upload_layers(0, w.group, w.mask)
    check(True)
    w.group[dyn], w.mask[dyn] = sw.GROUP_STATIC, sw.MASK_STATIC
    t.upload_layers(0, w.group, w.mask)
    check(False)
    t.close(); ow.close()


def _static_city_with_wanderers(n, dyn, seed, spread=440.0):
    """Static props (group 2 / mask 1) everywhere, a few dynamic bodies (1 / all) among them: most bins admit no pair."""
    w = worlds.random_world(n, seed=seed, spread=spread, max_depth=1, p_child=0.2, p_no_bounds=0.02)
    w.group[:] = sw.GROUP_STATIC; w.mask[:] = sw.MASK_STATIC
    roots = np.flatnonzero(w.parent < 0)
    w.pos[roots, 1] *= np.float32(0.02)                                  # a flat city: boxes meet in y
    d = roots[:dyn]
    w.group[d] = sw.GROUP_DYNAMIC; w.mask[d] = sw.MASK_ALL
    w.bmin[d] = np.float32([-3.0, -1.0, -3.0]); w.bmax[d] = np.float32([3.0, 2.0, 3.0])
    return w, d


@pytest.mark.parametrize("variant", ["0", "32"])
def test_unwritten_bins_are_rebuilt_when_a_body_from_elsewhere_arrives(oracle, monkeypatch, variant):
    """Lazy records: the remembered slots of a bin whose own records admit no pair -- static props only --
    are not written between learn ticks; when a dynamic body drives into such a sector, the wave that searches it rebuilds the
    props' records from their world matrices.  Dynamic bodies jump to new places every other tick, every root drifts 1.7 m a
    tick (so props leave their sectors too and their slots lapse), one learn tick at the start: the pair set is the
    oracle's on every tick.  Variant 32 writes every slot every tick (the A/B switch): same pairs."""
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", "1000")
    monkeypatch.setenv("SC_TICK_VARIANT", variant)
    w, dyn = _static_city_with_wanderers(7000, 60, seed=91)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 18)
    rng = np.random.default_rng(17)
    seen = 0
    for k in range(24):
        if k:
            ow.nudge_roots_x(1.7); t.nudge_roots_x(1.7)
        if k % 2 == 1:
            pos = t.positions()
            pos[dyn] = rng.uniform(-450, 450, (len(dyn), 3)).astype(np.float32) * np.float32([1, 0.01, 1])
            ow.set_local_positions(np.arange(w.n, dtype=np.uint32), pos)
            t.upload_positions(0, pos)
        ow.transform_system(); t.run(FLAGS)
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        got, total = t.pairs()
        assert total == len(want), f"tick {k}: {total} pairs, oracle {len(want)}"
        assert np.array_equal(sorted_pairs(got), want), f"tick {k}"
        seen += total
        if k == 0:
            first = t.bin_stats()
    assert seen > 200 and t.counts().pairs_truncated == 0
    bs = t.bin_stats()
    assert bs["learn_ticks"] == 1 and bs["remembered_slots"] == first["remembered_slots"] > w.n // 2
    # after the learn tick some bins are written on every tick (a dynamic body lives there; ring sectors), the others are not;
    # every bin a dynamic body jumped into since was rebuilt once and is written from then on
    assert 0 < first["written_every_tick"] < first["remembered_slots"] * 3 // 4
    if variant == "0":
        assert bs["lazy_last_tick"] and first["written_every_tick"] < bs["written_every_tick"] <= bs["remembered_slots"]
    else:
        assert not bs["lazy_last_tick"] and bs["written_every_tick"] == first["written_every_tick"]
    t.close(); ow.close()


def test_unwritten_bins_and_a_big_box_that_appears(oracle, monkeypatch):
    """A big box (tested against every bin's primary records) appears between learn ticks: on its first tick the bins were
    not written (nobody knew), so every wave rebuilds what it needs; from the next tick on the fused kernel writes every record
    again.  Then it shrinks back.  Pairs equal the oracle's throughout, and ray queries (which read the bins) switch the
    lazy records off for their tick."""
    from tests.test_gpu_rays import compare
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", "1000")
    w, dyn = _static_city_with_wanderers(5000, 30, seed=92, spread=430.0)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 19)
    big = int(dyn[0])
    small = (w.bmin.copy(), w.bmax.copy())
    state = {}

    def check(flags=FLAGS):
        ow = worlds.oracle_world(oracle, w, camera=False)      # (nothing moves in this test: the oracle's world is rebuilt from w)
        ow.transform_system(); t.run(flags)
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        got, total = t.pairs()
        assert total == len(want) and np.array_equal(sorted_pairs(got), want)
        state["aabbs"] = (mn, mx)
        ow.close()
        return total
    check(); check(); base = check()
    w.bmin[big] = np.float32([-400, -5, -400]); w.bmax[big] = np.float32([400, 5, 400])
    t.upload_bounds(0, w.bmin, w.bmax, w.has_bounds)
    grown = check()
    assert grown > base + 500                                    # the big dynamic box meets the props under it
    assert check() == grown and check() == grown
    w.bmin[:], w.bmax[:] = small
    t.upload_bounds(0, w.bmin, w.bmax, w.has_bounds)
    assert check() == base and check() == base and check() == base
    # a tick with ray queries reads the bins: every record is there
    rays = (np.float32([[0, 50, 0], [100, 50, -80], [-200, 40, 150]]), np.float32([[0, -1, 0], [0.6, -0.8, 0], [0, -1, 0]]),
            np.float32([200, 300, 200]), np.uint32([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF]))
    t.set_ray_queries(*rays)
    assert check(FLAGS | capi.RAYS) == base
    mn, mx = state["aabbs"]
    compare(t.ray_hits(), oracle.raycast_boxes(mn, mx, w.group, w.mask, *rays))
    assert check() == base
    t.close()


def test_records_of_boxes_that_did_not_move_stay_in_their_slots(oracle, monkeypatch):
    """Between learn ticks an entity whose matrix was not rebuilt leaves its bin slots alone -- they hold this very record.
    Only ranges of the world move from tick to tick (the rest is clean), some ticks run the transforms WITHOUT the broadphase
    (the bins do not follow: the next broadphase tick must write every record), bounds change under clean entities: the pair
    set is the oracle's on every broadphase tick."""
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", "1000")
    w = worlds.random_world(6000, seed=93, spread=260.0, max_depth=2, p_child=0.3, p_no_bounds=0.03)
    roots = np.flatnonzero(w.parent < 0)
    w.pos[roots, 1] *= np.float32(0.05)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 20)
    rng = np.random.default_rng(23)
    pos = w.pos.copy()
    seen = 0
    for k in range(26):
        xform_only = k in (7, 8, 15)
        if k:
            a = int(rng.integers(0, w.n - 700)); b = a + int(rng.integers(50, 700))
            sel = np.arange(a, b, dtype=np.uint32)
            mv = sel[w.parent[sel] < 0]
            pos[mv] += rng.uniform(-6, 6, (len(mv), 3)).astype(np.float32) * np.float32([1, 0.02, 1])
            ow.set_local_positions(sel, pos[sel])
            t.upload_positions(a, pos[a:b])
        if k == 19:                                      # boxes change under entities that do not move
            w.bmin[1000:3000] *= np.float32(1.5); w.bmax[1000:3000] *= np.float32(1.5)
            ow.close(); ow = worlds.oracle_world(oracle, w, camera=False)
            ow.set_local_positions(np.arange(w.n, dtype=np.uint32), pos)
            t.upload_bounds(0, w.bmin, w.bmax, w.has_bounds)
            t.mark_dirty(0, w.n)
        ow.transform_system()
        if xform_only:
            t.run(capi.XFORM)
            continue
        t.run(FLAGS)
        mn, mx = ow.world_aabbs(); gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx), f"tick {k}"
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        got, total = t.pairs()
        assert total == len(want), f"tick {k}: {total} pairs, oracle {len(want)}"
        assert np.array_equal(sorted_pairs(got), want), f"tick {k}"
        seen += total
    assert seen > 1000
    t.close(); ow.close()


def test_a_host_that_works_between_the_halves_of_a_split_tick_does_not_change_its_pairs(oracle, monkeypatch):
    """include/sc_tick.h offers the split flow -- scTickRun(.. | SC_TICK_SPLIT_PAIRS), the caller's own exchange, scTickRunPairs -- to
    hosts that "interleave other work".  With lazy records the pair half would rebuild unwritten bins from the world matrices,
    bounds and layers as they stand when it RUNS; a host that uploads matrices or changes bounds in between would then change
    tick t's pair set.  So the library writes every record when the host owns
    the gap (the pair half reads tick t's snapshot only): dynamic bodies jump into prop-only sectors every tick, the host
    scrambles the world between the halves, and the pairs are still the oracle's pairs of tick t."""
    monkeypatch.setenv("SC_TICK_HOME_PERIOD", "1000")
    w, dyn = _static_city_with_wanderers(6000, 50, seed=23)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 18)
    rng = np.random.default_rng(29)
    flags = FLAGS | capi.SPLIT_PAIRS
    seen = 0
    for k in range(8):
        pos = t.positions()
        if k:
            pos[dyn] = rng.uniform(-430, 430, (len(dyn), 3)).astype(np.float32) * np.float32([1, 0.01, 1])
            ow.set_local_positions(np.arange(w.n, dtype=np.uint32), pos)
            t.upload_positions(0, pos)
        ow.transform_system(); t.run(flags)
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
        assert not t.bin_stats()["lazy_last_tick"]                      # the host owns the gap: nothing is left to a rebuild
        # ---- the host's "other work" between the halves: none of it belongs to tick t
        t.sync()
        m = t.world_matrices()
        m[:, 12:15] += rng.uniform(-300, 300, (w.n, 3)).astype(np.float32)          # every box somewhere else
        t.upload_world_matrices(0, m)
        if k % 2:
            t.upload_bounds(0, w.bmin * np.float32(3.0), w.bmax * np.float32(3.0), w.has_bounds)
        t.run_pairs()
        got, total = t.pairs()
        assert total == len(want), f"tick {k}: {total} pairs, oracle {len(want)}"
        assert np.array_equal(sorted_pairs(got), want), f"tick {k}"
        seen += total
        # ---- and back: the world as the oracle has it
        t.upload_bounds(0, w.bmin, w.bmax, w.has_bounds)
        t.upload_positions(0, pos); t.mark_dirty(0, w.n)
    assert seen > 100
    t.close(); ow.close()
