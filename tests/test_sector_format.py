"""The product's .scsector reader (scTickSectorParse, libsc_tick.so) against the REAL reference.

tests/golden/sectors/*.scsector were written by the reference's own WriteSectorFile and
tests/golden/sc_sector_ref.json holds what its own ReadSectorFile returns for each of them, variants
included (oracle/make_golden_sectors.py; tools/shared/world_format.cpp compiled unmodified).  Host
code only: runs without a GPU.
"""
import json
import os

import numpy as np
import pytest

from sc_gameengine_amd import sectors

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF = json.load(open(os.path.join(GOLD, "sc_sector_ref.json")))
CASES = sorted(REF["cases"])


def _check(sec, want):
    assert sec.version == want["version"]
    assert list(sec.sector) == want["sector"]
    assert [sec.instances, sec.lanes, sec.spawners, sec.colliders] == want["counts"]
    assert sec.lane_points == want["lane_points"]
    n = len(want["id"])
    assert sec.instances == n
    for field in ("id", "model_id", "mesh_id", "material_id", "albedo_texture_id", "material_flags", "tags"):
        assert getattr(sec, field).tolist() == want[field], field
    trs = np.concatenate([sec.pos, sec.rot, sec.scale], axis=1).astype(np.float32) if n else np.zeros((0, 9), np.float32)
    assert trs.view(np.uint32).tolist() == want["trs_bits"]          # bit patterns, not values
    assert [bytes(r).hex() for r in sec.name64] == want["name_hex"]


@pytest.mark.parametrize("case", CASES)
def test_reader_matches_reference_reader(case):
    want = REF["cases"][case]
    path = os.path.join(GOLD, "sectors", case + ".scsector")
    assert os.path.getsize(path) == want["bytes"]
    from_file = sectors.read_sector_file(path)
    from_bytes = sectors.parse_sector(open(path, "rb").read())
    if not want["ok"]:
        assert from_file is None and from_bytes is None
        return
    _check(from_file, want)
    _check(from_bytes, want)


def test_truncation_is_reported_only_when_data_ran_out():
    flag = {c: sectors.read_sector_file(os.path.join(GOLD, "sectors", c + ".scsector")) for c in CASES if REF["cases"][c]["ok"]}
    cut = {c for c, s in flag.items() if s.truncated}
    assert cut == {"cut_mid_record", "cut_in_count", "cut_in_chunk_header", "cut_in_file_header"}


def test_missing_file_and_null_arguments():
    assert sectors.read_sector_file(os.path.join(GOLD, "sectors", "no_such.scsector")) is None
    assert sectors.parse_sector(b"") is None
    from sc_gameengine_amd import capi
    lib = capi.load()
    assert lib.scTickSectorParse(None, 0, None, None) == 0
    assert lib.scTickSectorReadFile(None, None, None) == 0


def test_partial_capacity_keeps_the_first_records():
    import ctypes as C
    from sc_gameengine_amd import capi
    lib = capi.load()
    data = open(os.path.join(GOLD, "sectors", "v4_sixteen.scsector"), "rb").read()
    ids = np.full(8, 0xFFFFFFFFFFFFFFFF, np.uint64)
    out = capi.SectorInstances()
    out.capacity = 5
    out.id = ids.ctypes.data_as(capi.U64P)            # every other array stays NULL
    info = capi.SectorInfo()
    assert lib.scTickSectorParse(data, len(data), C.byref(info), C.byref(out)) == 1
    assert info.instances == 16
    assert ids[:5].tolist() == REF["cases"]["v4_sixteen"]["id"][:5]
    assert (ids[5:] == 0xFFFFFFFFFFFFFFFF).all()


def test_hash_asset_path_matches_reference():
    for path, want in REF["hash_asset_path"].items():
        assert sectors.hash_asset_path(path) == want, path
    assert sectors.hash_asset_path(None) == 0


def test_sector_path_matches_reference():
    for key, want in REF["sector_path"].items():
        root, x, z = json.loads(key)
        assert sectors.sector_path(root, x, z) == want


def test_spawn_names_follow_read_sector_file():
    sec = sectors.read_sector_file(os.path.join(GOLD, "sectors", "v4_full.scsector"))
    names = sectors.spawn_names(sec)
    assert names[0] == "prop_v4_full_0"
    assert names[2] == f"Inst_{int(sec.id[2])}"       # unnamed instance
    assert names[3] == "x" * 31                       # 63 kept by the reader, 31 by Name::kMax
