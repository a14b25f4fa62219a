"""Sector data on the host side of the tick: the .scsector reader and the spawn records the streamer
builds from it (WorldPartition::readSectorFile, src/engine/world/sc_world_partition.cpp:696-730).

All parsing is done by the C ABI (scTickSectorParse in libsc_tick.so, after tools/shared/
world_format.cpp:185-338); this module only shapes the result as numpy arrays.  No oracle import.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi

NAME_MAX = 32          # Name::kMax, sc_ecs.h:148-152
INSTANCE_NAME_MAX = 64  # kInstanceNameMax, world_format.h:14


@dataclass
class SectorData:
    version: int
    sector: tuple
    instances: int          # records in the file
    lanes: int
    lane_points: int
    spawners: int
    colliders: int
    truncated: bool
    id: np.ndarray
    model_id: np.ndarray
    mesh_id: np.ndarray
    material_id: np.ndarray
    albedo_texture_id: np.ndarray
    material_flags: np.ndarray
    tags: np.ndarray
    pos: np.ndarray         # [n, 3] float32: Instance::transform
    rot: np.ndarray
    scale: np.ndarray
    name64: np.ndarray      # [n, 64] uint8, NUL-terminated

    def names(self):
        return [bytes(r).split(b"\0", 1)[0].decode("latin-1") for r in self.name64]


def _parse(call):
    lib = capi.load()
    info = capi.SectorInfo()
    if not call(lib, C.byref(info), None):
        return None
    n = min(info.instances, (1 << 24) - 1)      # SC_TICK_MAX_ENTITIES: no context could take more
    u64 = [np.zeros(n, np.uint64) for _ in range(5)]
    u32 = [np.zeros(n, np.uint32) for _ in range(2)]
    f3 = [np.zeros((n, 3), np.float32) for _ in range(3)]
    # the reference's defaults for records the data does not reach (world_format.h:31-36)
    f3[2][:] = 1.0
    names = np.zeros((n, INSTANCE_NAME_MAX), np.uint8)
    out = capi.SectorInstances()
    out.capacity = n
    for field, arr in zip(("id", "model_id", "mesh_id", "material_id", "albedo_texture_id"), u64):
        setattr(out, field, arr.ctypes.data_as(capi.U64P))
    out.material_flags = u32[0].ctypes.data_as(capi.U32P)
    out.tags = u32[1].ctypes.data_as(capi.U32P)
    out.pos3, out.rot3, out.scale3 = (a.ctypes.data_as(capi.F32P) for a in f3)
    out.name64 = names.ctypes.data_as(C.c_void_p)
    if n and not call(lib, C.byref(info), C.byref(out)):
        return None
    return SectorData(info.version, (info.sector_x, info.sector_z), info.instances, info.lanes, info.lane_points,
                      info.spawners, info.colliders, bool(info.truncated), *u64, *u32, *f3, names)


def parse_sector(data: bytes):
    """SectorFile from bytes; None when the data is not a sector file (ReadSectorFile returning false)."""
    buf = (C.c_char * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    return _parse(lambda lib, info, out: lib.scTickSectorParse(C.cast(buf, C.c_void_p), len(data), info, out))


def read_sector_file(path: str):
    return _parse(lambda lib, info, out: lib.scTickSectorReadFile(path.encode(), info, out))


def hash_asset_path(path):
    return int(capi.load().scTickHashAssetPath(None if path is None else path.encode()))


def sector_path(world_root, x, z):
    buf = C.create_string_buffer(4096)
    capi.load().scTickSectorPath(None if world_root is None else world_root.encode(), int(x), int(z), buf, 4096)
    return buf.value.decode()


def spawn_names(sec: SectorData):
    """SpawnRecord::name as readSectorFile fills it (:710-713): the instance name, else "Inst_<id>", cut to Name::kMax-1."""
    out = []
    for nm, ident in zip(sec.names(), sec.id):
        text = nm if nm else f"Inst_{int(ident)}"
        out.append(text[:NAME_MAX - 1])
    return out
