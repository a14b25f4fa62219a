"""ctypes binding of libsc_tick.so -- exactly the symbols include/sc_tick.h declares.

There is no CPU fallback: if the HIP library is missing or no AMD GPU is present, loading or
context creation raises.  Nothing in this package imports oracle/.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_TICK_LIB") or os.path.join(HERE, "libsc_tick.so")      # SC_TICK_LIB: A/B builds (tools/)

# scTickRun flags (include/sc_tick.h)
XFORM, CULL, BROADPHASE, CULLED_LIST, DRAWS, DENSE_AABBS, SPLIT_PAIRS, SORT_DRAWS, RAYS, PRODUCE_NEXT = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
FULL = XFORM | CULL | BROADPHASE
K_XFORM_CULL, K_COMPACT, K_PAIRS, K_NUDGE, K_COUNT = 0, 1, 2, 3, 4
NO_PARENT = -1
COMM_ID_BYTES = 128
HAVE_PAIR_SEARCH = True       # flipped when the broadphase pair kernels are in the library

F32P = C.POINTER(C.c_float)
U32P = C.POINTER(C.c_uint32)
I32P = C.POINTER(C.c_int32)
U8P = C.POINTER(C.c_uint8)
U64P = C.POINTER(C.c_uint64)


class ContextDesc(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("capacity", C.c_uint32),
                ("tile_origin_x", C.c_int32), ("tile_origin_z", C.c_int32),
                ("tile_sectors_x", C.c_uint32), ("tile_sectors_z", C.c_uint32),
                ("sector_size", C.c_float), ("max_pairs", C.c_uint32),
                ("max_draws_budget", C.c_uint32), ("reserved", C.c_uint32)]


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "entities", "renderables_total", "visible", "culled", "pairs", "pairs_truncated",
        "draws_emitted", "draws_dropped", "max_depth", "unreachable", "bin_overflow", "big_boxes", "draws_sorted", "border_lost", "relinks",
        "vocabulary_violations")]


class DrawItem(C.Structure):
    _fields_ = [("dense_index", C.c_uint32), ("mesh_id", C.c_uint32), ("material_id", C.c_uint32),
                ("pad", C.c_uint32), ("model", C.c_float * 16)]


class RayHit(C.Structure):
    _fields_ = [("hit", C.c_uint32), ("id", C.c_uint32), ("distance", C.c_float), ("position", C.c_float * 3),
                ("normal", C.c_float * 3), ("layer", C.c_uint32), ("pad", C.c_uint32 * 2)]


class Frame(C.Structure):
    _fields_ = [("tick", C.c_uint64), ("renderables_total", C.c_uint32), ("visible", C.c_uint32), ("culled", C.c_uint32),
                ("draws_emitted", C.c_uint32), ("draws_dropped", C.c_uint32), ("draws_sorted", C.c_uint32),
                ("visible_in_buffer", C.c_uint32), ("draws_in_buffer", C.c_uint32),
                ("visible_indices", U32P), ("draws", C.POINTER(DrawItem))]


class LaneGraph(C.Structure):
    _fields_ = [("segments", C.c_uint32), ("nodes", C.c_uint32), ("connections", C.c_uint32),
                ("seg_start3", F32P), ("seg_dir3", F32P), ("seg_length", F32P), ("seg_active", U8P), ("seg_end_node", U32P),
                ("seg_speed_limit", F32P), ("node_pos3", F32P), ("node_conn_offset", U32P), ("node_conn", U32P)]


class TierParams(C.Structure):
    _fields_ = [("tier_a_enter", C.c_float), ("tier_a_exit", C.c_float), ("tier_b_enter", C.c_float), ("tier_b_exit", C.c_float),
                ("max_physics", C.c_uint32), ("max_kinematic", C.c_uint32)]


class TierCounts(C.Structure):
    _fields_ = [("physics", C.c_uint32), ("kinematic", C.c_uint32), ("on_rails", C.c_uint32), ("total", C.c_uint32)]


class CommInfo(C.Structure):
    _fields_ = [("has_communicator", C.c_uint32), ("world_size", C.c_uint32), ("rank", C.c_uint32), ("rccl_version", C.c_uint32),
                ("neighbour_mask", C.c_uint32), ("peer_rank", C.c_int32 * 8), ("operations_per_group", C.c_uint32),
                ("pipeline_depth", C.c_uint32), ("border_records_per_sector", C.c_uint32), ("bytes_sent_per_step", C.c_uint64),
                ("host_steps", C.c_uint64), ("host_tick_half_us", C.c_double), ("host_pair_half_us", C.c_double)]


class SectorInfo(C.Structure):
    _fields_ = [("version", C.c_uint32), ("sector_x", C.c_int32), ("sector_z", C.c_int32), ("instances", C.c_uint32),
                ("lanes", C.c_uint32), ("lane_points", C.c_uint32), ("spawners", C.c_uint32), ("colliders", C.c_uint32),
                ("truncated", C.c_uint32)]


class SectorInstances(C.Structure):
    _fields_ = [("capacity", C.c_uint32),
                ("id", U64P), ("model_id", U64P), ("mesh_id", U64P), ("material_id", U64P), ("albedo_texture_id", U64P),
                ("material_flags", U32P), ("tags", U32P),
                ("pos3", F32P), ("rot3", F32P), ("scale3", F32P), ("name64", C.c_void_p)]


# every symbol of include/sc_tick.h: name -> (restype, argtypes)
_CTX = C.c_void_p
SYMBOLS = {
    "scTickGetApiVersion": (C.c_uint32, []),
    "scTickCreateContext": (_CTX, [C.POINTER(ContextDesc)]),
    "scTickDestroyContext": (None, [_CTX]),
    "scTickGetLastError": (C.c_char_p, [_CTX]),
    "scTickSetEntityCount": (C.c_int, [_CTX, C.c_uint32]),
    "scTickUploadLocals": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P, F32P, F32P, U8P]),
    "scTickUploadPositions": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickUploadBounds": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P, F32P, U8P]),
    "scTickUploadRenderMeshes": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U8P, U32P, U32P]),
    "scTickUploadLayers": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U32P, U32P]),
    "scTickSetTopology": (C.c_int, [_CTX, I32P, C.c_uint32]),
    "scTickMarkDirty": (C.c_int, [_CTX, C.c_uint32, C.c_uint32]),
    "scTickMarkDirtyIndices": (C.c_int, [_CTX, U32P, C.c_uint32]),
    "scTickUploadWorldMatrices": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickSetDirtyFlags": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U8P]),
    "scTickSetDrawBudget": (C.c_int, [_CTX, C.c_uint32]),
    "scTickSetDrawSortTable": (C.c_int, [_CTX, U8P, C.c_uint32, C.c_uint32]),
    "scTickSetTile": (C.c_int, [_CTX, C.c_uint32, C.c_uint32]),
    "scTickSetTileGrid": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "scTickSetBorderCapacity": (C.c_int, [_CTX, C.c_uint32]),
    "scTickBorderBytes": (C.c_uint32, [_CTX, C.c_uint32]),
    "scTickBindBorderBuffers": (C.c_int, [_CTX, C.c_uint32, C.c_void_p, C.c_void_p]),
    "scTickRunPairs": (C.c_int, [_CTX]),
    "scTickSetPairsStream": (C.c_int, [_CTX, C.c_void_p]),
    "scTickBindBorderBuffersParity": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "scTickGetBorderBuffer": (C.c_void_p, [_CTX, C.c_uint32, C.c_uint32, C.c_int]),
    "scTickSetStream": (C.c_int, [_CTX, C.c_void_p, C.c_int]),
    "scTickCommGetUniqueId": (C.c_int, [U8P]),
    "scTickCommInit": (C.c_int, [_CTX, U8P, C.c_uint32, C.c_uint32]),
    "scTickCommSetPeers": (C.c_int, [_CTX, I32P]),
    "scTickCommDestroy": (C.c_int, [_CTX]),
    "scTickSetPipelined": (C.c_int, [_CTX, C.c_int]),
    "scTickGetCommInfo": (C.c_int, [_CTX, C.POINTER(CommInfo)]),
    "scTickGatherVisibleCounts": (C.c_int, [_CTX, U32P, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "scTickGetBinStats": (C.c_int, [_CTX, U32P]),
    "scTickGetLearnTicks": (C.c_int, [_CTX, U32P]),
    "scTickSetWorldLayers": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, C.c_int]),
    "scTickResetHostTimes": (C.c_int, [_CTX]),
    "scTickTileStep": (C.c_int, [_CTX, C.c_uint32]),
    "scTickExchangeBorders": (C.c_int, [_CTX]),
    "scTickUploadMovers": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U8P, F32P, F32P, F32P]),
    "scTickAdvanceMovers": (C.c_int, [_CTX, C.c_float]),
    "scTickReadMoverVelocities": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickSetFrameProducer": (C.c_int, [_CTX, C.c_uint32, C.c_float]),
    "scTickSetFrameReadback": (C.c_int, [_CTX, C.c_uint32, C.c_uint32]),
    "scTickAcquireFrame": (C.c_int, [_CTX, C.c_uint32, C.POINTER(Frame)]),
    "scTickSetLaneGraph": (C.c_int, [_CTX, C.POINTER(LaneGraph)]),
    "scTickSetLaneActive": (C.c_int, [_CTX, U32P, C.c_uint32, C.c_int]),
    "scTickUploadTrafficAgents": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U8P, U32P, F32P, F32P, U8P, F32P]),
    "scTickReadTrafficAgents": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U32P, F32P, F32P, U8P]),
    "scTickSetTrafficSpeedMultiplier": (C.c_int, [_CTX, C.c_float]),
    "scTickSetTrafficSensors": (C.c_int, [_CTX, C.c_int, C.c_float, C.c_float]),
    "scTickReadTrafficBrakes": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickUploadTrafficSensors": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P, F32P]),
    "scTickReadTrafficSensors": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P, U8P]),
    "scTickSelectTrafficTiers": (C.c_int, [_CTX, F32P, C.POINTER(TierParams), C.POINTER(TierCounts)]),
    "scTickSelectTrafficDespawns": (C.c_int, [_CTX, F32P, C.c_uint32, U32P, C.c_uint32, U32P]),
    "scTickSetViewProj": (C.c_int, [_CTX, F32P]),
    "scTickSetFrustumPlanes": (C.c_int, [_CTX, F32P, C.c_int]),
    "scTickGetFrustumPlanes": (C.c_int, [_CTX, F32P, C.POINTER(C.c_int)]),
    "scTickSetFreezeCulling": (C.c_int, [_CTX, C.c_int]),
    "scTickRun": (C.c_int, [_CTX, C.c_uint32]),
    "scTickSynchronize": (C.c_int, [_CTX]),
    "scTickNudgeRootsX": (C.c_int, [_CTX, C.c_float]),
    "scTickGetCounts": (C.c_int, [_CTX, C.POINTER(Counts)]),
    "scTickReadVisible": (C.c_int, [_CTX, U32P, C.c_uint32, U32P]),
    "scTickReadCulled": (C.c_int, [_CTX, U32P, C.c_uint32, U32P]),
    "scTickReadVisibilityBits": (C.c_int, [_CTX, U64P, C.c_uint32]),
    "scTickReadWorldMatrices": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickReadWorldMatricesIndexed": (C.c_int, [_CTX, U32P, C.c_uint32, F32P]),
    "scTickReadDirty": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, U8P]),
    "scTickReadPositions": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P]),
    "scTickReadWorldAabbs": (C.c_int, [_CTX, C.c_uint32, C.c_uint32, F32P, F32P]),
    "scTickReadPairs": (C.c_int, [_CTX, U32P, C.c_uint32, U32P]),
    "scTickReadDraws": (C.c_int, [_CTX, C.POINTER(DrawItem), C.c_uint32, U32P]),
    "scTickSetRayQueries": (C.c_int, [_CTX, C.c_uint32, F32P, F32P, F32P, U32P]),
    "scTickReadRayHits": (C.c_int, [_CTX, C.POINTER(RayHit), C.c_uint32, U32P]),
    "scTickQueryOccupied": (C.c_int, [_CTX, C.c_uint32, F32P, F32P, U32P, U8P]),
    "scTickSetProfiling": (C.c_int, [_CTX, C.c_int]),
    "scTickSetProfilingKernels": (C.c_int, [_CTX, C.c_uint32]),
    "scTickGetKernelTimes": (C.c_int, [_CTX, C.c_uint32, F32P, C.c_uint32, U32P]),
    "scTickSetGraphMode": (C.c_int, [_CTX, C.c_int]),
    "scTickGetStream": (C.c_void_p, [_CTX]),
    "scTickSectorParse": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(SectorInfo), C.POINTER(SectorInstances)]),
    "scTickSectorReadFile": (C.c_int, [C.c_char_p, C.POINTER(SectorInfo), C.POINTER(SectorInstances)]),
    "scTickHashAssetPath": (C.c_uint64, [C.c_char_p]),
    "scTickSectorPath": (C.c_uint32, [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_uint32]),
    "scTickAppendEntities": (C.c_int, [_CTX, C.c_uint32, F32P, F32P, F32P, F32P, F32P, U32P, U32P, U32P, U32P, I32P, U32P]),
    "scTickRemoveEntities": (C.c_int, [_CTX, U32P, C.c_uint32, U32P, U32P, U32P]),
    "scTickHostMat4Mul": (C.c_int, [F32P, F32P, F32P]),
    "scTickHostMat4Trs": (C.c_int, [F32P, F32P, F32P, F32P]),
    "scTickHostMat4Inverse": (C.c_int, [F32P, F32P]),
    "scTickHostMat4PerspectiveRhZo": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, F32P]),
    "scTickHostCameraViewProj": (C.c_int, [F32P, C.c_float, C.c_float, C.c_float, C.c_float, F32P]),
}

_LIB = None


class ScTickError(RuntimeError):
    pass


def load():
    """Load libsc_tick.so and bind every declared symbol.  Raises if the library is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ScTickError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C sc_gameengine_amd/csrc).  There is no CPU fallback for the tick path.")
    lib = C.CDLL(LIB_PATH)
    lax = bool(os.environ.get("SC_TICK_LAX_BIND"))     # tools/ab_step.py only: A/B against a build of an older commit
    for name, (res, args) in SYMBOLS.items():
        if lax and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 calls this; the host hands the bytes to every rank)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    if not load().scTickCommGetUniqueId(buf):
        raise ScTickError("scTickCommGetUniqueId failed: " + (load().scTickGetLastError(None) or b"").decode())
    return bytes(buf)
