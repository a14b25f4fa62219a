"""Procedural lane graph of a rectangle of sectors, as flat arrays for scTickSetLaneGraph.

Host-side restatement (numpy, float32 / int32 arithmetic) of TrafficLaneGraph::buildProceduralForSector
(src/engine/traffic/sc_traffic_lanes.cpp:158-225) called for every sector in the order given -- two lanes per axis, half a
lane width (3.5 m / 2) off the sector's centre lines -- with addNode's de-duplication by quantised position and direction
(:33-43, :65-91: neighbouring sectors share the nodes on their common edge, which is what links their lanes) and
addSegment's direction / length (:93-135).  Node and segment ids come out as the reference's: in order of first insertion.
"""
from dataclasses import dataclass

import numpy as np

LANE_WIDTH = np.float32(3.5)          # sc_traffic_lanes.h:92
SPEED_LIMIT = np.float32(12.0)        # sc_traffic_lanes.h:93
INVALID_LANE = 0xFFFFFFFF
TIER_PHYSICS, TIER_KINEMATIC, TIER_ON_RAILS = 0, 1, 2       # TrafficSimMode, sc_traffic_common.h:11-16


@dataclass
class LaneGraph:
    seg_start: np.ndarray       # (S, 3) float32
    seg_dir: np.ndarray         # (S, 3) float32
    seg_length: np.ndarray      # (S,) float32
    seg_active: np.ndarray      # (S,) uint8
    seg_start_node: np.ndarray  # (S,) uint32
    seg_end_node: np.ndarray    # (S,) uint32
    seg_speed_limit: np.ndarray # (S,) float32
    node_pos: np.ndarray        # (N, 3) float32
    node_conn_offset: np.ndarray  # (N + 1,) uint32
    node_conn: np.ndarray       # (C,) uint32
    sector_segments: np.ndarray  # (sectors, 4) uint32: +x, -x, +z, -z lane of each sector, in build order

    @property
    def segments(self):
        return len(self.seg_length)

    @property
    def nodes(self):
        return len(self.node_pos)


def _quant(v, scale):
    s = (v.astype(np.float32) * np.float32(scale)).astype(np.float32)
    return np.floor(s + np.where(s >= 0, np.float32(0.5), np.float32(-0.5)).astype(np.float32)).astype(np.int64)


def build_procedural(cx, cz, sector_size=64.0, lane_width=LANE_WIDTH, speed_limit=SPEED_LIMIT):
    """cx, cz: sector coordinates in build order (e.g. SynthWorld's row-major order inside a tile)."""
    cx, cz = np.asarray(cx, np.int32), np.asarray(cz, np.int32)
    S = len(cx)
    size = np.float32(sector_size)
    min_x, min_z = cx.astype(np.float32) * size, cz.astype(np.float32) * size
    max_x, max_z = (min_x + size).astype(np.float32), (min_z + size).astype(np.float32)
    ctr_x, ctr_z = ((min_x + max_x) * np.float32(0.5)).astype(np.float32), ((min_z + max_z) * np.float32(0.5)).astype(np.float32)
    off = np.float32(lane_width) * np.float32(0.5)
    zero = np.zeros(S, np.float32)
    # the eight addNode calls of a sector, in call order: (pos, dir) of start and end of the +x, -x, +z, -z lanes
    calls = [
        ((min_x, zero, ctr_z - off), (1, 0, 0)), ((max_x, zero, ctr_z - off), (1, 0, 0)),
        ((max_x, zero, ctr_z + off), (-1, 0, 0)), ((min_x, zero, ctr_z + off), (-1, 0, 0)),
        ((ctr_x + off, zero, min_z), (0, 0, 1)), ((ctr_x + off, zero, max_z), (0, 0, 1)),
        ((ctr_x - off, zero, max_z), (0, 0, -1)), ((ctr_x - off, zero, min_z), (0, 0, -1)),
    ]
    pos = np.stack([np.stack([np.asarray(c, np.float32) for c in p], axis=1) for p, _ in calls], axis=1)    # (S, 8, 3)
    dirs = np.broadcast_to(np.asarray([d for _, d in calls], np.float32)[None], (S, 8, 3))
    flat_pos, flat_dir = pos.reshape(-1, 3), dirs.reshape(-1, 3)
    key = np.concatenate([_quant(flat_pos, 100.0), _quant(flat_dir, 1000.0)], axis=1)                       # LaneNodeKey
    _, first, inverse = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inverse = inverse.reshape(-1)
    order = np.argsort(first, kind="stable")                 # unique keys by first appearance = node id order
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    node_of_call = rank[inverse].astype(np.uint32).reshape(S, 8)
    node_pos = flat_pos[first[order]].astype(np.float32)      # the FIRST inserted node's position is the one kept
    start_node = node_of_call[:, 0::2].reshape(-1)            # segments in call order: 4 per sector
    end_node = node_of_call[:, 1::2].reshape(-1)
    a, b = node_pos[start_node], node_pos[end_node]
    d = (b - a).astype(np.float32)
    length = np.sqrt(((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
                     + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
    inv = (np.float32(1.0) / length).astype(np.float32)
    seg_dir = (d * inv[:, None]).astype(np.float32)
    nseg, nnode = len(start_node), len(node_pos)
    # LaneNode::connections: the segments that start at the node, in insertion order
    conn_order = np.argsort(start_node, kind="stable").astype(np.uint32)
    counts = np.bincount(start_node, minlength=nnode)
    offset = np.zeros(nnode + 1, np.uint32)
    offset[1:] = np.cumsum(counts)
    return LaneGraph(
        seg_start=a.astype(np.float32), seg_dir=seg_dir, seg_length=length, seg_active=np.ones(nseg, np.uint8),
        seg_start_node=start_node.astype(np.uint32), seg_end_node=end_node.astype(np.uint32),
        seg_speed_limit=np.full(nseg, speed_limit, np.float32), node_pos=node_pos,
        node_conn_offset=offset, node_conn=conn_order, sector_segments=np.arange(nseg, dtype=np.uint32).reshape(S, 4))
