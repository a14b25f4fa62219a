#!/usr/bin/env python3
"""Generate the .scsector fixtures under tests/golden/sectors/ from the REAL reference code.

TEST INFRASTRUCTURE ONLY.  Run where /root/reference exists:

    make -C oracle ref && python oracle/make_golden_sectors.py

Every base file is written by the reference's own WriteSectorFile (tools/shared/world_format.cpp,
compiled unmodified into oracle/_ref/libsc_ref.so); the variants (cut short, padded records, foreign
and empty chunks) are byte edits of those files made here.  For every file the expected content is
what the reference's own ReadSectorFile returns for it.  The fixtures are DATA: the files, and in
sc_sector_ref.json the reader's output (floats as uint32 bit patterns, names as hex).
"""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
SECT = os.path.join(OUT, "sectors")
CAP = 64


def ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def load_ref():
    lib_path = os.path.join(HERE, "_ref", "libsc_ref.so")
    if not os.path.exists(lib_path):
        sys.exit("oracle/_ref/libsc_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    ref = C.CDLL(lib_path)
    ref.ref_hash_asset_path.restype = C.c_uint64
    ref.ref_hash_asset_path.argtypes = [C.c_char_p]
    ref.ref_sector_path.restype = C.c_uint32
    ref.ref_sector_path.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_uint32]
    return ref


def ref_read(ref, path):
    ver = C.c_uint32()
    sxz = np.zeros(2, np.int32)
    counts = np.zeros(4, np.uint32)
    pts = C.c_uint32()
    u64 = [np.zeros(CAP, np.uint64) for _ in range(5)]
    u32 = [np.zeros(CAP, np.uint32) for _ in range(2)]
    trs = np.zeros((CAP, 9), np.float32)
    names = np.zeros((CAP, 64), np.uint8)
    ok = ref.ref_sector_read(path.encode(), C.byref(ver), ptr(sxz, C.c_int32), ptr(counts, C.c_uint32), C.byref(pts), CAP,
                             *[ptr(a, C.c_uint64) for a in u64], *[ptr(a, C.c_uint32) for a in u32],
                             ptr(trs, C.c_float), ptr(names, C.c_char))
    if not ok:
        return {"ok": 0}
    n = int(min(counts[0], CAP))
    return {
        "ok": 1, "version": int(ver.value), "sector": sxz.tolist(), "counts": counts.tolist(), "lane_points": int(pts.value),
        "id": u64[0][:n].tolist(), "model_id": u64[1][:n].tolist(), "mesh_id": u64[2][:n].tolist(),
        "material_id": u64[3][:n].tolist(), "albedo_texture_id": u64[4][:n].tolist(),
        "material_flags": u32[0][:n].tolist(), "tags": u32[1][:n].tolist(),
        "trs_bits": trs[:n].view(np.uint32).tolist(), "name_hex": [bytes(names[i]).hex() for i in range(n)],
    }


def main():
    ref = load_ref()
    os.makedirs(SECT, exist_ok=True)
    rng = np.random.default_rng(424242)
    cases = {}

    def write(name, version, n, sx, sz, lanes=0, ppl=0, spawners=0, colliders=0):
        ids = rng.integers(1, 2**62, n, dtype=np.uint64)
        model, mesh, mat, alb = (rng.integers(0, 2**63, n, dtype=np.uint64) for _ in range(4))
        flags = rng.integers(0, 2, n).astype(np.uint32)
        tags = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        trs = np.concatenate([rng.uniform(-4096, 4096, (n, 3)), rng.uniform(-3.2, 3.2, (n, 3)), rng.uniform(0.1, 64, (n, 3))], axis=1).astype(np.float32)
        if n > 2:
            trs[1, 6:9] = 0.0          # the all-zero scale TransformSystem repairs
            trs[2, 3:6] = 0.0
        names = np.zeros((n, 64), np.uint8)
        for i in range(n):
            if i % 3 != 2:             # every third instance is unnamed ("Inst_<id>" fallback downstream)
                s = f"prop_{name}_{i}".encode()
                names[i, :len(s)] = np.frombuffer(s, np.uint8)
        if n > 3:
            names[3, :] = ord("x")     # 64 bytes without a terminator: the reader must cut it at 63
        path = os.path.join(SECT, name + ".scsector")
        ok = ref.ref_sector_write(path.encode(), version, sx, sz, n, ptr(ids, C.c_uint64), ptr(model, C.c_uint64), ptr(mesh, C.c_uint64),
                                  ptr(mat, C.c_uint64), ptr(alb, C.c_uint64), ptr(flags, C.c_uint32), ptr(tags, C.c_uint32),
                                  ptr(trs, C.c_float), ptr(names, C.c_char), lanes, ppl, spawners, colliders)
        assert ok, name
        return path

    def edit(name, src, fn):
        data = bytearray(open(src, "rb").read())
        data = fn(data)
        path = os.path.join(SECT, name + ".scsector")
        open(path, "wb").write(bytes(data))
        return path

    paths = {}
    paths["v4_full"] = write("v4_full", 4, 7, -3, 12, lanes=2, ppl=3, spawners=2, colliders=3)
    paths["v3_overrides"] = write("v3_overrides", 3, 5, 0, 0, lanes=1, ppl=2)
    paths["v2_names"] = write("v2_names", 2, 4, 100, -100, colliders=1)
    paths["v1_bare"] = write("v1_bare", 1, 6, 1, 2, spawners=1)
    paths["v4_empty"] = write("v4_empty", 4, 0, 5, 5)
    paths["v4_only_lanes"] = write("v4_only_lanes", 4, 0, 7, -7, lanes=3, ppl=1)
    paths["v4_sixteen"] = write("v4_sixteen", 4, 16, 2, 3)

    base = paths["v4_full"]
    inst_rec = 8 + 8 + 8 + 8 + 36 + 64 + 4 + 12                       # v4 record with name and overrides
    hdr = 16                                                           # magic, version, sector
    paths["cut_mid_record"] = edit("cut_mid_record", base, lambda d: d[:hdr + 8 + 4 + 2 * inst_rec + 30])
    paths["cut_in_count"] = edit("cut_in_count", base, lambda d: d[:hdr + 8 + 2])
    paths["cut_in_chunk_header"] = edit("cut_in_chunk_header", base, lambda d: d[:hdr + 5])
    paths["cut_after_header"] = edit("cut_after_header", base, lambda d: d[:hdr])
    paths["cut_in_file_header"] = edit("cut_in_file_header", base, lambda d: d[:9])
    paths["bad_magic"] = edit("bad_magic", base, lambda d: bytearray(b"SECX") + d[4:])
    paths["three_bytes"] = edit("three_bytes", base, lambda d: d[:3])
    paths["foreign_chunk_first"] = edit("foreign_chunk_first", base,
                                        lambda d: d[:hdr] + bytearray(struct.pack("<4sI", b"XTRA", 10)) + bytearray(range(10)) + d[hdr:])
    paths["empty_chunk_header"] = edit("empty_chunk_header", base,
                                       lambda d: d[:hdr] + bytearray(struct.pack("<4sI", b"INST", 0)) + d[hdr:])

    def pad_records(d):
        # grow every INST record by 5 bytes the reader has to skip (record size comes from the chunk size)
        size, count = struct.unpack_from("<II", d, hdr + 4)
        assert (size - 4) // count == inst_rec
        out = bytearray(d[:hdr]) + bytearray(struct.pack("<4sII", b"INST", 4 + count * (inst_rec + 5), count))
        at = hdr + 12
        for _ in range(count):
            out += d[at:at + inst_rec] + bytearray(b"\xEE" * 5)
            at += inst_rec
        return out + d[at:]
    paths["padded_records"] = edit("padded_records", base, pad_records)

    def version_lies(d):
        # a v4 header over v3-sized records: the reader believes the header for model_id and finds no room for overrides
        src = bytearray(open(paths["v3_overrides"], "rb").read())
        struct.pack_into("<I", src, 4, 4)
        return src
    paths["v4_header_v3_records"] = edit("v4_header_v3_records", base, version_lies)

    def second_inst_chunk(d):
        other = bytearray(open(paths["v4_sixteen"], "rb").read())
        return d + other[hdr:]           # a second, longer INST chunk after everything else: it replaces the first
    paths["two_inst_chunks"] = edit("two_inst_chunks", base, second_inst_chunk)

    for name, path in sorted(paths.items()):
        cases[name] = ref_read(ref, path)
        cases[name]["bytes"] = os.path.getsize(path)

    hashes = {}
    for p in ["meshes/cube.obj", "Meshes\\Cube.OBJ", "a/./b/../c.TXT", "", "textures/checker.ppm", "x//y///z", "./rel/path/", "../up/One"]:
        hashes[p] = int(ref.ref_hash_asset_path(p.encode()))
    sector_paths = {}
    buf = C.create_string_buffer(512)
    for root, x, z in [("world", 0, 0), ("/abs/root/", -12, 7), ("", 3, -4), (None, 1, 1)]:
        ref.ref_sector_path(root.encode() if root is not None else None, x, z, buf, 512)
        sector_paths[json.dumps([root, x, z])] = buf.value.decode()

    with open(os.path.join(OUT, "sc_sector_ref.json"), "w") as f:
        json.dump({"generated_by": "oracle/make_golden_sectors.py over oracle/_ref/libsc_ref.so (reference tools/shared/world_format.cpp)",
                   "cases": cases, "hash_asset_path": hashes, "sector_path": sector_paths}, f, indent=1)
    print("wrote", len(cases), "sector fixtures to", os.path.normpath(SECT))


if __name__ == "__main__":
    main()
