"""Writer/reader restatement of an on-disk NGH index directory.

TEST INFRASTRUCTURE ONLY (see oracle/vs_oracle.h).  PARITY UNPINNED: the
reference cannot run here, so no directory written by it exists; this module
restates the WRITER from the Dart text so that the product's loader
(tsh_index_open_ngh) can be tested against bytes laid out the way the reference
lays them out.  "ref:" = path under /root/reference/lib/src/.

  <ngh>/meta.json                       jsonEncode(NghIndexMeta.toJson())   ref: core/vector_index_manager.dart:623-634,
                                                                                 model/ngh_index_meta.dart:410-447
  <ngh>/rawvec/dir_{p//E}/p{p}.ngh      page 0 NghPartitionMetaPage, pages 1.. NghRawVectorPage
  <ngh>/graph/dir_{p//E}/p{p}.ngh       page 0 NghPartitionMetaPage, pages 1.. NghGraphPage
                                        ref: core/path_manager.dart:293-324 (E = maxEntriesPerDir)
Addressing ref: model/ngh_index_meta.dart:451-490; page frame ref: core/btree_page.dart:132-234.
"""
from __future__ import annotations

import json
import os
import struct
import zlib

import numpy as np

from . import np_oracle as npo

PAGE_HEADER = 20
SAFETY = 64  # encodingSafetyMargin, ref: core/ngh_page.dart:548-552
TYPE_NGH_META, TYPE_NGH_GRAPH, TYPE_NGH_RAWVEC = 5, 6, 8  # ref: core/btree_page.dart:14-55 enum order
METRIC_NAMES = ["l2", "innerProduct", "cosine"]  # ref: model/table_schema.dart:2511-2531
PRECISION_NAMES = ["float64", "float32", "int8"]  # ref: model/table_schema.dart:2481-2500


def frame(page_type: int, payload: bytes, page_size: int) -> bytes:
    """ref: core/btree_page.dart:188-213 (BTreePageIO.buildPageBytes)."""
    if PAGE_HEADER + len(payload) > page_size:
        raise ValueError("page overflow")
    hdr = struct.pack("<IHBBIII", 0x32475054, PAGE_HEADER, page_type, 0, len(payload),
                      zlib.crc32(payload) & 0xFFFFFFFF, 0)
    return (hdr + payload).ljust(page_size, b"\0")


def nodes_per_graph_page(page_size: int, max_degree: int) -> int:
    """ref: core/ngh_page.dart:556-565."""
    usable = page_size - PAGE_HEADER - 4 - SAFETY
    return usable // (2 + max_degree * 4) if usable > 0 else 0


def graph_page_payload(flags: np.ndarray, degrees: np.ndarray, neighbors: np.ndarray, max_degree: int) -> bytes:
    """ref: core/ngh_page.dart:171-191 (NghGraphPage.encodePayload): [slotCount u16][maxDegree u16] then per slot
    [flags u8][actualDegree u8][maxDegree x u32, entries past actualDegree written as 0]."""
    n = len(flags)
    out = bytearray(struct.pack("<HH", n, max_degree))
    for i in range(n):
        nb = np.zeros(max_degree, "<u4")
        d = int(degrees[i])
        nb[:d] = neighbors[i][:d]
        out += bytes([int(flags[i]) & 0xFF, d & 0xFF]) + nb.tobytes()
    return bytes(out)


def partition_meta_page(partition_no: int, category: int, total_entries: int, file_size: int, page_size: int) -> bytes:
    from . import ngh_meta_page_build  # C restatement, ref: core/ngh_page.dart:29-98

    return ngh_meta_page_build(partition_no, category, total_entries, file_size, page_size)


def write_ngh_dir(root: str, vectors: np.ndarray, *, metric: int, precision: int = 1, page_size: int = 16384,
                  max_partition_file_size: int = 16 * 1024 * 1024, max_degree: int = 64, max_entries_per_dir: int = 500,
                  deleted=(), seed: int = 0, skip_rawvec_partitions=(), skip_graph_partitions=(),
                  name: str = "idx_embedding", table: str = "docs", field: str = "embedding") -> dict:
    """Writes the directory; returns the meta dict.  Graph neighbours are random (they are not read by the
    exhaustive path); `deleted` node ids get NghNodeFlags.deleted (ref: core/ngh_page.dart:105-108)."""
    v = np.asarray(vectors, np.float32)
    n, dims = v.shape
    bpe = {0: 8, 1: 4, 2: 1}[precision]
    vpp = npo.vectors_per_raw_page(page_size, dims, bpe)
    npg = nodes_per_graph_page(page_size, max_degree)
    ppp = max_partition_file_size // page_size
    rng = np.random.default_rng(seed)
    dead = np.zeros(n, bool)
    dead[list(deleted)] = True

    def part_path(cat, p):
        d = os.path.join(root, cat, f"dir_{p // max_entries_per_dir}")
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, f"p{p}.ngh")

    n_raw_pages = (n + vpp - 1) // vpp
    raw_parts = (n_raw_pages + ppp - 1) // ppp if n else 1
    for p in range(raw_parts):
        if p in skip_rawvec_partitions:
            continue
        pages = min(ppp, n_raw_pages - p * ppp)
        with open(part_path("rawvec", p), "wb") as f:
            lo = p * ppp * vpp
            f.write(partition_meta_page(p, 2, min(n - lo, pages * vpp), (pages + 1) * page_size, page_size))
            for lp in range(pages):
                chunk = np.zeros((vpp, dims), np.float32)  # pages are always full-capacity, ref: ngh_graph_engine.dart:691-720
                part = v[lo + lp * vpp: lo + (lp + 1) * vpp]
                chunk[:len(part)] = part
                f.write(npo.rawvec_page_build(chunk, precision, page_size))
    n_graph_pages = (n + npg - 1) // npg
    graph_parts = (n_graph_pages + ppp - 1) // ppp if n else 1
    for p in range(graph_parts):
        if p in skip_graph_partitions:
            continue
        pages = min(ppp, n_graph_pages - p * ppp)
        with open(part_path("graph", p), "wb") as f:
            lo = p * ppp * npg
            f.write(partition_meta_page(p, 0, min(n - lo, pages * npg), (pages + 1) * page_size, page_size))
            for lp in range(pages):
                ids = np.arange(lo + lp * npg, lo + (lp + 1) * npg)
                live = ids < n
                flags = np.where(live & dead[np.minimum(ids, n - 1)], 1, 0)
                flags = flags | np.where(live & (rng.random(npg) < 0.05), 2, 0)  # NghNodeFlags.updated: must be ignored
                deg = np.where(live, rng.integers(0, max_degree + 1, npg), 0)
                nb = rng.integers(0, max(n, 1), (npg, max_degree))
                f.write(frame(TYPE_NGH_GRAPH, graph_page_payload(flags, deg, nb, max_degree), page_size))
    meta = {  # key order of NghIndexMeta.toJson, ref: model/ngh_index_meta.dart:410-447
        "version": 1, "name": name, "tableName": table, "fieldName": field, "dimensions": dims,
        "distanceMetric": METRIC_NAMES[metric], "precision": PRECISION_NAMES[precision],
        "timestamps": {"created": "2026-06-12T00:00:00.000", "modified": "2026-06-12T00:00:00.000"},
        "maxDegree": max_degree, "efSearch": 64, "constructionEf": 128, "pruneAlpha": 1.2,
        "pqSubspaces": min(max(dims // 8, 8), 128), "pqCentroids": 256, "pqTrained": n >= 100,
        "totalVectors": int(n - dead.sum()), "deletedCount": int(dead.sum()), "medoidNodeId": 0 if n else -1,
        "nextNodeId": n, "nghPageSize": page_size, "graphPartitionCount": graph_parts,
        "graphNextPageNo": 1 + (n_graph_pages - (graph_parts - 1) * ppp if n else 0),
        "pqCodePartitionCount": 1, "pqCodeNextPageNo": 1, "rawVectorPartitionCount": raw_parts,
        "rawVectorNextPageNo": 1 + (n_raw_pages - (raw_parts - 1) * ppp if n else 0),
        "totalSizeInBytes": 0,
        "nodeIdToPkMeta": {"name": name + "__nid2pk", "nested": {"a": [1, 2, {"b": "}\\\"]"}]}},
        "graphFreeListHeads": {}, "pqCodeFreeListHeads": {"0": -1}, "rawVectorFreeListHeads": {},
        "maxPartitionFileSize": max_partition_file_size,
    }
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "meta.json"), "w") as f:
        f.write(json.dumps(meta, separators=(",", ":")))  # dart:convert jsonEncode emits no whitespace
    return meta


def read_ngh_dir(root: str, max_entries_per_dir: int = 500):
    """Reader restatement (what the reference's cold read path yields per node id): returns
    (meta, vectors float32 [nextNodeId x dims], deleted bool [nextNodeId]).
    ref: core/ngh_partition_manager.dart:131-178,239-287; model/ngh_index_meta.dart:451-490."""
    meta = json.load(open(os.path.join(root, "meta.json")))
    dims, n = int(meta["dimensions"]), int(meta.get("nextNodeId", 0))
    page_size = int(meta.get("nghPageSize", 16384))
    precision = {"float64": 0, "int8": 2}.get(meta.get("precision"), 1)
    bpe = {0: 8, 1: 4, 2: 1}[precision]
    R = int(meta.get("maxDegree", 64))
    vpp, npg = npo.vectors_per_raw_page(page_size, dims, bpe), nodes_per_graph_page(page_size, R)
    ppp = int(meta.get("maxPartitionFileSize", 16 << 20)) // page_size
    vec = np.zeros((n, dims), np.float32)
    dead = np.zeros(n, bool)

    def page(cat, part, page_no):
        path = os.path.join(root, cat, f"dir_{part // max_entries_per_dir}", f"p{part}.ngh")
        if not os.path.exists(path):
            return None
        with open(path, "rb") as f:
            f.seek(page_no * page_size)
            raw = f.read(page_size)
        if not raw:
            return None
        magic, hsz, ptype, _, plen, crc, _ = struct.unpack_from("<IHBBIII", raw)
        if magic != 0x32475054 or hsz != 20 or 20 + plen > len(raw) or zlib.crc32(raw[20:20 + plen]) & 0xFFFFFFFF != crc:
            raise ValueError(f"corrupt page {path}:{page_no}")
        return raw[20:20 + plen]

    from . import rawvec_page_parse

    for lp in range((n + vpp - 1) // vpp):
        part, page_no = lp // ppp, 1 + lp % ppp
        path = os.path.join(root, "rawvec", f"dir_{part // max_entries_per_dir}", f"p{part}.ngh")
        if not os.path.exists(path):
            continue  # empty page: zero vectors
        with open(path, "rb") as f:
            f.seek(page_no * page_size)
            raw = f.read(page_size)
        if not raw:
            continue
        res = rawvec_page_parse(raw.ljust(page_size, b"\0"), dims, vpp)
        if res is None:
            raise ValueError(f"corrupt page {path}:{page_no}")
        rows = res[0]
        lo = lp * vpp
        take = min(vpp, n - lo)
        vec[lo:lo + take] = np.asarray(rows)[:take]
    for lp in range((n + npg - 1) // npg):
        pl = page("graph", lp // ppp, 1 + lp % ppp)
        if pl is None or len(pl) < 4:
            continue
        sc, deg = struct.unpack_from("<HH", pl)
        ss = 2 + 4 * deg
        if deg == 0 or len(pl) < 4 + sc * ss:
            continue
        for s in range(min(sc, npg)):
            nid = lp * npg + s
            if nid < n and pl[4 + s * ss] & 1:
                dead[nid] = True
    return meta, vec, dead
