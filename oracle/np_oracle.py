"""NumPy restatement of ToStore's exact vector-search arithmetic.

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/vs_oracle.c, straight
from the Dart text, so that the two restatements can be cross-checked bit for
bit before a golden fixture is committed (SURVEY.md section 8c "Independence").
PARITY UNPINNED by the reference's own tests: it has none for vectorSearch.

"ref:" = path under /root/reference/lib/src/.

Every accumulation walks i = 0..d-1 in order with one IEEE binary64 rounding
per multiply and per add (Dart `double`, no fused multiply-add); the loops are
vectorised across ROWS only, which does not change any per-row rounding.
"""
from __future__ import annotations

import math

import numpy as np

L2, IP, COSINE = 0, 1, 2  # ref: model/table_schema.dart:2511-2531 enum order


def to_float32(values, dim: int) -> np.ndarray:
    """A1. ref: core/vector_index_manager.dart:1385-1392,
    core/compute/vector_batch_prepare_compute.dart:79-86."""
    out = np.zeros(dim, dtype=np.float32)
    vals = np.asarray(values, dtype=np.float64)
    n = min(vals.shape[0], dim)
    with np.errstate(over="ignore"):
        out[:n] = vals[:n].astype(np.float32)  # RNE, overflow -> inf
    return out


def normalize_f32(v: np.ndarray) -> np.ndarray:
    """A2. ref: core/vector_index_manager.dart:1395-1408."""
    v = np.asarray(v, dtype=np.float32)
    mag = np.float64(0.0)
    for x in v.astype(np.float64):
        mag = mag + x * x
    mag = np.sqrt(mag)
    if mag == 0:
        return v
    inv = np.float64(1.0) / mag
    return (v.astype(np.float64) * inv).astype(np.float32)


def _seq_sums(query: np.ndarray, rows: np.ndarray, metric: int):
    q = np.asarray(query, dtype=np.float32).astype(np.float64)
    r = np.asarray(rows, dtype=np.float32).astype(np.float64)
    n, d = r.shape
    if metric == L2:
        s = np.zeros(n, dtype=np.float64)
        for i in range(d):
            diff = q[i] - r[:, i]
            s = s + diff * diff
        return (s,)
    if metric == IP:
        s = np.zeros(n, dtype=np.float64)
        for i in range(d):
            s = s + q[i] * r[:, i]
        return (s,)
    dot = np.zeros(n, dtype=np.float64)
    mag_a = np.zeros(n, dtype=np.float64)
    mag_b = np.zeros(n, dtype=np.float64)
    for i in range(d):
        dot = dot + q[i] * r[:, i]
        mag_a = mag_a + q[i] * q[i]
        mag_b = mag_b + r[:, i] * r[:, i]
    return dot, mag_a, mag_b


def all_distances(query, rows, metric: int) -> np.ndarray:
    """A3-A5 over every row.  ref: core/ngh_graph_engine.dart:908-946."""
    rows = np.asarray(rows, dtype=np.float32)
    if rows.shape[0] == 0:
        return np.zeros(0, dtype=np.float64)
    with np.errstate(all="ignore"):
        if metric == L2:
            (s,) = _seq_sums(query, rows, metric)
            return np.sqrt(s)
        if metric == IP:
            (s,) = _seq_sums(query, rows, metric)
            return -s
        dot, mag_a, mag_b = _seq_sums(query, rows, metric)
        denom = np.sqrt(mag_a) * np.sqrt(mag_b)
        sim = np.where(denom > 0, dot / np.where(denom > 0, denom, 1.0), 0.0)
        return 1.0 - sim


def compare_double(a: float, b: float) -> int:
    """Dart double.compareTo [external: Dart SDK]."""
    if a < b:
        return -1
    if a > b:
        return 1
    if a == b:
        if a == 0.0:
            an, bn = math.copysign(1.0, a) < 0, math.copysign(1.0, b) < 0
            if an == bn:
                return 0
            return -1 if an else 1
        return 0
    if math.isnan(a):
        return 0 if math.isnan(b) else 1
    return -1


def _sort_key(dist: np.ndarray) -> np.ndarray:
    """Total-order key matching compare_double: maps f64 bits to uint64."""
    d = np.array(dist, dtype=np.float64, copy=True)
    d[np.isnan(d)] = np.nan  # canonical positive quiet NaN sorts last
    bits = d.view(np.uint64)
    neg = (bits >> np.uint64(63)) == 1
    key = np.where(neg, ~bits, bits | np.uint64(1 << 63))
    # canonical NaN (0x7ff8...) is above +inf after the transform; a NaN with
    # the sign bit set cannot remain because of the canonicalisation above
    return key


def search_exhaustive(rows, query, metric: int, k: int, threshold=None, keep=None):
    """A6 applied to every live row; ties by id ascending.
    ref: core/ngh_graph_engine.dart:122-134."""
    rows = np.asarray(rows, dtype=np.float32)
    n = rows.shape[0]
    if n == 0 or k <= 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float64)
    dist = all_distances(query, rows, metric)
    ids = np.arange(n, dtype=np.int64)
    live = np.ones(n, dtype=bool)
    if keep is not None:
        kb = np.unpackbits(np.asarray(keep, dtype=np.uint8), bitorder="little")[:n]
        live &= kb.astype(bool)
    if threshold is not None and not math.isnan(threshold):
        with np.errstate(invalid="ignore"):
            live &= ~(dist > threshold)
    ids, dist = ids[live], dist[live]
    order = np.lexsort((ids, _sort_key(dist)))
    order = order[:k]
    return ids[order], dist[order]


def distance_to_score(distance: float, metric: int) -> float:
    """A8. ref: core/vector_index_manager.dart:1411-1423."""
    if metric == L2:
        return 1.0 / (1.0 + distance)
    if metric == IP:
        try:
            return 1.0 / (1.0 + math.exp(-(-distance)))
        except OverflowError:
            return 0.0
    s = 1.0 - distance
    if compare_double(s, 0.0) < 0:
        return 0.0
    if compare_double(s, 1.0) > 0:
        return 1.0
    return s


def pq_encode(codebook, subspaces: int, centroids: int, sub_dim: int, vectors) -> np.ndarray:
    """N4. ref: core/compute_tasks.dart:2292-2326 (f64 accumulation over the sub-space, first minimum)."""
    cb = np.asarray(codebook, np.float32).reshape(subspaces, centroids, sub_dim).astype(np.float64)
    v = np.asarray(vectors, np.float32).astype(np.float64)
    codes = np.zeros((v.shape[0], subspaces), np.uint8)
    with np.errstate(all="ignore"):
        for m in range(subspaces):
            sub = v[:, m * sub_dim:(m + 1) * sub_dim]
            dist = np.zeros((v.shape[0], centroids))
            for dd in range(sub_dim):  # sequential accumulation, one rounding per multiply and add
                diff = sub[:, dd:dd + 1] - cb[m, :, dd][None, :]
                dist = dist + diff * diff
            best = np.full(v.shape[0], np.inf)
            idx = np.zeros(v.shape[0], np.int64)
            for c in range(centroids):  # strict `<`: ties and NaN keep the lower index
                lt = dist[:, c] < best
                best = np.where(lt, dist[:, c], best)
                idx = np.where(lt, c, idx)
            codes[:, m] = idx.astype(np.uint8)
    return codes


def pq_train_subspace(data, k: int, iterations: int, init_index) -> np.ndarray:
    """N4. ref: core/compute_tasks.dart:2135-2266, arithmetic widths as in the Dart text
    (Float32List stores round to f32; `double` temporaries are f64)."""
    data = np.asarray(data, np.float32)
    n, sd = data.shape
    f32, f64 = np.float32, np.float64
    cent = data[np.asarray(init_index, np.int64)].copy()
    simd = sd % 4 == 0
    d64 = data.astype(f64)
    with np.errstate(all="ignore"):
        for _ in range(iterations):
            norm = np.zeros(k, f64)
            for d in range(sd):
                v = cent[:, d].astype(f64)
                norm = norm + v * v
            norms = (norm * 0.5).astype(f32)
            dot = np.zeros((n, k), f64)
            if simd:
                for d in range(0, sd, 4):
                    r = [(data[:, d + u:d + u + 1] * cent[:, d + u][None, :]).astype(f32).astype(f64) for u in range(4)]
                    dot = dot + (((r[0] + r[1]) + r[2]) + r[3])
            else:
                c64 = cent.astype(f64)
                for d in range(sd):
                    dot = dot + d64[:, d:d + 1] * c64[:, d][None, :]
            score = dot - norms.astype(f64)[None, :]
            best = np.full(n, -np.inf)
            assign = np.zeros(n, np.int64)
            for c in range(k):
                gt = score[:, c] > best
                best = np.where(gt, score[:, c], best)
                assign = np.where(gt, c, assign)
            sums = np.zeros((k, sd), f32)
            counts = np.zeros(k, np.int64)
            for i in range(n):  # Float32List += : one f32 rounding per add, in sample order
                c = assign[i]
                counts[c] += 1
                sums[c] = (sums[c].astype(f64) + d64[i]).astype(f32)
            changed = False
            for c in range(k):
                if counts[c] == 0:
                    continue
                new = sums[c].astype(f64) * (1.0 / counts[c])
                if (np.abs(cent[c].astype(f64) - new) > 1e-4).any():
                    changed = True
                cent[c] = new.astype(f32)
            if not changed:
                break
    return cent


# ---- A7 page framing ------------------------------------------------------
def pq_train_subspace_pp(data, k: int, iterations: int, first: int, draws) -> np.ndarray:
    """N3. ref: core/vector_quantizer.dart:81-350 (VectorQuantizer.train, one sub-space): k-means++ seeding, then
    Lloyd iterations on squared distances.  `first` / `draws` stand for Dart's Random(42) (nextInt, nextDouble)."""
    data = np.asarray(data, np.float32)
    n, sd = data.shape
    f32, f64 = np.float32, np.float64
    simd = sd % 4 == 0
    d64 = data.astype(f64)

    def dists(cvec):  # squared distance of every sample to one centre, widths as in the Dart text
        acc = np.zeros(n, f64)
        if simd:
            for d in range(0, sd, 4):
                m = []
                for u in range(4):
                    diff = (data[:, d + u] - f32(cvec[d + u])).astype(f32)
                    m.append((diff * diff).astype(f32).astype(f64))
                acc = acc + (((m[0] + m[1]) + m[2]) + m[3])
        else:
            c64 = cvec.astype(f64)
            for d in range(sd):
                diff = d64[:, d] - c64[d]
                acc = acc + diff * diff
        return acc

    cent = np.zeros((k, sd), f32)
    cent[0] = data[first]
    with np.errstate(all="ignore"):
        min_d = np.full(n, np.inf)
        for c in range(1, k):
            min_d = np.minimum(min_d, dists(cent[c - 1]))
            total = 0.0
            for i in range(n):
                total += float(min_d[i])
            selected = n - 1
            if total > 0:
                thr = float(draws[c - 1]) * total
                for i in range(n):
                    thr -= float(min_d[i])
                    if thr <= 0:
                        selected = i
                        break
            cent[c] = data[selected]
        for _ in range(iterations):
            dm = np.stack([dists(cent[c]) for c in range(k)], axis=1)
            best = np.full(n, np.inf)
            assign = np.zeros(n, np.int64)
            for c in range(k):
                lt = dm[:, c] < best
                best = np.where(lt, dm[:, c], best)
                assign = np.where(lt, c, assign)
            sums = np.zeros((k, sd), f32)
            counts = np.zeros(k, np.int64)
            for i in range(n):
                c = assign[i]
                counts[c] += 1
                sums[c] = (sums[c].astype(f64) + d64[i]).astype(f32)
            changed = False
            for c in range(k):
                if counts[c] == 0:
                    continue
                new = sums[c].astype(f64) * (1.0 / counts[c])
                if (np.abs(cent[c].astype(f64) - new) > 1e-6).any():
                    changed = True
                cent[c] = new.astype(f32)
            if not changed:
                break
    return cent


def crc32(data: bytes) -> int:
    """ref: core/btree_page.dart:61-89 (IEEE CRC-32; same as zlib.crc32)."""
    import zlib

    return zlib.crc32(data) & 0xFFFFFFFF


def vectors_per_raw_page(page_size: int, dims: int, bpe: int) -> int:
    """ref: core/ngh_page.dart:575-579."""
    usable = page_size - 20 - 8 - 64
    vec = dims * bpe
    return usable // vec if usable > 0 and vec > 0 else 0


def rawvec_page_build(vectors: np.ndarray, precision: int, page_size: int) -> bytes:
    """ref: core/ngh_page.dart:397-429 + core/btree_page.dart:148-213."""
    import struct

    v = np.asarray(vectors, dtype=np.float32)
    count, dims = v.shape
    if precision == 1:
        data = v.astype("<f4").tobytes()
    elif precision == 0:
        data = v.astype("<f8").tobytes()
    else:
        c = v.astype(np.float64)
        c = np.where(np.isnan(c), 1.0, np.clip(c, -1.0, 1.0))
        x = c * 127.0
        r = np.where(x < 0, -np.floor(-x + 0.5), np.floor(x + 0.5))
        data = r.astype(np.int8).tobytes()
    payload = struct.pack("<HHB3x", count, dims, precision) + data
    if 20 + len(payload) > page_size:
        raise ValueError("page overflow")
    hdr = struct.pack("<IHBBIII", 0x32475054, 20, 8, 0, len(payload), crc32(payload), 0)
    return (hdr + payload).ljust(page_size, b"\0")
