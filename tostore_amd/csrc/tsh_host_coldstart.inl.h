// tsh_host_coldstart.inl.h -- cold start from ToStore's files: raw-vector partition loader, meta.json, graph tombstones
// Part of the single translation unit tsh_lib.hip (textually included there; not compiled alone).

// ---- rawvec partition file loader -------------
namespace {

// IEEE CRC-32 (core/btree_page.dart:61-89: the reference's byte-at-a-time table loop), computed
// eight bytes per step (slicing-by-8) -- same polynomial, same result
uint32_t crc32_ieee(const uint8_t *p, size_t n) {
  static uint32_t table[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
  });
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  for (size_t i = 0; i < n; ++i) c = table[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
inline uint32_t rd16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t *p) { return rd16(p) | (rd16(p + 2) << 16); }

enum PageKind { PAGE_ERROR = -1, PAGE_ABSENT = 0, PAGE_OK = 1, PAGE_UNDECODABLE = 2 };
constexpr int PAGE_TYPE_NGH_RAW_VECTOR = 8;  // BTreePageType.nghRawVector.index, core/btree_page.dart:14-55

// Decodes one page into out (vpp x dim floats); *vcount = vectors present.
//   PAGE_ABSENT       nothing stored at this offset (file ends before it).  The reference's reader hands out
//                     NghRawVectorPage.empty there -- `capacity` all-zero vectors (ngh_partition_manager.dart:
//                     270-281) -- which it only ever touches if its graph walk reaches such a node.  An
//                     exhaustive scan would offer every one of those zero rows to every query, so here the ids
//                     of an absent page are ABSENT rows (never returned), like ids skipped by an append gap.
//   PAGE_UNDECODABLE  frame and CRC are fine, type is nghRawVector, but the payload is not an NghRawVectorPage
//                     (tryDecodePayload -> null, ngh_page.dart:431-450): what ciphertext looks like when the
//                     index was written with encryptVectorIndex (_decodePayload, ngh_partition_manager.dart:
//                     1091-1102), which this loader cannot undo -> the caller refuses the index.
//   PAGE_ERROR        bad magic / header size / type / CRC (the reference throws: btree_page.dart:215-233), or a
//                     page of another type or another dimension where a raw-vector page of this index belongs.
PageKind decode_rawvec_page(const uint8_t *pg, size_t avail, int page_size, int dim, int vpp, float *out,
                            int *vcount) {
  *vcount = 0;
  if (avail == 0) return PAGE_ABSENT;
  if (avail < 20) return PAGE_ERROR;              // btree_page.dart:162-163 -> StateError
  if (rd32(pg) != 0x32475054u) return PAGE_ERROR; // 'TPG2'
  if (rd16(pg + 4) != 20) return PAGE_ERROR;
  if (pg[6] != PAGE_TYPE_NGH_RAW_VECTOR) return PAGE_ERROR;
  uint32_t plen = rd32(pg + 8), crc = rd32(pg + 12);
  if ((uint64_t)20 + plen > avail) return PAGE_ERROR;  // :221-224
  const uint8_t *pl = pg + 20;
  if (crc32_ieee(pl, plen) != crc) return PAGE_ERROR;  // :226-230
  // NghRawVectorPage.tryDecodePayload, ngh_page.dart:431-450
  if (plen < 8) return PAGE_UNDECODABLE;
  uint32_t vc = rd16(pl), dims = rd16(pl + 2);
  int prec = pl[4];
  if (dims == 0) return PAGE_UNDECODABLE;
  int bpe = prec == 0 ? 8 : (prec == 2 ? 1 : 4);
  if ((uint64_t)plen < 8 + (uint64_t)vc * dims * bpe) return PAGE_UNDECODABLE;
  if ((int)dims != dim) return PAGE_ERROR;  // not this index's column
  int take = (int)std::min<uint32_t>(vc, (uint32_t)vpp);
  const uint8_t *d = pl + 8;
  if (prec == 1) {  // little-endian f32 on a little-endian host: the per-element getFloat32 loop is a copy
    memcpy(out, d, (size_t)take * dim * sizeof(float));
    *vcount = take;
    return PAGE_OK;
  }
  for (int v = 0; v < take; ++v)
    for (int i = 0; i < dim; ++i) {  // getVectorAsFloat32, ngh_page.dart:364-391
      const uint8_t *e = d + ((size_t)v * dim + i) * bpe;
      float f;
      if (prec == 0) {
        uint64_t u = (uint64_t)rd32(e) | ((uint64_t)rd32(e + 4) << 32);
        double dv;
        memcpy(&dv, &u, 8);
        f = (float)dv;
      } else {
        f = (float)((double)(int8_t)*e / 127.0);
      }
      out[(size_t)v * dim + i] = f;
    }
  *vcount = take;
  (void)page_size;
  return PAGE_OK;
}

// The loader as a two-stage pipeline (round 6).  A partition file is taken in batches of ~32 MB of rows:
//   stage A, on the host pool: every worker reads ITS pages itself (pread: the copies out of the page cache run side by
//            side, where one fread of the whole batch was one core's memcpy), checks their CRC and copies / widens their
//            vectors into the batch's row buffer -- pinned host memory, so that
//   stage B, on a helper thread: tsh_index_append's H2D copy of the batch runs at the link's speed and, with the norms'
//            kernel behind it, WHILE stage A is on the next batch (two row buffers; batches are appended in order, one
//            at a time; the pipeline runs on across the files of an index).
// Before: fread | decode | append one after the other, 3.8 GB/s of page bytes (round 1).
struct RawvecPipe {
  struct Slot {
    float *rows = nullptr;
    bool pinned = false;
    size_t cap = 0;  // floats
    std::vector<uint8_t> raw;
    std::vector<int> kinds, counts;
  } slot[2];
  int cur = 0;
  std::thread helper;
  int pending_rc = TSH_OK;
  std::string pending_err;
  int64_t loaded = 0;  // rows appended (written by the helper; read after wait())

  float *rows_of(Slot &sl, size_t floats) {
    if (sl.cap >= floats) return sl.rows;
    release(sl);
    void *p = nullptr;
    if (hipHostMalloc(&p, floats * sizeof(float), hipHostMallocPortable) == hipSuccess) {
      sl.pinned = true;
    } else {
      (void)hipGetLastError();
      p = malloc(floats * sizeof(float));
      sl.pinned = false;
    }
    sl.rows = static_cast<float *>(p);
    sl.cap = p ? floats : 0;
    return sl.rows;
  }
  void release(Slot &sl) {
    if (sl.rows) {
      if (sl.pinned) (void)hipHostFree(sl.rows);
      else free(sl.rows);
    }
    sl.rows = nullptr;
    sl.cap = 0;
  }
  // the append in flight, if any: its status (and its message, which lives in the helper's thread otherwise); reported once
  int wait() {
    if (helper.joinable()) helper.join();
    const int rc = pending_rc;
    if (rc != TSH_OK) g_err = pending_err;
    pending_rc = TSH_OK;
    return rc;
  }
  struct Run {
    int64_t first_id, len;
    size_t src;  // float offset into the slot's rows
  };
  // appends `runs` out of slot `si` on the helper thread (the previous append must have been waited for)
  void append_async(tsh_index *idx, int si, std::vector<Run> runs, int dim) {
    (void)dim;
    pending_rc = TSH_OK;
    auto work = [this, idx, si, runs = std::move(runs)]() {
      for (const Run &r : runs) {
        const int rc = tsh_index_append(idx, r.first_id, r.len, slot[si].rows + r.src);
        if (rc != TSH_OK) {
          pending_rc = rc;
          pending_err = g_err;
          return;
        }
        loaded += r.len;
      }
    };
    try {
      helper = std::thread(work);
    } catch (const std::system_error &) {  // no thread to be had: the append runs here; wait() reports its status all the same
      work();
    }
  }
  ~RawvecPipe() {
    (void)wait();
    release(slot[0]);
    release(slot[1]);
  }
};

// tsh_index_load_rawvec_file with the page census tsh_index_open_ngh reports
// lo_row_id: only node ids >= lo_row_id are loaded (tsh_index_open_ngh_shard: a rank's range may begin inside a
// partition file and inside a page); pages that hold none of them are neither read nor counted
// pipe: the caller's pipeline (tsh_index_open_ngh: one over all files; the last batch's append is then still in flight on
// return, *out_rows is not written, and the caller waits and reads pipe->loaded); NULL: a pipeline of this call's own
int load_rawvec_file(tsh_index *idx, const char *path, int32_t page_size, int32_t precision, int64_t first_row_id,
                     int64_t max_rows, int64_t *out_rows, int64_t *out_absent_pages, int64_t lo_row_id = -1,
                     RawvecPipe *pipe = nullptr);

}  // namespace

extern "C" int32_t tsh_index_load_rawvec_file(tsh_index *idx, const char *path, int32_t page_size,
                                              int32_t precision, int64_t first_row_id, int64_t max_rows,
                                              int64_t *out_rows) {
  return load_rawvec_file(idx, path, page_size, precision, first_row_id, max_rows, out_rows, nullptr);
}

namespace {
int load_rawvec_file(tsh_index *idx, const char *path, int32_t page_size, int32_t precision, int64_t first_row_id,
                     int64_t max_rows, int64_t *out_rows, int64_t *out_absent_pages, int64_t lo_row_id, RawvecPipe *pipe) {
  if (out_rows) *out_rows = 0;
  if (out_absent_pages) *out_absent_pages = 0;
  if (!idx || !path) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  if (page_size < 64 || precision < 0 || precision > 2 || first_row_id < 0 || max_rows < 0)
    return set_err(TSH_E_BAD_ARG, "bad page_size / precision / row range");
  int dim = idx->dim;
  int bpe = precision == 0 ? 8 : (precision == 2 ? 1 : 4);
  int usable = page_size - 20 - 8 - 64;  // ngh_page.dart:575-579
  int vpp = usable > 0 ? usable / (dim * bpe) : 0;
  if (vpp <= 0) return set_err(TSH_E_BAD_ARG, "page_size %d holds no %d-dim vector", page_size, dim);
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return set_err(TSH_E_IO, "cannot open %s", path);
  RawvecPipe own;
  RawvecPipe *pp = pipe ? pipe : &own;
  const int64_t loaded0 = pp->loaded;  // (a caller's pipeline: nothing of it is in flight between two files' batches but the last append)
  int64_t n_pages = (max_rows + vpp - 1) / vpp;  // data pages needed to cover the ids
  const int64_t lo_rel = std::max<int64_t>(0, lo_row_id - first_row_id);  // first id wanted, relative to the file's first
  const int64_t p_first = lo_rel / vpp;                                   // ... and the data page that holds it
  const int BATCH = std::max(1, (int)((32 << 20) / ((int64_t)vpp * dim * 4)));
  int64_t absent = 0;
  int rc = TSH_OK;
  for (int64_t p0 = p_first; p0 < n_pages && rc == TSH_OK; p0 += BATCH) {
    const int64_t nb = std::min<int64_t>(BATCH, n_pages - p0);
    const int si = pp->cur;
    RawvecPipe::Slot &sl = pp->slot[si];  // (free: the append that read it was waited for before the last one was started)
    // (sized for the batches this file has: 16 MB partition files never fill a 32 MB batch, and pinning is not free)
    float *rows = pp->rows_of(sl, (size_t)std::min<int64_t>(BATCH, n_pages - p_first) * vpp * dim);
    if (!rows) {
      rc = set_err(TSH_E_OOM, "no host memory for a batch of rows");
      break;
    }
    if (sl.raw.size() < (size_t)nb * (size_t)page_size) sl.raw.resize((size_t)nb * (size_t)page_size);
    if (sl.kinds.size() < (size_t)nb) {
      sl.kinds.resize((size_t)nb);
      sl.counts.resize((size_t)nb);
    }
    // stage A: pages read and decoded (CRC + copy / widen) in parallel on the host pool
    parallel_for((int32_t)nb, [&](int32_t b) {
      uint8_t *pg = sl.raw.data() + (size_t)b * (size_t)page_size;
      size_t got = 0;
      const off_t at = (off_t)(1 + p0 + b) * page_size;  // pageNo 0 is the partition meta page
      while (got < (size_t)page_size) {
        const ssize_t r = pread(fd, pg + got, (size_t)page_size - got, at + (off_t)got);
        if (r <= 0) break;  // (the end of the file, or an error: what is missing reads as absent / a bad page)
        got += (size_t)r;
      }
      int vc = 0;
      sl.kinds[(size_t)b] = (int)decode_rawvec_page(pg, got, page_size, dim, vpp, rows + (size_t)b * vpp * dim, &vc);
      sl.counts[(size_t)b] = vc;
    });
    // runs of consecutive present rows inside the batch are appended together; the ids of absent pages and of
    // slots past a page's vectorCount stay absent rows
    std::vector<RawvecPipe::Run> runs;
    int64_t run_start = -1, run_len = 0;
    auto flush = [&]() {
      if (run_len > 0) runs.push_back(RawvecPipe::Run{first_row_id + p0 * vpp + run_start, run_len, (size_t)run_start * (size_t)dim});
      run_start = -1;
      run_len = 0;
    };
    int bad_kind = 0;
    int64_t bad_page = 0;
    for (int64_t b = 0; b < nb; ++b) {
      const int kind = sl.kinds[(size_t)b];
      if (kind == (int)PAGE_ERROR || kind == (int)PAGE_UNDECODABLE) {  // the rows in front of it are still appended
        bad_kind = kind;
        bad_page = 1 + p0 + b;
        break;
      }
      if (kind == (int)PAGE_ABSENT) {
        ++absent;
        flush();
        continue;
      }
      int64_t base = (p0 + b) * vpp;  // id offset of this page's slot 0
      int64_t lim = std::min<int64_t>(sl.counts[(size_t)b], max_rows - base);
      const int64_t skip = std::max<int64_t>(0, lo_rel - base);  // slots below the range's first id (its first page only)
      if (lim <= skip) continue;
      if (run_len > 0 && skip == 0 && run_start + run_len == b * vpp) {
        run_len += lim;
      } else {
        flush();
        run_start = b * vpp + skip;
        run_len = lim - skip;
      }
      if (lim < vpp) flush();  // slots past vectorCount are absent rows
    }
    flush();
    // stage B: the previous batch's append must be through (order, and its status); then this batch's starts
    rc = pp->wait();
    if (rc == TSH_OK && !runs.empty()) {
      pp->append_async(idx, si, std::move(runs), dim);
      pp->cur ^= 1;
    }
    if (rc == TSH_OK && bad_kind) {
      rc = pp->wait();
      if (rc == TSH_OK) {
        if (bad_kind == (int)PAGE_ERROR)
          rc = set_err(TSH_E_FORMAT, "%s: page %lld has a bad header / type / CRC / dimension", path, (long long)bad_page);
        else
          rc = set_err(TSH_E_FORMAT, "%s: page %lld passes its CRC but is no raw-vector payload -- an index written "
                       "with encryptVectorIndex cannot be opened without Dart", path, (long long)bad_page);
      }
    }
  }
  close(fd);
  if (!pipe || rc != TSH_OK) {  // this call's own pipeline (or a failure): nothing stays in flight
    const std::string keep = g_err;
    const int wrc = pp->wait();
    if (rc == TSH_OK) rc = wrc;
    else g_err = keep;
    if (out_rows) *out_rows = pp->loaded - loaded0;
  }
  if (out_absent_pages) *out_absent_pages = absent;
  return rc;
}
}  // namespace

// ---- open an on-disk NGH index directory (N1: meta.json + rawvec + graph tombstones) ----
namespace {

// Top-level scalar members of a JSON object (what NghIndexMeta.fromJson reads for the
// fields used here, model/ngh_index_meta.dart:359-408); nested values are skipped.
struct JsonScan {
  const char *p, *end;
  bool ok = true;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
  bool lit(const char *w) {
    size_t n = strlen(w);
    if ((size_t)(end - p) >= n && memcmp(p, w, n) == 0) { p += n; return true; }
    return false;
  }
  std::string str() {
    std::string o;
    if (p >= end || *p != '"') { ok = false; return o; }
    for (++p; p < end && *p != '"'; ++p) {
      if (*p == '\\' && p + 1 < end) {
        ++p;
        switch (*p) {
          case 'n': o += '\n'; break;
          case 't': o += '\t'; break;
          case 'r': o += '\r'; break;
          case 'b': o += '\b'; break;
          case 'f': o += '\f'; break;
          case 'u': o += '?'; p += (end - p > 4) ? 4 : 0; break;  // names only; not needed verbatim
          default: o += *p;
        }
      } else {
        o += *p;
      }
    }
    if (p >= end) { ok = false; return o; }
    ++p;
    return o;
  }
  void skip() {  // any value
    ws();
    if (p >= end) { ok = false; return; }
    if (*p == '"') { str(); return; }
    if (*p == '{' || *p == '[') {
      const char close = *p == '{' ? '}' : ']';
      const bool obj = *p == '{';
      ++p; ws();
      if (p < end && *p == close) { ++p; return; }
      while (ok) {
        if (obj) { ws(); str(); ws(); if (p >= end || *p != ':') { ok = false; return; } ++p; }
        skip(); ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == close) { ++p; return; }
        ok = false;
      }
      return;
    }
    if (lit("true") || lit("false") || lit("null")) return;
    const char *q = p;
    while (p < end && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) ++p;
    if (p == q) ok = false;
  }
};

bool json_top_level(const std::string &text, std::map<std::string, std::string> &out) {
  JsonScan j{text.data(), text.data() + text.size()};
  j.ws();
  if (j.p >= j.end || *j.p != '{') return false;
  ++j.p; j.ws();
  if (j.p < j.end && *j.p == '}') return true;
  while (j.ok) {
    j.ws();
    std::string key = j.str();
    j.ws();
    if (!j.ok || j.p >= j.end || *j.p != ':') return false;
    ++j.p; j.ws();
    if (j.p < j.end && *j.p == '"') {
      out[key] = j.str();
    } else {
      const char *q = j.p;
      j.skip();
      if (j.ok && *q != '{' && *q != '[') out[key] = std::string(q, j.p);
    }
    j.ws();
    if (j.p < j.end && *j.p == ',') { ++j.p; continue; }
    if (j.p < j.end && *j.p == '}') return j.ok;
    return false;
  }
  return false;
}

// (json[k] as num?)?.toInt() ?? dflt -- Dart's toInt truncates a fractional number
int64_t json_int(const std::map<std::string, std::string> &m, const char *k, int64_t dflt, bool *present = nullptr) {
  auto it = m.find(k);
  if (present) *present = false;
  if (it == m.end() || it->second == "null" || it->second.empty()) return dflt;
  char *e = nullptr;
  double v = strtod(it->second.c_str(), &e);
  if (e == it->second.c_str()) return dflt;
  if (present) *present = true;
  if (it->second.find_first_of(".eE") == std::string::npos) return strtoll(it->second.c_str(), nullptr, 10);
  return (int64_t)v;
}

std::string ngh_partition_path(const std::string &dir, const char *category, int64_t partition, int64_t per_dir) {
  // core/path_manager.dart:293-324: <ngh>/<category>/dir_{partition ~/ maxEntriesPerDir}/p{partition}.ngh
  char buf[96];
  snprintf(buf, sizeof buf, "/%s/dir_%lld/p%lld.ngh", category, (long long)(partition / per_dir), (long long)partition);
  return dir + buf;
}

// Validates a page frame (BTreePageIO.parsePageBytes, core/btree_page.dart:215-233).
// Returns payload pointer/len, nullptr + *err=false for "nothing there" and *err=true for a corrupt page.
const uint8_t *page_payload(const uint8_t *pg, size_t avail, uint32_t *plen, int *type, bool *err) {
  *err = false;
  if (avail == 0) return nullptr;
  *err = true;
  if (avail < 20 || rd32(pg) != 0x32475054u || rd16(pg + 4) != 20 || pg[6] >= 10) return nullptr;
  *plen = rd32(pg + 8);
  if ((uint64_t)20 + *plen > avail) return nullptr;
  if (crc32_ieee(pg + 20, *plen) != rd32(pg + 12)) return nullptr;
  *type = pg[6];
  *err = false;
  return pg + 20;
}

}  // namespace

namespace {
// tsh_index_open_ngh (world == 0: the whole index on n_devices) and tsh_index_open_ngh_shard (world >= 1: node ids
// [rank * per, (rank + 1) * per), per = ceil(nextNodeId / world), as a shard handle on `device`)
int32_t open_ngh_impl(const char *ngh_dir, int32_t max_entries_per_dir, int32_t n_devices, int32_t device, int32_t world,
                      int32_t rank, tsh_index **out, tsh_ngh_info *info) {
  if (out) *out = nullptr;
  if (info) memset(info, 0, sizeof *info);
  if (!ngh_dir || !out) return set_err(TSH_E_BAD_ARG, "NULL pointer");
  if (max_entries_per_dir <= 0) max_entries_per_dir = 500;  // handler/common.dart:43
  const std::string dir(ngh_dir);
  std::string text;
  {
    FILE *f = fopen((dir + "/meta.json").c_str(), "rb");
    if (!f) return set_err(TSH_E_IO, "cannot open %s/meta.json", ngh_dir);
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
  }
  std::map<std::string, std::string> m;
  if (!json_top_level(text, m)) return set_err(TSH_E_FORMAT, "%s/meta.json is not a JSON object", ngh_dir);
  bool has_dim = false;
  const int64_t dim = json_int(m, "dimensions", 0, &has_dim);
  if (!has_dim || dim < 1 || dim > 65535) return set_err(TSH_E_FORMAT, "meta.json: dimensions missing or out of range");
  const std::string ms = m.count("distanceMetric") ? m["distanceMetric"] : "";
  const int metric = ms == "l2" ? 0 : (ms == "innerProduct" ? 1 : 2);  // ngh_index_meta.dart:494-503
  const std::string ps = m.count("precision") ? m["precision"] : "";
  const int precision = ps == "float64" ? 0 : (ps == "int8" ? 2 : 1);  // :505-514
  const int64_t next_id = json_int(m, "nextNodeId", 0);
  const int64_t page_size = json_int(m, "nghPageSize", 16384);
  const int64_t max_file = json_int(m, "maxPartitionFileSize", 16 * 1024 * 1024);
  const int64_t max_degree = json_int(m, "maxDegree", 64);
  if (page_size < 64 || page_size > (1 << 26) || next_id < 0 || max_degree < 1 || max_degree > 65535)
    return set_err(TSH_E_FORMAT, "meta.json: nghPageSize / nextNodeId / maxDegree out of range");
  const int bpe = precision == 0 ? 8 : (precision == 2 ? 1 : 4);
  const int64_t usable_raw = page_size - 20 - 8 - 64;  // ngh_page.dart:575-579
  const int64_t vpp = usable_raw > 0 ? usable_raw / (dim * bpe) : 0;
  const int64_t usable_graph = page_size - 20 - 4 - 64;  // ngh_page.dart:556-565
  const int64_t slot_size = 2 + max_degree * 4;
  const int64_t npg = usable_graph > 0 ? usable_graph / slot_size : 0;
  const int64_t ppp = max_file / page_size;  // ngh_index_meta.dart:178
  if (vpp <= 0 || ppp <= 0) return set_err(TSH_E_FORMAT, "meta.json: a %lld-byte page holds no %lld-dim vector", (long long)page_size, (long long)dim);
  if (info) {
    info->dimensions = (int32_t)dim;
    info->metric = metric;
    info->precision = precision;
    info->page_size = (int32_t)page_size;
    info->max_degree = (int32_t)max_degree;
    info->next_node_id = next_id;
    info->total_vectors = json_int(m, "totalVectors", 0);
    info->deleted_count = json_int(m, "deletedCount", 0);
    info->max_partition_file_size = max_file;
  }
  // the node ids this handle is opened for
  int64_t lo = 0, hi = next_id;
  if (world >= 1) {
    const int64_t per = (next_id + world - 1) / world;
    lo = std::min<int64_t>(next_id, (int64_t)rank * per);
    hi = std::min<int64_t>(next_id, lo + per);
  }
  if (info) {
    info->row_base = lo;
    info->row_end = hi;
  }
  tsh_index *idx = nullptr;
  int32_t rc = world >= 1 ? tsh_index_create_shard((int32_t)dim, metric, hi - lo, device, lo, &idx)
                          : tsh_index_create((int32_t)dim, metric, next_id, n_devices, &idx);
  if (rc != TSH_OK) return rc;
  // raw vectors: node id -> (partition, page, slot), ngh_index_meta.dart:480-490.  A range of node ids is a run of
  // partition files and, inside the first and the last, of pages: nothing else is opened or read
  int64_t rows_loaded = 0, files = 0, absent_pages = 0, absent_files = 0;
  const int64_t rows_per_part = ppp * vpp;
  RawvecPipe pipe;  // one pipeline over all files: a file's last append runs while the next file's pages are read
  for (int64_t part = lo / rows_per_part; rc == TSH_OK && hi > lo && part * rows_per_part < hi; ++part) {
    const int64_t first = part * rows_per_part, want = std::min(rows_per_part, hi - first);
    const std::string path = ngh_partition_path(dir, "rawvec", part, max_entries_per_dir);
    if (access(path.c_str(), R_OK) == 0) {
      int64_t absent = 0;
      rc = load_rawvec_file(idx, path.c_str(), (int32_t)page_size, precision, first, want, nullptr, &absent, lo, &pipe);
      absent_pages += absent;
      ++files;
    } else {
      // A missing partition file: its ids are absent rows (see decode_rawvec_page: the reference's reader makes
      // up zero vectors here, which an exhaustive scan must not offer to every query).
      ++absent_files;
      absent_pages += (want + vpp - 1) / vpp - std::max<int64_t>(0, lo - first) / vpp;  // (the range's pages of it)
    }
  }
  if (rc == TSH_OK) rc = pipe.wait();
  else {
    const std::string keep = g_err;
    (void)pipe.wait();
    g_err = keep;
  }
  rows_loaded = pipe.loaded;
  // tombstones: flags byte of each graph slot (ngh_page.dart:105-108,198-213)
  int64_t tombstones = 0;
  if (rc == TSH_OK && npg > 0) {
    const int64_t ids_per_part = ppp * npg;
    const int64_t BLOCK = std::max<int64_t>(64, (32 << 20) / page_size);  // pages per read
    std::vector<uint8_t> raw((size_t)BLOCK * (size_t)page_size);
    std::vector<std::vector<int64_t>> found((size_t)BLOCK);  // per page, filled in parallel
    std::vector<char> bad_page((size_t)BLOCK);
    std::vector<int64_t> dead;
    for (int64_t part = lo / ids_per_part; rc == TSH_OK && hi > lo && part * ids_per_part < hi; ++part) {
      const std::string path = ngh_partition_path(dir, "graph", part, max_entries_per_dir);
      const int gfd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
      if (gfd < 0) continue;  // no file: every page reads as NghGraphPage.empty -> flags 0
      ++files;
      const int64_t first = part * ids_per_part;
      const int64_t n_pages = (std::min(ids_per_part, hi - first) + npg - 1) / npg;  // ... up to the range's last id
      for (int64_t p0 = std::max<int64_t>(0, lo - first) / npg; p0 < n_pages && rc == TSH_OK; p0 += BLOCK) {
        const int64_t nb = std::min(BLOCK, n_pages - p0);
        std::atomic<int> any_bytes{0};
        parallel_for((int32_t)nb, [&](int32_t b) {  // (every worker reads its own pages: as in load_rawvec_file)
          found[(size_t)b].clear();
          bad_page[(size_t)b] = 0;
          const size_t off = (size_t)b * (size_t)page_size;
          size_t got = 0;
          const off_t at = (off_t)(1 + p0 + b) * page_size;  // page 0 is the partition meta page
          while (got < (size_t)page_size) {
            const ssize_t r = pread(gfd, raw.data() + off + got, (size_t)page_size - got, at + (off_t)got);
            if (r <= 0) break;
            got += (size_t)r;
          }
          if (got) any_bytes.store(1, std::memory_order_relaxed);
          uint32_t plen = 0;
          int type = 0;
          bool bad = false;
          const uint8_t *pl = page_payload(raw.data() + off, got, &plen, &type, &bad);
          if (bad) {
            bad_page[(size_t)b] = 1;
            return;
          }
          if (pl && type != 6 /* BTreePageType.nghGraph.index */) {
            bad_page[(size_t)b] = 1;
            return;
          }
          // NghGraphPage.tryDecodePayload, ngh_page.dart:193-222 (null -> empty page)
          if (!pl || plen < 4) return;
          const uint32_t slot_count = rd16(pl), deg = rd16(pl + 2);
          if (deg == 0) return;
          const uint64_t ss = 2 + (uint64_t)deg * 4;
          if ((uint64_t)plen < 4 + slot_count * ss) return;
          const int64_t base = first + (p0 + b) * npg;
          for (uint32_t sl = 0; sl < slot_count && (int64_t)sl < npg; ++sl) {
            const int64_t id = base + sl;
            if (id >= hi) break;
            if (id >= lo && (pl[4 + sl * ss] & 0x01)) found[(size_t)b].push_back(id);
          }
        });
        if (!any_bytes.load()) break;  // past the end of the file: empty pages from here on
        for (int64_t b = 0; b < nb; ++b) {
          if (bad_page[(size_t)b]) {
            rc = set_err(TSH_E_FORMAT, "%s: page %lld has a bad header / CRC", path.c_str(), (long long)(1 + p0 + b));
            break;
          }
          dead.insert(dead.end(), found[(size_t)b].begin(), found[(size_t)b].end());
        }
      }
      close(gfd);
    }
    if (rc == TSH_OK && !dead.empty()) {
      rc = tsh_index_set_deleted(idx, dead.data(), (int64_t)dead.size());
      tombstones = (int64_t)dead.size();
    }
  }
  if (rc != TSH_OK) {
    const std::string keep = g_err;
    tsh_index_destroy(idx);
    return set_err(rc, "%s", keep.c_str());
  }
  if (info) {
    info->rows_loaded = rows_loaded;
    info->tombstones = tombstones;
    info->files_read = files;
    info->pages_absent = absent_pages;
    info->files_absent = absent_files;
  }
  *out = idx;
  return TSH_OK;
}
}  // namespace

extern "C" int32_t tsh_index_open_ngh(const char *ngh_dir, int32_t max_entries_per_dir, int32_t n_devices,
                                      tsh_index **out, tsh_ngh_info *info) {
  return open_ngh_impl(ngh_dir, max_entries_per_dir, n_devices, -1, 0, 0, out, info);
}

extern "C" int32_t tsh_index_open_ngh_shard(const char *ngh_dir, int32_t max_entries_per_dir, int32_t device,
                                            int32_t world, int32_t rank, tsh_index **out, tsh_ngh_info *info) {
  if (out) *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return set_err(TSH_E_BAD_ARG, "bad world / rank");
  return open_ngh_impl(ngh_dir, max_entries_per_dir, 1, device, world, rank, out, info);
}

