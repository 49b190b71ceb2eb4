// np_loader.cpp -- reads the next-plaid crate's on-disk index directory, unchanged.
//
// Replaces the file-parsing half of MmapIndex::load (next-plaid/src/index.rs:1026-1139):
//   metadata.json                      index.rs:104-155   (num_documents inferred from doclens if 0)
//   centroids.npy, bucket_weights.npy  codec.rs:548-612 ('<f4'; fast-plaid's '<f2' is widened, mmap.rs:1757-1778)
//   ivf.npy (<i8), ivf_lengths.npy (<i4; fast-plaid writes <i8, mmap.rs:1780-1789)
//   doclens.{i}.json, {i}.codes.npy (<i8), {i}.residuals.npy (|u1 or <u1, mmap.rs:1791-1808)
// NPY v1.0 / v2.0 headers as parsed by mmap.rs:659-749.  The merged_*.npy caches are NOT used:
// chunk files are mapped and concatenated in chunk order, which is what merge_codes_chunks /
// merge_residuals_chunks (mmap.rs:1266-1704) produce minus the never-addressed padding rows.
#include "np_internal.h"

#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

namespace np {

// ---- thread-local error string -------------------------------------------------------------
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char* last_error() { return g_err.c_str(); }
void clear_error() { g_err.clear(); }

HostIndex::~HostIndex() {
  for (auto& m : maps)
    if (m.first && m.second) munmap(m.first, m.second);
}

// ---- files -----------------------------------------------------------------------------------
static int map_file(const std::string& path, HostIndex* hi, const uint8_t** data, size_t* size, int err_code) {
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    set_error("Failed to open %s: %s", path.c_str(), strerror(errno));
    return err_code;
  }
  struct stat st;
  if (fstat(fd, &st) != 0) {
    set_error("Failed to stat %s: %s", path.c_str(), strerror(errno));
    close(fd);
    return NP_ERR_IO;
  }
  *size = (size_t)st.st_size;
  if (*size == 0) {
    close(fd);
    *data = nullptr;
    return NP_OK;
  }
  void* p = mmap(nullptr, *size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    set_error("Failed to mmap %s: %s", path.c_str(), strerror(errno));
    return NP_ERR_IO;
  }
  hi->maps.emplace_back(p, *size);
  *data = (const uint8_t*)p;
  return NP_OK;
}

static int read_text(const std::string& path, std::string* out, int err_code) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    set_error("Failed to open %s: %s", path.c_str(), strerror(errno));
    return err_code;
  }
  char buf[65536];
  size_t n;
  out->clear();
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  fclose(f);
  return NP_OK;
}

// ---- NPY (mmap.rs:659-749) ---------------------------------------------------------------------
struct Npy {
  std::string descr;
  bool fortran = false;
  std::vector<int64_t> shape;
  const uint8_t* data = nullptr;
  size_t data_bytes = 0;
  int64_t count() const {
    int64_t c = 1;
    for (auto s : shape) c *= s;
    return c;
  }
};

static int parse_npy(const std::string& path, const uint8_t* m, size_t len, Npy* out) {
  static const uint8_t magic[6] = {0x93, 'N', 'U', 'M', 'P', 'Y'};
  if (len < 10) {
    set_error("NPY file %s too small: %zu bytes", path.c_str(), len);
    return NP_ERR_INDEX_LOAD;
  }
  if (memcmp(m, magic, 6) != 0) {
    set_error("Invalid NPY magic in %s", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  int major = m[6];
  size_t hlen, hstart;
  if (major == 1) {
    hlen = (size_t)m[8] | ((size_t)m[9] << 8);
    hstart = 10;
  } else if (major == 2 || major == 3) {
    if (len < 12) {
      set_error("NPY v2 file %s too small", path.c_str());
      return NP_ERR_INDEX_LOAD;
    }
    hlen = (size_t)m[8] | ((size_t)m[9] << 8) | ((size_t)m[10] << 16) | ((size_t)m[11] << 24);
    hstart = 12;
  } else {
    set_error("Unsupported NPY version: %d (%s)", major, path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  size_t hend = hstart + hlen;
  if (len < hend) {
    set_error("NPY header exceeds file size for %s", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  std::string h((const char*)m + hstart, hlen);
  size_t p = h.find("'descr':");
  if (p != std::string::npos) {
    size_t a = h.find('\'', p + 8);
    size_t b = a == std::string::npos ? a : h.find('\'', a + 1);
    if (b != std::string::npos) out->descr = h.substr(a + 1, b - a - 1);
  }
  out->fortran = h.find("'fortran_order': True") != std::string::npos;
  p = h.find("'shape':");
  if (p == std::string::npos) {
    set_error("No shape in NPY header (%s)", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  size_t a = h.find('(', p), b = h.find(')', p);
  if (a == std::string::npos || b == std::string::npos || b < a) {
    set_error("No shape tuple in NPY header (%s)", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  out->shape.clear();
  const char* s = h.c_str() + a + 1;
  const char* e = h.c_str() + b;
  while (s < e) {
    while (s < e && (*s == ' ' || *s == ',')) ++s;
    if (s >= e) break;
    char* endp = nullptr;
    long long v = strtoll(s, &endp, 10);
    if (endp == s) {
      set_error("Invalid shape dimension in %s", path.c_str());
      return NP_ERR_INDEX_LOAD;
    }
    out->shape.push_back((int64_t)v);
    s = endp;
  }
  out->data = m + hend;
  out->data_bytes = len - hend;
  return NP_OK;
}

static int elem_size(const std::string& d) {
  if (d.size() < 3) return 0;
  return atoi(d.c_str() + 2);
}

static int open_npy(const std::string& path, HostIndex* hi, Npy* out, const char* want_kind, int want_size,
                    size_t want_ndim) {
  const uint8_t* m;
  size_t len;
  NP_TRY(map_file(path, hi, &m, &len, NP_ERR_INDEX_LOAD));
  NP_TRY(parse_npy(path, m, len, out));
  if (out->fortran && out->shape.size() > 1) {
    set_error("fortran_order NPY not supported: %s", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  if (out->descr.size() < 3 || !strchr(want_kind, out->descr[1]) || elem_size(out->descr) != want_size ||
      out->descr[0] == '>') {
    set_error("Unexpected dtype '%s' in %s", out->descr.c_str(), path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  if (out->shape.size() != want_ndim) {
    set_error("Unexpected rank %zu in %s", out->shape.size(), path.c_str());
    return NP_ERR_SHAPE;
  }
  if ((size_t)out->count() * (size_t)want_size > out->data_bytes) {
    set_error("NPY file size too small for %lld elements: %s", (long long)out->count(), path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  return NP_OK;
}

// IEEE half -> float (what the reference's convert_f16_to_f32_npy does with half::f16::to_f32, mmap.rs:1760-1778)
static float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: renormalise
      int e = -1;
      do {
        ++e;
        man <<= 1;
      } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

// f32 array that fast-plaid may have written as '<f2' (mmap.rs:1757-1778 converts those files to '<f4' on load;
// here the widened copy lives in an owned buffer and the file is left untouched).
static int open_npy_f32(const std::string& path, HostIndex* hi, Npy* out, size_t want_ndim, const float** data) {
  const uint8_t* m;
  size_t len;
  NP_TRY(map_file(path, hi, &m, &len, NP_ERR_INDEX_LOAD));
  NP_TRY(parse_npy(path, m, len, out));
  if (out->fortran && out->shape.size() > 1) {
    set_error("fortran_order NPY not supported: %s", path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  const int es = elem_size(out->descr);
  if (out->descr.size() < 3 || out->descr[1] != 'f' || (es != 4 && es != 2) || out->descr[0] == '>') {
    set_error("Unexpected dtype '%s' in %s", out->descr.c_str(), path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  if (out->shape.size() != want_ndim) {
    set_error("Unexpected rank %zu in %s", out->shape.size(), path.c_str());
    return NP_ERR_SHAPE;
  }
  if ((size_t)out->count() * (size_t)es > out->data_bytes) {
    set_error("NPY file size too small for %lld elements: %s", (long long)out->count(), path.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  if (es == 4) {
    *data = (const float*)out->data;
    return NP_OK;
  }
  hi->owned.emplace_back((size_t)std::max<int64_t>(out->count(), 1) * 4);
  float* o = (float*)hi->owned.back().data();
  for (int64_t i = 0; i < out->count(); ++i) {
    uint16_t h;
    memcpy(&h, out->data + 2 * i, 2);
    o[i] = half_to_float(h);
  }
  *data = o;
  return NP_OK;
}

// ---- JSON (only what metadata.json / doclens.N.json need) ---------------------------------------
static bool json_number(const std::string& j, const char* key, double* out) {
  std::string k = std::string("\"") + key + "\"";
  size_t p = j.find(k);
  if (p == std::string::npos) return false;
  p = j.find(':', p + k.size());
  if (p == std::string::npos) return false;
  ++p;
  while (p < j.size() && isspace((unsigned char)j[p])) ++p;
  char* endp = nullptr;
  double v = strtod(j.c_str() + p, &endp);
  if (endp == j.c_str() + p) return false;
  *out = v;
  return true;
}

static int json_int_array(const std::string& path, const std::string& j, std::vector<int64_t>* out) {
  size_t a = j.find('['), b = j.rfind(']');
  if (a == std::string::npos || b == std::string::npos || b < a) {
    set_error("JSON error: expected an array in %s", path.c_str());
    return NP_ERR_IO;
  }
  const char* s = j.c_str() + a + 1;
  const char* e = j.c_str() + b;
  while (s < e) {
    while (s < e && (isspace((unsigned char)*s) || *s == ',')) ++s;
    if (s >= e) break;
    char* endp = nullptr;
    long long v = strtoll(s, &endp, 10);
    if (endp == s) {
      set_error("JSON error: bad integer in %s", path.c_str());
      return NP_ERR_IO;
    }
    out->push_back((int64_t)v);
    s = endp;
    if (s < e && (*s == '.' || *s == 'e' || *s == 'E')) {  // tolerate "300.0"
      strtod(endp, &endp);
      s = endp;
    }
  }
  return NP_OK;
}

// ---- index.rs:1026-1139 ------------------------------------------------------------------------------
int load_index_dir(const char* dir, HostIndex* hi) {
  std::string base(dir ? dir : "");
  if (base.empty()) {
    set_error("Index load failed: empty path");
    return NP_ERR_INDEX_LOAD;
  }
  if (base.back() != '/') base += '/';
  std::string meta;
  if (read_text(base + "metadata.json", &meta, NP_ERR_INDEX_LOAD) != NP_OK) {
    std::string e = last_error();
    set_error("Index load failed: Failed to open metadata: %s", e.c_str());
    return NP_ERR_INDEX_LOAD;
  }
  double v;
  if (!json_number(meta, "num_chunks", &v)) {
    set_error("JSON error: num_chunks missing in metadata.json");
    return NP_ERR_IO;
  }
  int64_t num_chunks = (int64_t)v;
  if (!json_number(meta, "nbits", &v)) {
    set_error("Index load failed: nbits not found in metadata");
    return NP_ERR_INDEX_LOAD;
  }
  hi->nbits = (int32_t)v;
  if (hi->nbits <= 0 || 8 % hi->nbits != 0) {  // codec.rs:161-166
    set_error("Codec error: nbits must be a divisor of 8, got %d", hi->nbits);
    return NP_ERR_CODEC;
  }
  hi->num_embeddings_total = json_number(meta, "num_embeddings", &v) ? (int64_t)v : 0;
  hi->avg_doclen = json_number(meta, "avg_doclen", &v) ? v : 0.0;

  Npy cen, bw, ivf, ivl;
  NP_TRY(open_npy_f32(base + "centroids.npy", hi, &cen, 2, &hi->centroids));
  hi->K = cen.shape[0];
  hi->dim = (int32_t)cen.shape[1];
  {
    struct stat st;
    if (stat((base + "bucket_weights.npy").c_str(), &st) != 0) {  // codec.rs:428-431
      set_error("Codec error: bucket_weights required for decompression");
      return NP_ERR_CODEC;
    }
  }
  NP_TRY(open_npy_f32(base + "bucket_weights.npy", hi, &bw, 1, &hi->bucket_weights));
  if (bw.shape[0] != (1 << hi->nbits)) {
    set_error("Codec error: bucket_weights has %lld entries, expected %d", (long long)bw.shape[0], 1 << hi->nbits);
    return NP_ERR_CODEC;
  }
  NP_TRY(open_npy(base + "ivf.npy", hi, &ivf, "i", 8, 1));
  hi->ivf = (const int64_t*)ivf.data;
  hi->ivf_size = ivf.shape[0];
  {
    // ivf_lengths: <i4 (next-plaid) or <i8 (fast-plaid, converted on load by the reference)
    const uint8_t* m;
    size_t len;
    std::string p = base + "ivf_lengths.npy";
    NP_TRY(map_file(p, hi, &m, &len, NP_ERR_INDEX_LOAD));
    NP_TRY(parse_npy(p, m, len, &ivl));
    int es = elem_size(ivl.descr);
    if (ivl.shape.size() != 1 || (es != 4 && es != 8) || ivl.descr[1] != 'i' ||
        (size_t)ivl.shape[0] * es > ivl.data_bytes) {
      set_error("Unexpected dtype/shape '%s' in %s", ivl.descr.c_str(), p.c_str());
      return NP_ERR_INDEX_LOAD;
    }
    if (es == 4) {
      hi->ivf_lengths = (const int32_t*)ivl.data;
    } else {
      hi->owned.emplace_back((size_t)ivl.shape[0] * 4);
      int32_t* o = (int32_t*)hi->owned.back().data();
      const int64_t* s = (const int64_t*)ivl.data;
      for (int64_t i = 0; i < ivl.shape[0]; ++i) o[i] = (int32_t)s[i];
      hi->ivf_lengths = o;
    }
    if (ivl.shape[0] != hi->K) {
      set_error("Shape error: ivf_lengths has %lld entries, centroids has %lld rows", (long long)ivl.shape[0],
                (long long)hi->K);
      return NP_ERR_SHAPE;
    }
  }
  int64_t ivf_sum = 0;
  for (int64_t i = 0; i < hi->K; ++i) ivf_sum += hi->ivf_lengths[i];
  if (ivf_sum > hi->ivf_size) {
    set_error("Index load failed: ivf.npy holds %lld ids, ivf_lengths sums to %lld", (long long)hi->ivf_size,
              (long long)ivf_sum);
    return NP_ERR_INDEX_LOAD;
  }

  const int64_t pd = (int64_t)hi->dim * hi->nbits / 8;
  hi->doc_begin = 0;
  int64_t total_tokens = 0;
  for (int64_t c = 0; c < num_chunks; ++c) {
    char name[64];
    std::string txt;
    snprintf(name, sizeof name, "doclens.%lld.json", (long long)c);
    if (read_text(base + name, &txt, NP_ERR_IO) != NP_OK) return NP_ERR_IO;
    size_t before = hi->doc_lengths.size();
    NP_TRY(json_int_array(base + name, txt, &hi->doc_lengths));
    int64_t chunk_tokens = 0;
    for (size_t i = before; i < hi->doc_lengths.size(); ++i) {
      if (hi->doc_lengths[i] < 0) {
        set_error("Index load failed: negative doc length in %s", name);
        return NP_ERR_INDEX_LOAD;
      }
      chunk_tokens += hi->doc_lengths[i];
    }
    Npy codes, res;
    snprintf(name, sizeof name, "%lld.codes.npy", (long long)c);
    NP_TRY(open_npy(base + name, hi, &codes, "i", 8, 1));
    snprintf(name, sizeof name, "%lld.residuals.npy", (long long)c);
    NP_TRY(open_npy(base + name, hi, &res, "u", 1, 2));
    if (res.shape[1] != pd) {
      set_error("Shape error: residuals have %lld columns, expected dim*nbits/8 = %lld", (long long)res.shape[1],
                (long long)pd);
      return NP_ERR_SHAPE;
    }
    if (codes.shape[0] < chunk_tokens || res.shape[0] < chunk_tokens) {
      set_error("Index load failed: chunk %lld holds %lld codes / %lld residual rows, doclens sum to %lld",
                (long long)c, (long long)codes.shape[0], (long long)res.shape[0], (long long)chunk_tokens);
      return NP_ERR_INDEX_LOAD;
    }
    HostChunk hc;
    hc.codes = (const int64_t*)codes.data;
    hc.residuals = res.data;
    hc.n_tokens = chunk_tokens;
    hi->chunks.push_back(hc);
    total_tokens += chunk_tokens;
  }
  hi->num_documents_total = (int64_t)hi->doc_lengths.size();
  if (hi->num_embeddings_total == 0) hi->num_embeddings_total = total_tokens;
  if (hi->avg_doclen == 0.0 && hi->num_documents_total > 0)
    hi->avg_doclen = (double)total_tokens / (double)hi->num_documents_total;
  return NP_OK;
}

}  // namespace np
