// np_writer.cpp -- writes the next-plaid crate's on-disk index directory (host only, no device work).
//
// The other side of np_loader.cpp: the canonical file set of write_index_from_encoded_chunks
// (next-plaid/src/index.rs:373-528), in the formats of SURVEY.md Appendix A, so that an index encoded with
// np_hip_encode_tokens (or exported from a device handle) is an index the crate itself loads:
//   centroids.npy '<f4' [K, d]; bucket_cutoffs.npy '<f4' [2^nbits - 1]; bucket_weights.npy '<f4' [2^nbits];
//   avg_residual.npy '<f4' [d]; cluster_threshold.npy '<f4' [1]; plan.json {nbits, num_chunks};
//   per chunk i of <= chunk_docs documents: {i}.metadata.json {num_documents, num_embeddings, embedding_offset},
//   doclens.{i}.json, {i}.codes.npy '<i8' [tokens], {i}.residuals.npy '|u1' [tokens, d * nbits / 8];
//   ivf.npy '<i8', ivf_lengths.npy '<i4' [K]: per centroid the ascending unique ids of the documents that use it
//   (index.rs:479-504), built here from the codes when the caller does not pass them; metadata.json last.
// NPY files are format 1.0 with the header dict padded to a 64-byte boundary (mmap.rs:1176-1250; byte-identical to
// numpy.save).  Every file goes through a temporary name + fsync + rename (utils.rs:16-60) so an interrupted writer never
// leaves a truncated file under a final name; the merged_*.npy caches of a previous index in the same directory are
// removed (they are derived data, mmap.rs:1714-1743).
#include "np_internal.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

namespace np {
namespace {

struct TmpFile {   // <final>.tmp.<pid>[.n] -> fsync -> rename
  std::string final_path, tmp_path;
  int fd = -1;
  int open_for(const std::string& path) {
    final_path = path;
    for (int attempt = 0; attempt < 8; ++attempt) {
      tmp_path = path + ".tmp." + std::to_string((long long)getpid()) + (attempt ? "." + std::to_string(attempt) : "");
      fd = open(tmp_path.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
      if (fd >= 0) return NP_OK;
      if (errno != EEXIST) break;
    }
    set_error("Failed to create %s: %s", tmp_path.c_str(), strerror(errno));
    return NP_ERR_IO;
  }
  int put(const void* data, size_t n) {
    const char* p = static_cast<const char*>(data);
    while (n > 0) {
      ssize_t w = write(fd, p, std::min<size_t>(n, (size_t)1 << 30));
      if (w < 0) {
        if (errno == EINTR) continue;
        set_error("Failed to write %s: %s", tmp_path.c_str(), strerror(errno));
        return NP_ERR_IO;
      }
      p += w;
      n -= (size_t)w;
    }
    return NP_OK;
  }
  int commit() {
    if (fsync(fd) != 0 || close(fd) != 0) {
      set_error("Failed to flush %s: %s", tmp_path.c_str(), strerror(errno));
      fd = -1;
      return NP_ERR_IO;
    }
    fd = -1;
    if (rename(tmp_path.c_str(), final_path.c_str()) != 0) {
      set_error("Failed to rename %s: %s", tmp_path.c_str(), strerror(errno));
      return NP_ERR_IO;
    }
    tmp_path.clear();
    return NP_OK;
  }
  ~TmpFile() {
    if (fd >= 0) close(fd);
    if (!tmp_path.empty()) unlink(tmp_path.c_str());
  }
};

// NPY 1.0: magic, version, u16 header length, "{'descr': '<f4', 'fortran_order': False, 'shape': (3, 4), }" padded with
// spaces so that the data starts on a 64-byte boundary, '\n' last
static int write_npy(const std::string& path, const char* descr, const int64_t* shape, int ndim, const void* data,
                     size_t bytes) {
  std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
  for (int i = 0; i < ndim; ++i) {
    dict += std::to_string((long long)shape[i]);
    if (ndim == 1 || i + 1 < ndim) dict += ",";
    if (i + 1 < ndim) dict += " ";
  }
  dict += "), }";
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict.append(pad, ' ');
  dict += '\n';
  if (dict.size() > 65535) {
    set_error("NPY header too long for %s", path.c_str());
    return NP_ERR_IO;
  }
  unsigned char head[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(dict.size() & 0xFF),
                            (unsigned char)(dict.size() >> 8)};
  TmpFile f;
  NP_TRY(f.open_for(path));
  NP_TRY(f.put(head, 10));
  NP_TRY(f.put(dict.data(), dict.size()));
  if (bytes) NP_TRY(f.put(data, bytes));
  return f.commit();
}

static int write_text(const std::string& path, const std::string& text) {
  TmpFile f;
  NP_TRY(f.open_for(path));
  NP_TRY(f.put(text.data(), text.size()));
  return f.commit();
}

static std::string fmt_double(double v) {   // shortest decimal that parses back to the same f64 (serde_json / ryu do the same)
  char buf[40];
  for (int prec = 1; prec <= 17; ++prec) {
    snprintf(buf, sizeof buf, "%.*g", prec, v);
    if (strtod(buf, nullptr) == v) break;
  }
  std::string s = buf;
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";   // a JSON float, as serde writes f64
  return s;
}

}  // namespace
}  // namespace np

using namespace np;

extern "C" int np_hip_index_write_dir(const char* index_dir, const np_index_arrays* a, const np_write_opts* o) {
  clear_error();
  if (!index_dir || !a) {
    set_error("np_hip_index_write_dir: NULL argument");
    return NP_ERR_INVALID_ARGUMENT;
  }
  const int64_t K = a->num_centroids, N = a->num_docs;
  const int dim = a->dim, nbits = a->nbits;
  if (nbits <= 0 || 8 % nbits != 0) {   // codec.rs:161-166
    set_error("Codec error: nbits must be a divisor of 8, got %d", nbits);
    return NP_ERR_CODEC;
  }
  if (K <= 0 || dim <= 0 || (dim * nbits) % 8 != 0 || N < 0 || !a->centroids || !a->bucket_weights ||
      (N > 0 && !a->doc_lengths)) {
    set_error("Shape error: write_dir needs centroids [K > 0, dim > 0], dim * nbits %% 8 == 0, bucket_weights, doc_lengths");
    return NP_ERR_SHAPE;
  }
  if (a->doc_begin != 0 || (a->num_documents_total != 0 && a->num_documents_total != N)) {
    set_error("np_hip_index_write_dir writes a whole index: doc_begin must be 0 and num_docs the whole corpus");
    return NP_ERR_INVALID_ARGUMENT;
  }
  const int64_t pd = (int64_t)dim * nbits / 8;
  const int64_t chunk_docs = (o && o->chunk_docs > 0) ? o->chunk_docs : 50000;   // IndexConfig.batch_size, index.rs:92
  int64_t T = 0;
  for (int64_t d = 0; d < N; ++d) {
    if (a->doc_lengths[d] < 0) {
      set_error("Index write failed: negative document length at %lld", (long long)d);
      return NP_ERR_INVALID_ARGUMENT;
    }
    T += a->doc_lengths[d];
  }
  if (T > 0 && (!a->codes || !a->residuals)) {
    set_error("np_hip_index_write_dir: codes / residuals missing for %lld tokens", (long long)T);
    return NP_ERR_INVALID_ARGUMENT;
  }
  for (int64_t t = 0; t < T; ++t)
    if (a->codes[t] < 0 || a->codes[t] >= K) {
      set_error("Index write failed: code %lld is outside [0, %lld)", (long long)a->codes[t], (long long)K);
      return NP_ERR_INVALID_ARGUMENT;
    }
  const std::string dir = index_dir;
  if (mkdir(dir.c_str(), 0755) != 0 && errno != EEXIST) {
    // create_dir_all: parents one by one
    for (size_t i = 1; i <= dir.size(); ++i)
      if (i == dir.size() || dir[i] == '/') {
        const std::string part = dir.substr(0, i);
        if (mkdir(part.c_str(), 0755) != 0 && errno != EEXIST) {
          set_error("Failed to create directory %s: %s", part.c_str(), strerror(errno));
          return NP_ERR_IO;
        }
      }
  }
  for (const char* stale : {"merged_codes.npy", "merged_codes.npy.manifest.json", "merged_residuals.npy",
                            "merged_residuals.npy.manifest.json"})
    unlink((dir + "/" + stale).c_str());
  // The crate loads bucket_cutoffs.npy whenever it exists (codec.rs:491-500): a copy left by a previous index in this
  // directory would pair the old cutoffs with the new weights / nbits on the update path.
  if (!(o && o->bucket_cutoffs)) unlink((dir + "/bucket_cutoffs.npy").c_str());

  // codec files
  {
    const int64_t s2[2] = {K, dim};
    NP_TRY(write_npy(dir + "/centroids.npy", "<f4", s2, 2, a->centroids, (size_t)K * dim * 4));
    const int64_t nb = (int64_t)1 << nbits, nc = nb - 1, one = 1, dd = dim;
    if (o && o->bucket_cutoffs) NP_TRY(write_npy(dir + "/bucket_cutoffs.npy", "<f4", &nc, 1, o->bucket_cutoffs, (size_t)nc * 4));
    NP_TRY(write_npy(dir + "/bucket_weights.npy", "<f4", &nb, 1, a->bucket_weights, (size_t)nb * 4));
    std::vector<float> zeros((size_t)dim, 0.f);
    NP_TRY(write_npy(dir + "/avg_residual.npy", "<f4", &dd, 1, (o && o->avg_residual) ? o->avg_residual : zeros.data(),
                     (size_t)dim * 4));
    const float thr = o ? o->cluster_threshold : 0.f;
    NP_TRY(write_npy(dir + "/cluster_threshold.npy", "<f4", &one, 1, &thr, 4));
  }
  const int64_t n_chunks = (N + chunk_docs - 1) / chunk_docs;   // chunks.len(): 0 for an empty corpus (index.rs:436-474)
  NP_TRY(write_text(dir + "/plan.json", "{\n  \"nbits\": " + std::to_string(nbits) + ",\n  \"num_chunks\": " +
                                            std::to_string((long long)n_chunks) + "\n}\n"));
  // chunks
  int64_t tok = 0;
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int64_t d0 = c * chunk_docs, d1 = std::min(N, d0 + chunk_docs);
    int64_t nt = 0;
    std::string lens = "[";
    for (int64_t d = d0; d < d1; ++d) {
      nt += a->doc_lengths[d];
      lens += std::to_string((long long)a->doc_lengths[d]);
      if (d + 1 < d1) lens += ",";
    }
    lens += "]";
    const std::string ci = std::to_string((long long)c);
    NP_TRY(write_text(dir + "/" + ci + ".metadata.json",
                      "{\n  \"num_documents\": " + std::to_string((long long)(d1 - d0)) + ",\n  \"num_embeddings\": " +
                          std::to_string((long long)nt) + ",\n  \"embedding_offset\": " + std::to_string((long long)tok) + "\n}"));
    NP_TRY(write_text(dir + "/doclens." + ci + ".json", lens));
    NP_TRY(write_npy(dir + "/" + ci + ".codes.npy", "<i8", &nt, 1, a->codes ? a->codes + tok : nullptr, (size_t)nt * 8));
    const int64_t rs[2] = {nt, pd};
    NP_TRY(write_npy(dir + "/" + ci + ".residuals.npy", "|u1", rs, 2, a->residuals ? a->residuals + tok * pd : nullptr,
                     (size_t)nt * pd));
    tok += nt;
  }
  // chunk files of a previous, longer index in the same directory (the loader walks num_chunks, an update appends to the
  // last chunk it finds): remove {i}.* for i >= n_chunks until a gap
  for (int64_t c = n_chunks;; ++c) {
    const std::string ci = std::to_string((long long)c);
    int gone = 0;
    for (const std::string& f : {ci + ".codes.npy", ci + ".residuals.npy", ci + ".metadata.json", "doclens." + ci + ".json"})
      gone += unlink((dir + "/" + f).c_str()) == 0;
    if (!gone) break;
  }
  // IVF: the caller's (e.g. np_hip_index_export of a device handle), or counted here from the codes
  {
    std::vector<int32_t> own_len;
    std::vector<int64_t> own_ivf;
    const int64_t* ivf = a->ivf;
    const int32_t* ivf_len = a->ivf_lengths;
    if (!ivf_len || (!ivf && T > 0)) {
      // (code, doc) pairs, doc ascending inside a code because documents are walked in order; duplicates are adjacent
      std::vector<int64_t> start((size_t)K + 1, 0);
      for (int64_t t = 0; t < T; ++t) ++start[(size_t)a->codes[t] + 1];
      for (int64_t k = 0; k < K; ++k) start[(size_t)k + 1] += start[(size_t)k];
      std::vector<int64_t> slot((size_t)T), fill(start.begin(), start.end() - 1);
      int64_t t = 0;
      for (int64_t d = 0; d < N; ++d)
        for (int64_t i = 0; i < a->doc_lengths[d]; ++i, ++t) slot[(size_t)fill[(size_t)a->codes[t]]++] = d;
      own_len.assign((size_t)K, 0);
      own_ivf.reserve((size_t)T);
      for (int64_t k = 0; k < K; ++k) {
        int64_t last = -1;
        for (int64_t i = start[(size_t)k]; i < start[(size_t)k + 1]; ++i)
          if (slot[(size_t)i] != last) {
            last = slot[(size_t)i];
            own_ivf.push_back(last);
            ++own_len[(size_t)k];
          }
      }
      ivf = own_ivf.data();
      ivf_len = own_len.data();
    }
    int64_t total = 0;
    for (int64_t k = 0; k < K; ++k) {
      if (ivf_len[k] < 0) {
        set_error("Index write failed: negative posting-list length at centroid %lld", (long long)k);
        return NP_ERR_INVALID_ARGUMENT;
      }
      total += ivf_len[k];
    }
    if (ivf == a->ivf && ivf) {   // caller-supplied lists: unique ascending document ids in [0, N) per centroid (index.rs:479-504)
      int64_t i = 0;
      for (int64_t k = 0; k < K; ++k) {
        int64_t last = -1;
        for (int32_t j = 0; j < ivf_len[k]; ++j, ++i) {
          if (ivf[i] <= last || ivf[i] >= N) {
            set_error("Index write failed: posting list of centroid %lld is not strictly ascending inside [0, %lld) at entry %d",
                      (long long)k, (long long)N, j);
            return NP_ERR_INVALID_ARGUMENT;
          }
          last = ivf[i];
        }
      }
    } else if (total > 0 && !ivf) {
      set_error("np_hip_index_write_dir: ivf_lengths without ivf");
      return NP_ERR_INVALID_ARGUMENT;
    }
    NP_TRY(write_npy(dir + "/ivf.npy", "<i8", &total, 1, ivf, (size_t)total * 8));
    NP_TRY(write_npy(dir + "/ivf_lengths.npy", "<i4", &K, 1, ivf_len, (size_t)K * 4));
  }
  const double avg = N > 0 ? (double)T / (double)N : 0.0;
  NP_TRY(write_text(dir + "/metadata.json",
                    "{\n  \"num_chunks\": " + std::to_string((long long)n_chunks) + ",\n  \"nbits\": " + std::to_string(nbits) +
                        ",\n  \"num_partitions\": " + std::to_string((long long)K) + ",\n  \"num_embeddings\": " +
                        std::to_string((long long)T) + ",\n  \"avg_doclen\": " + fmt_double(avg) + ",\n  \"num_documents\": " +
                        std::to_string((long long)N) + ",\n  \"embedding_dim\": " + std::to_string(dim) +
                        ",\n  \"next_plaid_compatible\": true\n}"));
  int dfd = open(dir.c_str(), O_RDONLY);
  if (dfd >= 0) {
    (void)fsync(dfd);
    close(dfd);
  }
  return NP_OK;
}
