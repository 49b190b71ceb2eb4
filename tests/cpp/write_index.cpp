// next_plaid::write_index (index.rs:373-528 through np_hip_index_write_dir) from the C++ host mirror:
//   write_index <dir>   writes a small deterministic index, then parses it back with the loader (host only)
// Prints "ok <docs> <tokens> <K> <dim> <nbits>" or "error <message>".
#include <cstdio>
#include <vector>

#include "../../next-plaid_amd/cpp/next_plaid.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const size_t K = 12, dim = 24, N = 37;
  const int nbits = 2;
  std::vector<float> cen(K * dim);
  for (size_t i = 0; i < cen.size(); ++i) cen[i] = (float)((i * 37 % 101) - 50) / 64.f;
  next_plaid::IndexFiles f;
  f.num_centroids = K;
  f.dim = dim;
  f.nbits = nbits;
  f.centroids = cen.data();
  f.bucket_weights = {-0.03f, -0.01f, 0.01f, 0.03f};
  f.bucket_cutoffs = {-0.02f, 0.f, 0.02f};
  f.chunk_docs = 10;
  size_t T = 0;
  for (size_t d = 0; d < N; ++d) {
    f.doc_lengths.push_back((int64_t)(d % 5));   // some empty documents
    T += d % 5;
  }
  std::vector<int64_t> codes(T);
  std::vector<uint8_t> res(T * dim * nbits / 8);
  for (size_t t = 0; t < T; ++t) codes[t] = (int64_t)(t * 7 % K);
  for (size_t i = 0; i < res.size(); ++i) res[i] = (uint8_t)(i * 13);
  f.codes = codes.data();
  f.residuals = res.data();
  try {
    next_plaid::write_index(argv[1], f);
    np_info info;
    if (np_hip_index_probe_dir(argv[1], &info) != 0) {
      std::printf("error %s\n", np_hip_last_error());
      return 1;
    }
    std::printf("ok %lld %lld %lld %d %d\n", (long long)info.num_documents, (long long)info.num_embeddings,
                (long long)info.num_partitions, info.embedding_dim, info.nbits);
    f.bucket_weights.pop_back();
    try {
      next_plaid::write_index(argv[1], f);
      std::printf("no error\n");
    } catch (const next_plaid::Error& e) {
      std::printf("codec %d\n", (int)e.kind);
    }
  } catch (const next_plaid::Error& e) {
    std::printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
