// search_cli.cpp -- tiny C++ caller of the crate-shaped host API (next_plaid.hpp) over the C ABI.
//   np_search <index_dir> <queries.f32> <n_queries> <tokens_per_query> [top_k] [n_ivf_probe] [n_full_scores] [t_cs|-1]
// queries.f32: raw little-endian f32 [n_queries * tokens_per_query, dim].  Prints one JSON line per query.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "next_plaid.hpp"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s index_dir queries.f32 n_queries tokens_per_query [top_k] [nprobe] [n_full] [t_cs]\n", argv[0]);
    return 2;
  }
  try {
    auto index = next_plaid::MmapIndex::load(argv[1]);
    const size_t nq = std::strtoul(argv[3], nullptr, 10), lq = std::strtoul(argv[4], nullptr, 10);
    const size_t dim = index.embedding_dim();
    std::vector<float> buf(nq * lq * dim);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f || std::fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) {
      std::fprintf(stderr, "cannot read %zu floats from %s\n", buf.size(), argv[2]);
      return 2;
    }
    std::fclose(f);
    next_plaid::SearchParameters p;
    if (argc > 5) p.top_k = std::strtoul(argv[5], nullptr, 10);
    if (argc > 6) p.n_ivf_probe = std::strtoul(argv[6], nullptr, 10);
    if (argc > 7) p.n_full_scores = std::strtoul(argv[7], nullptr, 10);
    if (argc > 8) {
      float t = std::strtof(argv[8], nullptr);
      if (t < 0) p.centroid_score_threshold.reset(); else p.centroid_score_threshold = t;
    }
    std::vector<next_plaid::Query> qs(nq);
    for (size_t i = 0; i < nq; ++i) qs[i] = {buf.data() + i * lq * dim, lq};
    auto res = index.search_batch(qs.data(), nq, p, /*parallel=*/true);
    for (const auto& r : res) {
      std::printf("{\"query_id\": %zu, \"passage_ids\": [", r.query_id);
      for (size_t j = 0; j < r.passage_ids.size(); ++j) std::printf("%s%lld", j ? ", " : "", (long long)r.passage_ids[j]);
      std::printf("], \"scores\": [");
      for (size_t j = 0; j < r.scores.size(); ++j) std::printf("%s%.9g", j ? ", " : "", r.scores[j]);
      std::printf("]}\n");
    }
  } catch (const next_plaid::Error& e) {
    std::fprintf(stderr, "next-plaid error %d: %s\n", (int)e.kind, e.what());
    return 1;
  }
  return 0;
}
