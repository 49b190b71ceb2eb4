// Exercises next_plaid.hpp's accelerator policy (lib.rs:71-84, cuda.rs:52-182 precedent) with a stand-in CPU path.
//   fallback_policy <index_dir> <dim> <hook:0|1>
// Prints one line: "device" | "cpu <n results> broken=<0|1>" | "error <kind>".
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../next-plaid_amd/cpp/next_plaid.hpp"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const size_t dim = std::strtoul(argv[2], nullptr, 10);
  if (std::atoi(argv[3]))
    next_plaid::set_cpu_fallback([](const std::string&, const next_plaid::Query*, size_t n, size_t, const next_plaid::SearchParameters&,
                                    bool, const std::vector<int64_t>*) {
      std::vector<next_plaid::QueryResult> out(n);
      for (size_t i = 0; i < n; ++i) {
        out[i].query_id = i;
        out[i].passage_ids = {42};
        out[i].scores = {1.0f};
      }
      return out;
    });
  try {
    auto index = next_plaid::MmapIndex::load(argv[1]);
    index.cpu_dim = dim;
    std::vector<float> q(4 * dim, 0.1f);
    next_plaid::SearchParameters p;
    auto r = index.search(q.data(), 4, p);
    if (index.on_device()) std::printf("device\n");
    else std::printf("cpu %zu broken=%d\n", r.passage_ids.size(), (int)next_plaid::is_hip_broken());
    // second load: with the flag raised the device is not touched again (cuda.rs get_global_context fast path)
    auto again = next_plaid::MmapIndex::load(argv[1]);
    std::printf("again %s\n", again.on_device() ? "device" : "cpu");
    next_plaid::clear_hip_broken();
    std::printf("cleared broken=%d\n", (int)next_plaid::is_hip_broken());
    // geometry accessors stay valid on the CPU hand-off (host-only parse), device-only methods say so
    std::printf("geom %zu %zu %zu\n", again.num_documents(), again.embedding_dim(), again.num_partitions());
    again.reload();   // index.rs:1767: same policy as load, geometry refreshed from the directory
    std::printf("reloaded %s %zu\n", again.on_device() ? "device" : "cpu", again.num_documents());
    if (!again.on_device()) {
      try {
        (void)again.decompress_documents({0});
        std::printf("decompress ok\n");
      } catch (const next_plaid::Error& e) {
        std::printf("decompress error %d\n", (int)e.kind);
      }
    }
  } catch (const next_plaid::Error& e) {
    std::printf("error %d\n", (int)e.kind);
  }
  return 0;
}
