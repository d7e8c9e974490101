// ppcsr_cli — command-line driver with the reference's flags and stdout protocol (reference: src/main.cpp:111-189),
// running the update path on MI355X through the host shims.  Flags (same prefix matching, same order sensitivity):
//   -threads=N  -size=N  -lock_free  -insert  -delete  -ppcsr  -pppcsr  -pppcsrnuma  -partitions_per_domain=N
//   -core_graph=FILE  -update_file=FILE        additions:  -gpus=N (devices 0..N-1 for -pppcsr*)  -device=D  -verify
//   -bulk_core (with -ppcsr): load the core graph through the NON-parity bulk build instead of single inserts (same graph,
//               different slot layout; the timed update phase is unchanged)
// Edge-list lines: "src<sep>dst[<sep>1|0]" with a single separator character (main.cpp:29-62); the optional third
// column selects ADD (1) / DELETE (0), otherwise the default bound when -update_file= is parsed applies.
// The bench scripts of the reference scrape the SECOND "Elapsed wall clock time:" line (benchmark-strong-scaling.sh:116).
#include <algorithm>
#include <chrono>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "thread_pool.h"

enum class Operation { READ, ADD, DELETE };
using OpList = std::vector<std::tuple<Operation, int, int>>;

static bool starts_with(const std::string &s, const char *p) { return s.rfind(p, 0) == 0; }
static std::string after(const std::string &s, const char *p) { return s.substr(std::string(p).size()); }

static std::pair<OpList, int> read_input(const std::string &filename, Operation default_op) {
  std::ifstream f(filename);
  if (!f.good()) {
    std::cerr << "Invalid file" << std::endl;
    std::exit(EXIT_FAILURE);
  }
  OpList out;
  int num_nodes = 0;
  std::string line;
  while (std::getline(f, line)) {
    std::size_t p1 = 0, p2 = 0;
    const int src = std::stoi(line, &p1);
    const int dst = std::stoi(line.substr(p1 + 1), &p2);
    num_nodes = std::max(num_nodes, std::max(src, dst));
    Operation op = default_op;
    const std::size_t third = p1 + 1 + p2 + 1;
    if (third < line.length()) {
      if (line[third] == '1') op = Operation::ADD;
      else if (line[third] == '0') op = Operation::DELETE;
      else std::cerr << "Invalid operation";
    }
    out.emplace_back(op, src, dst);
  }
  return {out, num_nodes};
}

template <typename Pool>
static void update_existing_graph(const OpList &input, Pool *pool, int threads, int size) {
  for (int i = 0; i < size; i++) {
    switch (std::get<0>(input[i])) {
      case Operation::ADD: pool->submit_add(i % threads, std::get<1>(input[i]), std::get<2>(input[i])); break;
      case Operation::DELETE: pool->submit_delete(i % threads, std::get<1>(input[i]), std::get<2>(input[i])); break;
      case Operation::READ: std::cerr << "Not implemented\n"; break;
    }
  }
  pool->start(threads);
  pool->stop();
}

template <typename Pool>
static void update_existing_graph(const OpList &input, Pool *pool, int threads, int size);
template <typename Pool>
static void bulk_core(const OpList &core, Pool *pool, int threads) {  // partitioned structures: no bulk path, load as usual
  std::cerr << "-bulk_core applies to -ppcsr only; loading the core graph through single inserts" << std::endl;
  update_existing_graph(core, pool, threads, (int)core.size());
}
static void bulk_core(const OpList &core, ThreadPool *pool, int) {  // phase 1 through ppcsr_bulk_build; prints the phase-1 time line too
  std::vector<ppcsr_op> adds;
  adds.reserve(core.size());
  for (const auto &c : core)
    if (std::get<0>(c) == Operation::ADD) adds.push_back(ppcsr_op{(uint32_t)std::get<1>(c), (uint32_t)std::get<2>(c), 1u});
  const auto t0 = std::chrono::steady_clock::now();
  pool->pcsr->bulk_build(adds);
  const auto t1 = std::chrono::steady_clock::now();
  std::cout << "Elapsed wall clock time: " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << std::endl;
}

template <typename Pool>
static void execute(int threads, int size, const OpList &core, const OpList &updates, Pool *pool, bool verify, bool bulk = false) {
  if (bulk) bulk_core(core, pool, threads);
  else update_existing_graph(core, pool, threads, (int)core.size());  // phase 1: load the core graph
  update_existing_graph(updates, pool, threads, size);           // phase 2: the timed updates
  if (verify) {  // the reference's commented-out debugging check (main.cpp:93-106), enabled by -verify
    long missing = 0;
    for (int i = 0; i < size; i++)
      if (std::get<0>(updates[i]) == Operation::ADD && !pool->pcsr->edge_exists(std::get<1>(updates[i]), std::get<2>(updates[i]))) missing++;
    std::cout << "verify: " << missing << " inserted updates not found" << std::endl;
  }
}

int main(int argc, char *argv[]) {
  int threads = 8, size = 1000000, num_nodes = 0, partitions_per_domain = 1, gpus = 1, device = 0;
  bool lock_search = true, insert = true, verify = false, bulk = false;
  enum class Version { PPCSR, PPPCSR, PPPCSRNUMA } v = Version::PPPCSRNUMA;
  OpList core, updates;
  for (int i = 1; i < argc; i++) {
    const std::string s(argv[i]);
    if (starts_with(s, "-threads=")) threads = std::stoi(after(s, "-threads="));
    else if (starts_with(s, "-size=")) size = std::stoi(after(s, "-size="));
    else if (starts_with(s, "-lock_free")) lock_search = false;
    else if (starts_with(s, "-insert")) insert = true;
    else if (starts_with(s, "-delete")) insert = false;
    else if (starts_with(s, "-pppcsrnuma")) v = Version::PPPCSRNUMA;
    else if (starts_with(s, "-pppcsr")) v = Version::PPPCSR;
    else if (starts_with(s, "-ppcsr")) v = Version::PPCSR;
    else if (starts_with(s, "-partitions_per_domain=")) partitions_per_domain = std::stoi(after(s, "-partitions_per_domain="));
    else if (starts_with(s, "-gpus=")) gpus = std::max(1, std::stoi(after(s, "-gpus=")));
    else if (starts_with(s, "-device=")) device = std::stoi(after(s, "-device="));
    else if (starts_with(s, "-verify")) verify = true;
    else if (starts_with(s, "-bulk_core")) bulk = true;
    else if (starts_with(s, "-core_graph=")) {
      int t = 0;
      std::tie(core, t) = read_input(after(s, "-core_graph="), Operation::ADD);
      num_nodes = std::max(num_nodes, t);
    } else if (starts_with(s, "-update_file=")) {
      const std::string fn = after(s, "-update_file=");
      std::cout << fn << std::endl;
      int t = 0;
      std::tie(updates, t) = read_input(fn, insert ? Operation::ADD : Operation::DELETE);  // default op bound HERE
      num_nodes = std::max(num_nodes, t);
      size = (int)std::min((size_t)size, updates.size());
    }
  }
  if (core.empty()) {
    std::cout << "Core graph file not specified" << std::endl;
    return EXIT_FAILURE;
  }
  if (updates.empty()) {
    std::cout << "Updates file not specified" << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << "Core graph size: " << core.size() << std::endl;
  std::vector<int> devices;
  for (int g = 0; g < gpus; g++) devices.push_back(device + g);
  switch (v) {
    case Version::PPCSR: {
      auto pool = std::make_unique<ThreadPool>(threads, lock_search, num_nodes + 1, partitions_per_domain, device);
      execute(threads, size, core, updates, pool.get(), verify, bulk);
      break;
    }
    case Version::PPPCSR: {
      auto pool = std::make_unique<ThreadPoolPPPCSR>(threads, lock_search, num_nodes + 1, partitions_per_domain, false, devices);
      execute(threads, size, core, updates, pool.get(), verify);
      break;
    }
    default: {
      auto pool = std::make_unique<ThreadPoolPPPCSR>(threads, lock_search, num_nodes + 1, partitions_per_domain, true, devices);
      execute(threads, size, core, updates, pool.get(), verify);
    }
  }
  return 0;
}
