// What this box's tmpfs takes as an output sink, with the source in DRAM (1 GB of distinct bytes per writer, not an
// L3-resident piece as in r03_write_probe) and writers pinned: one file / several files, source on the writer's NUMA
// node or the other one, pwrite against a shared mapping with and without MADV_HUGEPAGE.  g++ -O2 -pthread.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static std::vector<int> cpulist(const std::string &path) {
  std::vector<int> out;
  std::ifstream f(path);
  std::string s;
  std::getline(f, s);
  size_t i = 0;
  while (i < s.size()) {
    const int a = std::atoi(s.c_str() + i);
    while (i < s.size() && isdigit(s[i])) ++i;
    int b = a;
    if (i < s.size() && s[i] == '-') { ++i; b = std::atoi(s.c_str() + i); while (i < s.size() && isdigit(s[i])) ++i; }
    for (int c = a; c <= b; ++c) out.push_back(c);
    if (i < s.size() && s[i] == ',') ++i;
  }
  return out;
}
static void pin(int cpu) {
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  sched_setaffinity(0, sizeof(set), &set);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
  std::vector<std::vector<int>> nodes;
  for (int n = 0; n < 16; ++n) {
    auto c = cpulist("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist");
    if (c.empty()) break;
    nodes.push_back(c);
  }
  if (nodes.empty()) { std::vector<int> all; for (unsigned c = 0; c < std::thread::hardware_concurrency(); ++c) all.push_back(c); nodes.push_back(all); }
  std::printf("nodes: %zu;", nodes.size());
  for (size_t n = 0; n < nodes.size(); ++n) std::printf(" node%zu: %zu cpus (%d..%d)", n, nodes[n].size(), nodes[n].front(), nodes[n].back());
  std::printf("\n");
  const size_t per = 1ull << 30, piece = 8ull << 20;
  auto cpu_of = [&](int node, int k) { const auto &v = nodes[node % nodes.size()]; return v[(k * 2) % v.size()]; };
  // mode: 0 pwrite one file, 1 pwrite one file per writer, 2 shared mapping, 3 shared mapping + MADV_HUGEPAGE, 4 /dev/null
  struct Case { const char *name; int mode; int writers; int src_node_shift; bool spread; };
  std::vector<Case> cases = {
      {"pwrite, one file, source local", 0, 1, 0, false},
      {"pwrite, one file, source on the other node", 0, 1, 1, false},
      {"pwrite, one file, 2 writers one node", 0, 2, 0, false},
      {"pwrite, one file, 4 writers one node", 0, 4, 0, false},
      {"pwrite, one file, 4 writers two nodes", 0, 4, 0, true},
      {"pwrite, file per writer, 2 writers one node", 1, 2, 0, false},
      {"pwrite, file per writer, 2 writers two nodes", 1, 2, 0, true},
      {"pwrite, file per writer, 4 writers two nodes", 1, 4, 0, true},
      {"pwrite, file per writer, 8 writers two nodes", 1, 8, 0, true},
      {"pwrite, file per writer, 16 writers two nodes", 1, 16, 0, true},
      {"shared mapping, 1 writer", 2, 1, 0, false},
      {"shared mapping, 8 writers two nodes", 2, 8, 0, true},
      {"shared mapping + MADV_HUGEPAGE, 1 writer", 3, 1, 0, false},
      {"shared mapping + MADV_HUGEPAGE, 8 writers", 3, 8, 0, true},
      {"pwrite to /dev/null, 1 writer", 4, 1, 0, false},
  };
  for (const Case &c : cases) {
    const int W = c.writers;
    std::vector<char *> src(W);
    std::vector<std::thread> th;
    // sources: touched by a thread on the node they should live on
    for (int w = 0; w < W; ++w)
      th.emplace_back([&, w] {
        const int node = (c.spread ? w : 0) + c.src_node_shift;
        pin(cpu_of(node, w));
        src[w] = static_cast<char *>(std::aligned_alloc(2u << 20, per));
        for (size_t i = 0; i < per; i += 64) src[w][i] = static_cast<char>(i * 31 + w);
      });
    for (auto &t : th) t.join();
    th.clear();
    std::vector<int> fds(W, -1);
    const bool one_file = c.mode == 0 || c.mode == 2 || c.mode == 3;
    const std::string base = dir + "/abm_sink_probe_";
    char *map = nullptr;
    for (int w = 0; w < (one_file ? 1 : W); ++w) {
      const std::string p = c.mode == 4 ? "/dev/null" : base + std::to_string(w);
      if (c.mode != 4) ::unlink(p.c_str());
      fds[w] = ::open(p.c_str(), O_RDWR | O_CREAT, 0644);
      if (fds[w] < 0) { std::perror("open"); return 1; }
    }
    const double t0 = now();
    if (c.mode == 2 || c.mode == 3) {
      if (::ftruncate(fds[0], per * W) != 0) { std::perror("ftruncate"); return 1; }
      map = static_cast<char *>(::mmap(nullptr, per * W, PROT_READ | PROT_WRITE, MAP_SHARED, fds[0], 0));
      if (map == MAP_FAILED) { std::perror("mmap"); return 1; }
      if (c.mode == 3) ::madvise(map, per * W, MADV_HUGEPAGE);
    }
    for (int w = 0; w < W; ++w)
      th.emplace_back([&, w] {
        pin(cpu_of(c.spread ? w : 0, w));
        const int fd = one_file ? fds[0] : fds[w];
        for (size_t at = 0; at < per; at += piece) {
          const size_t file_at = one_file ? (at / piece * W + w) * piece : at;  // writers interleave pieces in one file
          if (map) std::memcpy(map + file_at, src[w] + at, piece);
          else if (::pwrite(fd, src[w] + at, piece, static_cast<off_t>(c.mode == 4 ? 0 : file_at)) != static_cast<ssize_t>(piece)) { std::perror("pwrite"); std::exit(1); }
        }
      });
    for (auto &t : th) t.join();
    const double s = now() - t0;
    if (map) ::munmap(map, per * W);
    for (int w = 0; w < W; ++w) if (fds[w] >= 0) ::close(fds[w]);
    for (int w = 0; w < W; ++w) if (c.mode != 4) ::unlink((base + std::to_string(w)).c_str());
    for (int w = 0; w < W; ++w) std::free(src[w]);
    std::printf("%-52s %6.2f GB/s  (%d GB in %.3f s)\n", c.name, per * W / s / 1e9, W, s);
    std::fflush(stdout);
  }
  return 0;
}
