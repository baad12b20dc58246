// What one output file on this box's tmpfs takes: N threads writing disjoint 8 MB pieces of a 2 GB file through
// pwrite, through a fresh shared mapping (page faults), and through pwrite after fallocate.  g++ -O2 -pthread.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/dev/shm/abm_write_probe.bin";
  const size_t total = 2ull << 30, piece = 8ull << 20;
  std::vector<char> src(piece);
  for (size_t i = 0; i < piece; ++i) src[i] = static_cast<char>('A' + i % 23);
  for (int mode = 0; mode < 3; ++mode)
    for (int nt : {1, 4, 16, 64}) {
      ::unlink(path);
      const int fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { std::perror("open"); return 1; }
      char *map = nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 1) {
        if (::ftruncate(fd, total) != 0) { std::perror("ftruncate"); return 1; }
        map = static_cast<char *>(::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        if (map == MAP_FAILED) { std::perror("mmap"); return 1; }
      }
      if (mode == 2 && ::posix_fallocate(fd, 0, total) != 0) { std::perror("fallocate"); return 1; }
      const double t_prep = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::atomic<size_t> next{0};
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
          for (;;) {
            const size_t at = next.fetch_add(piece);
            if (at >= total) break;
            if (mode == 1) std::memcpy(map + at, src.data(), piece);
            else if (::pwrite(fd, src.data(), piece, static_cast<off_t>(at)) != static_cast<ssize_t>(piece)) { std::perror("pwrite"); std::exit(1); }
          }
        });
      for (auto &t : th) t.join();
      if (map) ::munmap(map, total);
      ::close(fd);
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("%-22s threads %2d: %.2f GB/s (%.3f s, of which preparing %.3f s)\n",
                  mode == 0 ? "pwrite" : (mode == 1 ? "shared mapping" : "pwrite after fallocate"), nt, total / s / 1e9, s, t_prep);
    }
  ::unlink(path);
  return 0;
}
