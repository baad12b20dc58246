// Test harness (CPU only): runs the product CLI's FASTQ splitter and parser -- the code of
// abismal_amd/csrc/abm_cli.cpp itself, included here -- over a file and prints what a batch would hand to
// the mapper: one "name<TAB>trimmed read" line per record, "#batch" between batches.
#define main abm_cli_main_unused
#include "../../abismal_amd/csrc/abm_cli.cpp"
#undef main

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: harness reads.fq[.gz] batch_records\n"); return 2; }
  try {
    RawSplitter s(argv[1]);
    const size_t want = static_cast<size_t>(std::atol(argv[2]));
    for (;;) {
      RawBuf raw;
      uint64_t first = 0;
      const uint64_t lines = s.next(want, raw, first);
      if (lines == 0) break;
      std::vector<NameRef> names;
      RawBuf blob;
      std::vector<uint64_t> off;
      parse_raw(raw, first, argv[1], names, blob, off);
      std::printf("#batch first_line=%llu records=%zu\n", static_cast<unsigned long long>(first), names.size());
      for (size_t i = 0; i < names.size(); ++i)
        std::printf("%.*s\t%.*s\n", static_cast<int>(names[i].n), names[i].p, static_cast<int>(off[i + 1] - off[i]), blob.data() + off[i]);
      if (s.exhausted()) break;
    }
  }
  catch (const std::exception &e) { std::printf("#error %s\n", e.what()); return 1; }
  return 0;
}
