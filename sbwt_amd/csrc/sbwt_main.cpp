// sbwt_main.cpp -- `sbwt build` / `sbwt map` command line with the option set of upstream main.cpp:50-254.
// Option parsing only; all work happens behind the C ABI of include/sbwt_amd.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/sbwt_amd.h"

static bool exists(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0;
}
static bool index_intact(const std::string& prefix) {  // main.cpp:34-47
  return exists(prefix + "t_table.bwt") && exists(prefix + "NposLen.z") && exists(prefix + "chrStart") &&
         exists(prefix + "chrLen");
}
static const char* kUsage =
    "\n+----+\n|sBWT|  (MI355X build)\n+----+\n\nUsage:\n\n 1. building index of the genome\n>\tsbwt build -i genome.fa -p prefix "
    "[-l 256] [-s 64] [-f]\n\n 2. mapping fastq to the index\n>\tsbwt map -i reads.fq -p prefix [-o stdout] [-n 1] [-m 1000] "
    "[-s 2] [-g 1]\n\t(-g: GPUs to map on, 0 = all; the index is loaded on each, the reads are shared out)\n\n";

struct Opt {
  const char *s, *l;
};
static bool is_opt(const char* a, const char* s, const char* l) { return !strcmp(a, s) || !strcmp(a, l); }

static int cmd_build(int argc, char** argv) {
  std::string fasta, prefix;
  long sort_len = 256, interval = 64;
  bool force = false;
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
    if (is_opt(a, "-f", "--force")) { force = true; continue; }
    if (is_opt(a, "-h", "--help")) { fputs(kUsage, stderr); return 1; }
    if (!v) { fprintf(stderr, "Error: option %s needs a value\n%s", a, kUsage); return 1; }
    if (is_opt(a, "-i", "--input")) fasta = v;
    else if (is_opt(a, "-p", "--prefix")) prefix = v;
    else if (is_opt(a, "-l", "--sort_len")) sort_len = atol(v);
    else if (is_opt(a, "-s", "--interval")) interval = atol(v);
    else { fprintf(stderr, "Error: unrecognised option %s\n%s", a, kUsage); return 1; }
    ++i;
  }
  if (fasta.empty() || prefix.empty()) { fprintf(stderr, "Error: the options --input and --prefix are required\n%s", kUsage); return 1; }
  if (!exists(fasta)) {
    fprintf(stderr, "Error: Input fasta file %s does not exist! Please double check. Existing...\n", fasta.c_str());
    return 1;
  }
  if (prefix.back() != '.') prefix += '.';
  if (index_intact(prefix) && !force) {
    fprintf(stderr, "Error: index files already exist. If you want to overwrite them, please run it again with option -f.\nExisting...\n");
    return 2;
  }
  fprintf(stderr, "Parameters fasta: %s\nParameters prefix: %s\nParameters sort_len: %ld\nParameters interval: %ld\n",
          fasta.c_str(), prefix.c_str(), sort_len, interval);
  int rc = sbwt_build_index(fasta.c_str(), prefix.c_str(), (uint64_t)sort_len, (uint64_t)interval, 0);
  if (rc != SBWT_OK) { fprintf(stderr, "Error: %s\n", sbwt_last_error()); return 1; }
  return 0;
}

static int cmd_map(int argc, char** argv) {
  std::string fastq, prefix, sam = "stdout";
  int nthread = 1, limit = 1000, strand = 2, gpus = 1;
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    const char* v = i + 1 < argc ? argv[i + 1] : nullptr;
    if (is_opt(a, "-h", "--help")) { fputs(kUsage, stderr); return 1; }
    if (!v) { fprintf(stderr, "Error: option %s needs a value\n%s", a, kUsage); return 1; }
    if (is_opt(a, "-i", "--input")) fastq = v;
    else if (is_opt(a, "-p", "--index")) prefix = v;
    else if (is_opt(a, "-o", "--output")) sam = v;
    else if (is_opt(a, "-n", "--thread")) nthread = atoi(v);
    else if (is_opt(a, "-m", "--multiple")) limit = atoi(v);
    else if (is_opt(a, "-s", "--strand")) strand = atoi(v);
    else if (is_opt(a, "-g", "--gpus")) gpus = atoi(v);
    else { fprintf(stderr, "Error: unrecognised option %s\n%s", a, kUsage); return 1; }
    ++i;
  }
  if (fastq.empty() || prefix.empty()) { fprintf(stderr, "Error: the options --input and --index are required\n%s", kUsage); return 1; }
  if (prefix.back() != '.') prefix += '.';
  if (!index_intact(prefix)) {
    fprintf(stderr, "Error: index files appear to be damaged. Please rebuild them.\nExiting...\n");
    return 2;
  }
  if (!exists(fastq)) {
    fprintf(stderr, "Error: Input fastq file %s does not exist! Please double check.\nExiting...\n", fastq.c_str());
    return 1;
  }
  const unsigned cores = std::thread::hardware_concurrency();  // main.cpp:215-220
  if (cores != 0 && nthread > (int)cores) {
    fprintf(stderr, "Warning: the number of threads set (%d) is larger than the number of cores available (%u) in this machine.\nSo reset -n=%u\n",
            nthread, cores, cores);
    nthread = (int)cores;
  }
  fprintf(stderr, "Parameters fastq: %s\nParameters prefix: %s\nParameters output: %s\nParameters thread: %d\nParameters multiple: %d\nParameters strand: %d\n",
          fastq.c_str(), prefix.c_str(), sam.c_str(), nthread, limit, strand);
  int n_dev = 0;
  if (sbwt_device_count(&n_dev) != SBWT_OK) { fprintf(stderr, "Error: %s\n", sbwt_last_error()); return 2; }
  if (n_dev <= 0) { fprintf(stderr, "Error: no GPU found\n"); return 2; }
  if (gpus > n_dev) fprintf(stderr, "Warning: -g %d asks for more GPUs than this machine shows (%d).\nSo reset -g=%d\n", gpus, n_dev, n_dev);
  if (gpus <= 0 || gpus > n_dev) gpus = n_dev;
  // one replica of the index per device, loaded side by side
  std::vector<sbwt_index*> ixs((size_t)gpus, nullptr);
  std::vector<int> rcs((size_t)gpus, SBWT_OK);
  std::vector<std::string> errs((size_t)gpus);
  {
    std::vector<std::thread> th;
    for (int d = 0; d < gpus; ++d)
      th.emplace_back([&, d] {
        rcs[(size_t)d] = sbwt_index_load(prefix.c_str(), d, &ixs[(size_t)d]);
        if (rcs[(size_t)d] != SBWT_OK) errs[(size_t)d] = sbwt_last_error();
      });
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < gpus; ++d)
    if (rcs[(size_t)d] != SBWT_OK) {
      fprintf(stderr, "Error: %s\n", errs[(size_t)d].c_str());
      for (auto* x : ixs) sbwt_index_free(x);
      return 2;
    }
  uint64_t st[5] = {0, 0, 0, 0, 0};
  int rc = sbwt_map_fastq_multi(ixs.data(), gpus, fastq.c_str(), sam.c_str(), limit, strand, nthread, st);
  for (auto* x : ixs) sbwt_index_free(x);
  if (rc != SBWT_OK) {
    fprintf(stderr, "Error: %s\n", sbwt_last_error());
    // upstream dies in std::terminate on such reads (SIGABRT); keep the same exit status
    return (rc == SBWT_ERR_ILLEGAL_CHAR || rc == SBWT_ERR_TOO_SHORT) ? 134 : 1;
  }
  fprintf(stderr, "Total reads: %llu\tSingle mapped reads: %llu\tMultiple mapped reads: %llu\t Mappability: %g%%\n",
          (unsigned long long)st[0], (unsigned long long)st[2], (unsigned long long)st[3],
          st[0] ? (double)st[1] / (double)st[0] * 100.0 : 0.0);
  fprintf(stderr, "Total sam results number: %llu\n", (unsigned long long)st[4]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fputs(kUsage, stderr); return 1; }
  if (!strcmp(argv[1], "build")) return cmd_build(argc - 1, argv + 1);
  if (!strcmp(argv[1], "map")) return cmd_map(argc - 1, argv + 1);
  fprintf(stderr, "Error: unrecognized option %s\n%s", argv[1], kUsage);
  return 1;
}
