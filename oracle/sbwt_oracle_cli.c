/* Command-line front end of the CPU oracle (test infrastructure only).
 * Mirrors `sbwt build -i -p [-s]` / `sbwt map -i -p -o [-m] [-s]` (main.cpp:50-254) with -n fixed at 1. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sbwt_oracle.h"

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: sbwt_oracle build|map ...\n"); return 1; }
  const char *in = NULL, *prefix = NULL, *out = "/dev/stdout";
  int limit = 1000, strand = 2, interval = 64, is_build = !strcmp(argv[1], "build");
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "-i")) in = argv[i + 1];
    else if (!strcmp(argv[i], "-p")) prefix = argv[i + 1];
    else if (!strcmp(argv[i], "-o")) out = argv[i + 1];
    else if (!strcmp(argv[i], "-m")) limit = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "-s")) { if (is_build) interval = atoi(argv[i + 1]); else strand = atoi(argv[i + 1]); }
    else if (!strcmp(argv[i], "-n") || !strcmp(argv[i], "-l")) {}
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 1; }
  }
  if (!in || !prefix) { fprintf(stderr, "-i and -p are required\n"); return 1; }
  if (is_build) {
    if (sbo_build(in, prefix, (uint64_t)interval)) { fprintf(stderr, "Error: %s\n", sbo_last_error()); return 1; }
    return 0;
  }
  sbo_index *ix = sbo_index_load(prefix);
  if (!ix) { fprintf(stderr, "Error: %s\n", sbo_last_error()); return 2; }
  uint64_t st[5];
  int rc = sbo_map_fastq(ix, in, out, limit, strand, st);
  if (rc) { fprintf(stderr, "Error: %s\n", sbo_last_error()); return 134; }
  fprintf(stderr, "Total reads: %llu\tSingle mapped reads: %llu\tMultiple mapped reads: %llu\n",
          (unsigned long long)st[0], (unsigned long long)st[2], (unsigned long long)st[3]);
  fprintf(stderr, "Total sam results number: %llu\n", (unsigned long long)st[4]);
  sbo_index_free(ix);
  return 0;
}
