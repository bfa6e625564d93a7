/*
 * sbwt_oracle.h -- CPU restatement of jhhung/sBWT's read-mapping path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU
 * baseline ("port").  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call it; the product library (libsbwt_amd.so) never
 * links or loads it.
 *
 * Each function cites the reference file:line (paths under the upstream
 * repository) it restates.  The Aligner_table method bodies are not shipped
 * as source upstream (only as lib/libaligner_table_linux.a); those parts
 * follow the recovered algorithm/format recorded in SURVEY.md 3.3 / 8b.
 *
 * Pinning: regenerated inputs G1/G2/G3 (tests/golden/make_inputs.py) must
 * produce the md5 sums of the reference's own outputs recorded in SURVEY.md
 * Appendix A (SAM for G1/G2/G3, t_table.bwt for G1/G3, NposLen.z for G1).
 */
#ifndef SBWT_ORACLE_H_
#define SBWT_ORACLE_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SBO_KMER 12                       /* searcher.hpp:299 */
#define SBO_NKMER (1ull << (2 * SBO_KMER)) /* 4^12 */

typedef struct sbo_index {
  /* aligner_table.hpp:98-161 (data members), layout as the reference keeps them */
  uint64_t interval, powerV, n, first_location, real_size;
  uint64_t c_function[256];
  uint64_t *c_functions;      /* 4^12+1 */
  uint64_t *occ[4];           /* occ_function['A','C','G','T'], occ_len entries each */
  uint64_t occ_len;
  uint64_t occ_jbwt[4][5][256];
  uint8_t *jbwt_seq;          /* n/4+1 bytes, base 0 of a byte = bits 7..6 */
  uint64_t jbwt_len;
  uint8_t *fbwt;              /* n bytes: SA[i] % interval */
  uint64_t *loc;              /* location_table: n_loc pairs (i, SA[i]) sorted by i */
  uint64_t n_loc;
  /* chr_start_pos / chr_umbiguous_starting_length (aligner_table.hpp:151-161) */
  uint64_t n_chr;
  uint64_t *chr_start;
  char **chr_name;
  uint64_t n_npos;
  uint64_t *npos_key, *npos_val;
  int borrowed;               /* 1: big arrays belong to the caller */
} sbo_index;

typedef struct sbo_hit {
  uint64_t pos0;   /* 0-based position within chromosome (N offset added back) */
  uint32_t chr_id; /* index into chr_name */
  uint32_t is_rc;  /* searcher.hpp:506-515: 1 -> FLAG 0, 0 -> FLAG 16 */
} sbo_hit;

/* status codes for per-read calls */
#define SBO_OK 0
#define SBO_ERR_ILLEGAL_CHAR (-1) /* searcher.hpp:399 throws "illegal char" */
#define SBO_ERR_TOO_SHORT (-2)    /* searcher.hpp:404 substr out_of_range */

const char *sbo_last_error(void);

/* index build: indexer.hpp:137-229 + genome_pre_handler.hpp:175-310 + createAllTable/saveTable */
int sbo_build(const char *fasta_path, const char *prefix, uint64_t interval);
/* build tables in memory from plain forward text (ACGT chars, no N) */
sbo_index *sbo_index_from_text(const char *fwd, uint64_t g, uint64_t interval);
int sbo_index_save(const sbo_index *ix, const char *table_path);

/* searcher.hpp:190-204 load_table */
sbo_index *sbo_index_load(const char *prefix);
/* wrap caller-owned reference-format tables (used by bench.py's cpu_baseline leg) */
sbo_index *sbo_index_wrap(uint64_t interval, uint64_t n, uint64_t first_location,
                          uint64_t real_size, const uint64_t *c_function4 /* C['A'],C['C'],C['G'],C['T'] */,
                          uint64_t *c_functions, uint64_t *occA, uint64_t *occC,
                          uint64_t *occG, uint64_t *occT, uint64_t occ_len,
                          uint8_t *jbwt_seq, uint8_t *fbwt, uint64_t *loc, uint64_t n_loc,
                          uint64_t n_chr, const uint64_t *chr_start,
                          uint64_t n_npos, const uint64_t *npos_key, const uint64_t *npos_val);
void sbo_index_free(sbo_index *ix);

/* Aligner_table::get_occ_using_jbwt(i, c), c in 0..3 */
uint64_t sbo_occ(const sbo_index *ix, uint64_t i, int c);

/* searcher.hpp:386-416: RC + 12-mer init + backward loop. */
int sbo_search(const sbo_index *ix, const char *seq, uint32_t len, uint64_t *lo, uint64_t *hi);

/* searcher.hpp:386-539: whole per-read path; hits in reference emission order.
 * returns NH (>=0) or a negative status. *hits is (re)allocated by the callee. */
int64_t sbo_map_read(const sbo_index *ix, const char *seq, uint32_t len, int limit, int strand,
                     uint64_t *lo, uint64_t *hi, sbo_hit **hits, uint64_t *hits_cap);

/* batch form over ASCII reads (concatenated, offsets[n+1]); nthreads>=1.
 * per read: lo, hi, nh, first (offset into hits_out); hits_out capacity hits_cap.
 * returns number of hits written, or negative status of the first failing read. */
int64_t sbo_map_batch(const sbo_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_reads,
                      int limit, int strand, int nthreads, uint64_t *lo, uint64_t *hi,
                      int32_t *nh, uint64_t *first, sbo_hit *hits_out, uint64_t hits_cap);

/* searcher.hpp:336-373 + job_distributer.hpp:37-90 (-n 1 order) + fastq_reader_impl.hpp:67-111
 * + position2sam/Sam::operator<< : FASTQ -> SAM text. stats = {total, mapped, single, multi, records} */
int sbo_map_fastq(const sbo_index *ix, const char *fastq_path, const char *sam_path, int limit,
                  int strand, uint64_t stats[5]);

#ifdef __cplusplus
}
#endif
#endif
