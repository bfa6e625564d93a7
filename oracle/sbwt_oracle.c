/*
 * sbwt_oracle.c -- CPU restatement of jhhung/sBWT's read-mapping path (and of
 * the index build needed to feed it).  TEST INFRASTRUCTURE ONLY: see the
 * header of sbwt_oracle.h.  Nothing under sbwt_amd/ links this file.
 *
 * Reference citations are paths under the upstream repository.
 */
#define _GNU_SOURCE
#include "sbwt_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char *sbo_last_error(void) { return g_err; }
#define FAIL(...)                                  \
  do {                                             \
    snprintf(g_err, sizeof g_err, __VA_ARGS__);    \
    return 0;                                      \
  } while (0)

static const char ACGT[4] = {'A', 'C', 'G', 'T'};
static inline int base_code(char c) {
  switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
  }
}

/* ------------------------------------------------------------------------ */
/* rank: Aligner_table::get_occ_using_jbwt (aligner_table.hpp:276; body only in
 * lib/libaligner_table_linux.a -- algorithm per SURVEY.md 3.3)              */
/* ------------------------------------------------------------------------ */
uint64_t sbo_occ(const sbo_index *ix, uint64_t i, int c) {
  uint64_t blk = i >> ix->powerV;
  uint64_t r = ix->occ[c][blk];
  if (i > ix->first_location && c == 0) r -= 1; /* '$' is stored as 'A' */
  uint64_t start = blk << ix->powerV;
  if (i == start) return r;
  const uint8_t *s = ix->jbwt_seq;
  for (uint64_t b = start >> 2; b < (i >> 2); ++b) r += ix->occ_jbwt[c][4][s[b]];
  r += ix->occ_jbwt[c][i & 3][s[i >> 2]];
  return r;
}

/* searcher.hpp:102-105 exec_exact_match */
static inline void exec_exact_match(const sbo_index *ix, uint64_t *lo, uint64_t *hi, int c) {
  uint64_t C = ix->c_function[(unsigned char)ACGT[c]];
  *lo = C + sbo_occ(ix, *lo, c);
  *hi = C + sbo_occ(ix, *hi, c);
}

/* searcher.hpp:386-410: reverse-complement the read, 12-mer init, backward loop */
static int search_rc(const sbo_index *ix, const char *seq, uint32_t len, uint8_t *q /* len codes of RC */,
                     uint64_t *lo, uint64_t *hi) {
  /* searcher.hpp:389-401 */
  for (uint32_t k = 0; k < len; ++k) {
    int b = base_code(seq[len - 1 - k]);
    if (b < 0) return SBO_ERR_ILLEGAL_CHAR;
    q[k] = (uint8_t)(3 - b);
  }
  /* searcher.hpp:404 substr(size-12, 12) throws for size < 12 */
  if (len < SBO_KMER) return SBO_ERR_TOO_SHORT;
  /* searcher.hpp:83-96 */
  uint64_t code = 0;
  for (uint32_t k = 0; k < SBO_KMER; ++k) code = code * 4 + q[len - SBO_KMER + k];
  *lo = ix->c_functions[code];
  *hi = ix->c_functions[code + 1];
  /* searcher.hpp:406-410 */
  for (int i = (int)len - 1 - SBO_KMER; i >= 0 && *lo < *hi; --i) exec_exact_match(ix, lo, hi, q[i]);
  return SBO_OK;
}

int sbo_search(const sbo_index *ix, const char *seq, uint32_t len, uint64_t *lo, uint64_t *hi) {
  uint8_t stackq[512];
  uint8_t *q = len <= sizeof stackq ? stackq : (uint8_t *)malloc(len);
  int st = search_rc(ix, seq, len, q, lo, hi);
  if (q != stackq) free(q);
  return st;
}

/* ------------------------------------------------------------------------ */
/* locate + filters                                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
  const sbo_index *ix;
  uint64_t qlen;
  int strand;
  int maybe_result_number; /* searcher.hpp:412: int */
  sbo_hit *res;
  uint64_t nres, cap;
} vars_t;

/* std::map::upper_bound(x) followed by advance(-1) over a sorted key array */
static inline uint64_t ub_minus1(const uint64_t *keys, uint64_t n, uint64_t x) {
  uint64_t l = 0, h = n;
  while (l < h) {
    uint64_t m = (l + h) >> 1;
    if (keys[m] <= x) l = m + 1; else h = m;
  }
  /* l == index of first key > x; the reference steps back one (begin()-1 would be UB upstream;
     key 0 is always present so l >= 1 for every x) */
  return l - 1;
}

/* searcher.hpp:504-539 */
static void push_position(vars_t *V, uint64_t position) {
  const sbo_index *ix = V->ix;
  uint32_t isRC;
  if (position >= ix->real_size) {
    isRC = 1;
    position = ix->real_size * 2 - position - V->qlen;
  } else {
    isRC = 0;
  }
  if ((V->strand == 0 && !isRC) || (V->strand == 1 && isRC)) return;
  uint64_t c1 = ub_minus1(ix->chr_start, ix->n_chr, position);
  uint64_t c3 = ub_minus1(ix->chr_start, ix->n_chr, position + V->qlen - 1);
  /* the reference compares chromosome NAMES (chr != chr3) */
  if (c1 != c3 && strcmp(ix->chr_name[c1], ix->chr_name[c3]) != 0) return;
  uint64_t n1 = ub_minus1(ix->npos_key, ix->n_npos, position);
  uint64_t n3 = ub_minus1(ix->npos_key, ix->n_npos, position + V->qlen - 1);
  if (n1 != n3) return;
  position = position - ix->chr_start[c1] + ix->npos_val[n1];
  if (V->nres == V->cap) {
    V->cap = V->cap ? V->cap * 2 : 16;
    V->res = (sbo_hit *)realloc(V->res, V->cap * sizeof(sbo_hit));
  }
  V->res[V->nres].pos0 = position;
  V->res[V->nres].chr_id = (uint32_t)c1;
  V->res[V->nres].is_rc = isRC;
  V->nres++;
}

/* std::lower_bound over location_table by .first */
static inline uint64_t loc_lower_bound(const sbo_index *ix, uint64_t i) {
  uint64_t l = 0, h = ix->n_loc;
  while (l < h) {
    uint64_t m = (l + h) >> 1;
    if (ix->loc[2 * m] < i) l = m + 1; else h = m;
  }
  return l;
}

/* searcher.hpp:460-503 */
static void find_possible(vars_t *V, int len, int c, uint64_t pos_f, uint64_t pos_e) {
  const sbo_index *ix = V->ix;
  if ((uint64_t)len >= ix->interval) return;
  if (V->nres == (uint64_t)(size_t)V->maybe_result_number) return;
  uint64_t C = ix->c_function[(unsigned char)ACGT[c]];
  uint64_t nf = C + sbo_occ(ix, pos_f, c);
  uint64_t ne = C + sbo_occ(ix, pos_e, c);
  if (nf >= ne) return;
  for (uint64_t i = nf; i < ne; ++i) {
    if (ix->fbwt[i] != 0) continue;
    uint64_t p = loc_lower_bound(ix, i);
    if (p < ix->n_loc && ix->loc[2 * p] == i) {
      push_position(V, (uint64_t)len + ix->loc[2 * p + 1]);
      if (ne - nf == 1) return;
    }
  }
  for (int cn = 0; cn < 4; ++cn) find_possible(V, len + 1, cn, nf, ne);
}

/* searcher.hpp:386-458 minus the SAM text */
int64_t sbo_map_read(const sbo_index *ix, const char *seq, uint32_t len, int limit, int strand,
                     uint64_t *lo_out, uint64_t *hi_out, sbo_hit **hits, uint64_t *hits_cap) {
  uint8_t stackq[512];
  uint8_t *q = len <= sizeof stackq ? stackq : (uint8_t *)malloc(len);
  uint64_t lo = 0, hi = 0;
  int st = search_rc(ix, seq, len, q, &lo, &hi);
  if (q != stackq) free(q);
  if (st != SBO_OK) return st;
  if (lo_out) *lo_out = lo;
  if (hi_out) *hi_out = hi;
  /* searcher.hpp:412-416: unsigned compare against (uint64)limit */
  if (hi - lo > (uint64_t)(int64_t)limit || hi - lo <= 0) return 0;
  vars_t V = {ix, len, strand, (int)(hi - lo), *hits, 0, *hits_cap};
  /* searcher.hpp:424-443 */
  for (uint64_t i = lo; i < hi; ++i) {
    if (ix->fbwt[i] != 0) continue;
    uint64_t p = loc_lower_bound(ix, i);
    if (p < ix->n_loc && ix->loc[2 * p] == i) {
      push_position(&V, ix->loc[2 * p + 1]);
      if (hi - lo == 1) break;
    }
  }
  /* searcher.hpp:446-449 */
  for (int cn = 0; cn < 4; ++cn) find_possible(&V, 1, cn, lo, hi);
  *hits = V.res;
  *hits_cap = V.cap;
  return (int64_t)V.nres; /* searcher.hpp:453-456,543 */
}

/* ------------------------------------------------------------------------ */
/* batch form (threads over contiguous read ranges; results in input order)  */
/* ------------------------------------------------------------------------ */
typedef struct {
  const sbo_index *ix;
  const char *seqs;
  const uint64_t *offsets;
  uint64_t r0, r1;
  int limit, strand;
  uint64_t *lo, *hi;
  int32_t *nh;
  sbo_hit *hits; /* private growing array */
  uint64_t nhits, cap;
  uint64_t *cnt; /* per-read hit count */
  int status;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *J = (batch_job *)arg;
  sbo_hit *tmp = NULL;
  uint64_t tmpcap = 0;
  for (uint64_t r = J->r0; r < J->r1; ++r) {
    uint64_t lo = 0, hi = 0;
    uint32_t len = (uint32_t)(J->offsets[r + 1] - J->offsets[r]);
    int64_t nh = sbo_map_read(J->ix, J->seqs + J->offsets[r], len, J->limit, J->strand, &lo, &hi, &tmp, &tmpcap);
    if (nh < 0) {
      if (!J->status) J->status = (int)nh;
      nh = 0;
    }
    J->lo[r] = lo;
    J->hi[r] = hi;
    J->nh[r] = (int32_t)nh;
    if (J->nhits + (uint64_t)nh > J->cap) {
      J->cap = (J->nhits + (uint64_t)nh) * 2 + 1024;
      J->hits = (sbo_hit *)realloc(J->hits, J->cap * sizeof(sbo_hit));
    }
    memcpy(J->hits + J->nhits, tmp, (size_t)nh * sizeof(sbo_hit));
    J->nhits += (uint64_t)nh;
  }
  free(tmp);
  return NULL;
}

int64_t sbo_map_batch(const sbo_index *ix, const char *seqs, const uint64_t *offsets, uint64_t n_reads,
                      int limit, int strand, int nthreads, uint64_t *lo, uint64_t *hi, int32_t *nh,
                      uint64_t *first, sbo_hit *hits_out, uint64_t hits_cap) {
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > n_reads) nthreads = n_reads ? (int)n_reads : 1;
  batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; ++t) {
    batch_job *J = &jobs[t];
    J->ix = ix; J->seqs = seqs; J->offsets = offsets;
    J->r0 = n_reads * (uint64_t)t / (uint64_t)nthreads;
    J->r1 = n_reads * (uint64_t)(t + 1) / (uint64_t)nthreads;
    J->limit = limit; J->strand = strand; J->lo = lo; J->hi = hi; J->nh = nh;
    if (nthreads == 1) batch_worker(J);
    else pthread_create(&th[t], NULL, batch_worker, J);
  }
  int64_t total = 0;
  int status = 0;
  for (int t = 0; t < nthreads; ++t) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].status && !status) status = jobs[t].status;
  }
  for (int t = 0; t < nthreads; ++t) {
    batch_job *J = &jobs[t];
    uint64_t off = (uint64_t)total;
    for (uint64_t r = J->r0; r < J->r1; ++r) {
      first[r] = off;
      off += (uint64_t)nh[r];
    }
    if ((uint64_t)total + J->nhits <= hits_cap && hits_out)
      memcpy(hits_out + total, J->hits, J->nhits * sizeof(sbo_hit));
    else if (J->nhits && !status) status = -3;
    total += (int64_t)J->nhits;
    free(J->hits);
  }
  free(jobs);
  free(th);
  if (status) {
    snprintf(g_err, sizeof g_err, "sbo_map_batch: status %d", status);
    return status;
  }
  return total;
}

/* ------------------------------------------------------------------------ */
/* FASTQ -> SAM text                                                         */
/* ------------------------------------------------------------------------ */
typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const char *s, size_t n) {
  if (b->n + n + 1 > b->cap) {
    b->cap = (b->n + n + 1) * 2 + 4096;
    b->p = (char *)realloc(b->p, b->cap);
  }
  memcpy(b->p + b->n, s, n);
  b->n += n;
}
static void sb_u64(sbuf *b, uint64_t v) {
  char t[32];
  int k = snprintf(t, sizeof t, "%llu", (unsigned long long)v);
  sb_put(b, t, (size_t)k);
}

/* std::getline semantics on a memory image: returns 0 when the stream would have failbit
 * (nothing extracted at EOF); *hit_eof set when the line ended at EOF without '\n'. */
typedef struct { const char *d; size_t n, pos; int eof, fail; } mstream;
static void ms_getline(mstream *s, const char **line, size_t *len) {
  *line = s->d + s->pos;
  *len = 0;
  if (s->fail || s->eof) { s->fail = 1; return; }
  if (s->pos >= s->n) { s->eof = 1; s->fail = 1; return; }
  const char *nl = (const char *)memchr(s->d + s->pos, '\n', s->n - s->pos);
  if (nl) {
    *len = (size_t)(nl - (s->d + s->pos));
    s->pos += *len + 1;
  } else {
    *len = s->n - s->pos;
    s->pos = s->n;
    s->eof = 1; /* reached EOF while extracting: eofbit, not failbit */
  }
}
/* boost::trim: std::isspace in the classic locale on both ends */
static void trim(const char **s, size_t *n) {
  while (*n && isspace((unsigned char)(*s)[0])) { ++*s; --*n; }
  while (*n && isspace((unsigned char)(*s)[*n - 1])) --*n;
}

static char *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *d = (char *)malloc((size_t)sz + 1);
  if (sz && fread(d, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(d); return NULL; }
  d[sz] = 0;
  fclose(f);
  *n = (size_t)sz;
  return d;
}

int sbo_map_fastq(const sbo_index *ix, const char *fastq_path, const char *sam_path, int limit, int strand,
                  uint64_t stats[5]) {
  size_t fn = 0;
  char *fd = slurp(fastq_path, &fn);
  if (!fd) { snprintf(g_err, sizeof g_err, "cannot read %s", fastq_path); return -10; }
  FILE *out = fopen(sam_path, "wb");
  if (!out) { free(fd); snprintf(g_err, sizeof g_err, "cannot write %s", sam_path); return -11; }
  mstream ms = {fd, fn, 0, 0, 0};
  sbuf sb = {0};
  sbo_hit *hits = NULL;
  uint64_t hcap = 0;
  uint64_t st[5] = {0, 0, 0, 0, 0};
  char *rcseq = NULL, *rq = NULL;
  size_t rcap = 0;
  int rc = 0;
  for (;;) {
    /* fastq_reader_impl.hpp:81-98 */
    if (ms.pos >= ms.n) break; /* peek() at EOF sets eofbit -> eof_flag */
    const char *l[4];
    size_t ln[4];
    for (int k = 0; k < 4; ++k) ms_getline(&ms, &l[k], &ln[k]);
    if (ms.eof || ms.fail) break; /* !good(): the record is dropped */
    for (int k = 0; k < 4; ++k) trim(&l[k], &ln[k]);
    uint32_t L = (uint32_t)ln[1];
    uint64_t lo, hi;
    int64_t nh = sbo_map_read(ix, l[1], L, limit, strand, &lo, &hi, &hits, &hcap);
    if (nh < 0) {
      snprintf(g_err, sizeof g_err, "read %llu: %s (the reference aborts here)", (unsigned long long)st[0],
               nh == SBO_ERR_ILLEGAL_CHAR ? "illegal char" : "shorter than 12");
      rc = (int)nh;
      break;
    }
    st[0]++;
    st[4] += (uint64_t)nh;
    if (nh == 1) { st[1]++; st[2]++; } else if (nh > 1) { st[1]++; st[3]++; }
    if (nh > 0) {
      if (rcap < (size_t)L + ln[3] + 2) {
        rcap = ((size_t)L + ln[3] + 2) * 2;
        rcseq = (char *)realloc(rcseq, rcap);
        rq = (char *)realloc(rq, rcap);
      }
      for (uint32_t k = 0; k < L; ++k) rcseq[k] = ACGT[3 - base_code(l[1][L - 1 - k])];
      for (size_t k = 0; k < ln[3]; ++k) rq[k] = l[3][ln[3] - 1 - k];
    }
    /* searcher.hpp:540-598 + sam.hpp:131-140 + tuple_utility.hpp:54-64 */
    for (int64_t h = 0; h < nh; ++h) {
      const sbo_hit *H = &hits[h];
      sb_put(&sb, l[0] + (ln[0] ? 1 : 0), ln[0] ? ln[0] - 1 : 0); /* fastq.hpp:134-137 substr(1) */
      sb_put(&sb, "\t", 1);
      sb_put(&sb, H->is_rc ? "0" : "16", H->is_rc ? 1 : 2);
      sb_put(&sb, "\t", 1);
      sb_put(&sb, ix->chr_name[H->chr_id], strlen(ix->chr_name[H->chr_id]));
      sb_put(&sb, "\t", 1);
      sb_u64(&sb, H->pos0 + 1);
      sb_put(&sb, "\t255\t", 5);
      sb_u64(&sb, L);
      sb_put(&sb, "M\t*\t0\t0\t", 8);
      if (H->is_rc) sb_put(&sb, l[1], L); else sb_put(&sb, rcseq, L);
      sb_put(&sb, "\t", 1);
      if (H->is_rc) sb_put(&sb, l[3], ln[3]); else sb_put(&sb, rq, ln[3]);
      sb_put(&sb, "\t", 1);
      sb_u64(&sb, (uint64_t)(int)nh);
      sb_put(&sb, "\t\n", 2);
    }
    if (sb.n > (1u << 20)) { fwrite(sb.p, 1, sb.n, out); sb.n = 0; }
  }
  if (sb.n) fwrite(sb.p, 1, sb.n, out);
  fclose(out);
  free(sb.p); free(hits); free(rcseq); free(rq); free(fd);
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* archive I/O (Boost binary_oarchive v10 layout)              */
/* ------------------------------------------------------------------------ */
static const unsigned char ARCH_HDR[40] = {22, 0, 0, 0, 0, 0, 0, 0, 's', 'e', 'r', 'i', 'a', 'l', 'i', 'z', 'a', 't',
                                           'i', 'o', 'n', ':', ':', 'a', 'r', 'c', 'h', 'i', 'v', 'e', 10, 0, 4, 8,
                                           4, 8, 1, 0, 0, 0};

static void w64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }
static void w32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void wz(FILE *f, int n) { static const char z[8] = {0}; fwrite(z, 1, (size_t)n, f); }

static void fill_occ_jbwt(sbo_index *ix) {
  /* using_jbwt (aligner_table.hpp:205, binary-only): [c][k][byte] = # of c among the first k bases
   * of the packed byte (k = 0..3), [c][4][byte] = whole byte.  jbit.hpp:193-206 packing. */
  for (int c = 0; c < 4; ++c)
    for (int b = 0; b < 256; ++b) {
      int cnt = 0;
      for (int k = 0; k < 4; ++k) {
        ix->occ_jbwt[c][k][b] = (uint64_t)cnt;
        if (((b >> (6 - 2 * k)) & 3) == c) ++cnt;
      }
      ix->occ_jbwt[c][4][b] = (uint64_t)cnt;
    }
}

int sbo_index_save(const sbo_index *ix, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
  setvbuf(f, NULL, _IOFBF, 1 << 22);
  fwrite(ARCH_HDR, 1, 40, f);
  w64(f, ix->interval);
  w64(f, 256); fwrite(ix->c_function, 8, 256, f);
  w64(f, SBO_NKMER + 1); fwrite(ix->c_functions, 8, SBO_NKMER + 1, f);
  wz(f, 5); w64(f, 256); w32(f, 0);
  for (int v = 0; v < 256; ++v) {
    int c = base_code((char)v);
    if (c < 0) { w64(f, 0); continue; }
    w64(f, ix->occ_len); fwrite(ix->occ[c], 8, ix->occ_len, f);
  }
  wz(f, 5); w64(f, ix->n_loc); fwrite(ix->loc, 16, ix->n_loc, f);
  w64(f, ix->n); fwrite(ix->fbwt, 1, ix->n, f);
  w64(f, ix->first_location);
  wz(f, 5); w64(f, 256); w32(f, 0);
  static const uint64_t zero256[256] = {0};
  for (int v = 0; v < 256; ++v) {
    int c = base_code((char)v);
    w64(f, 5); w32(f, 0);
    for (int k = 0; k < 5; ++k) {
      w64(f, 256);
      fwrite(c < 0 ? zero256 : ix->occ_jbwt[c][k], 8, 256, f);
    }
  }
  w64(f, ix->jbwt_len); fwrite(ix->jbwt_seq, 1, ix->jbwt_len, f);
  w64(f, 1024);
  for (int b = 0; b < 256; ++b)
    for (int k = 0; k < 4; ++k) fputc(ACGT[(b >> (6 - 2 * k)) & 3], f);
  int bad = ferror(f);
  if (fclose(f) || bad) { snprintf(g_err, sizeof g_err, "write error on %s", path); return -1; }
  return 0;
}

typedef struct { FILE *f; int bad; } rd;
static uint64_t r64(rd *r) { uint64_t v = 0; if (fread(&v, 8, 1, r->f) != 1) r->bad = 1; return v; }
static uint32_t r32(rd *r) { uint32_t v = 0; if (fread(&v, 4, 1, r->f) != 1) r->bad = 1; return v; }
static void rskip(rd *r, long n) { if (fseek(r->f, n, SEEK_CUR)) r->bad = 1; }
static void *rarr(rd *r, uint64_t count, size_t esz) {
  void *p = malloc(count * esz + 8);
  if (!p) { r->bad = 1; return NULL; }
  if (count && fread(p, esz, count, r->f) != count) r->bad = 1;
  return p;
}

static int read_table(sbo_index *ix, const char *path) {
  rd R = {fopen(path, "rb"), 0};
  if (!R.f) { snprintf(g_err, sizeof g_err, "cannot open %s", path); return -1; }
  setvbuf(R.f, NULL, _IOFBF, 1 << 22);
  unsigned char hdr[40];
  if (fread(hdr, 1, 40, R.f) != 40 || memcmp(hdr, ARCH_HDR, 40)) {
    fclose(R.f);
    snprintf(g_err, sizeof g_err, "%s: not a boost binary archive v10", path);
    return -1;
  }
  ix->interval = r64(&R);
  if (r64(&R) != 256) R.bad = 1;
  if (fread(ix->c_function, 8, 256, R.f) != 256) R.bad = 1;
  if (r64(&R) != SBO_NKMER + 1) R.bad = 1;
  if (!R.bad) ix->c_functions = (uint64_t *)rarr(&R, SBO_NKMER + 1, 8);
  rskip(&R, 5);
  if (r64(&R) != 256) R.bad = 1;
  r32(&R);
  for (int v = 0; v < 256 && !R.bad; ++v) {
    uint64_t len = r64(&R);
    int c = base_code((char)v);
    if (c >= 0) { ix->occ_len = len; ix->occ[c] = (uint64_t *)rarr(&R, len, 8); }
    else rskip(&R, (long)(len * 8));
  }
  rskip(&R, 5);
  ix->n_loc = r64(&R);
  if (!R.bad) ix->loc = (uint64_t *)rarr(&R, ix->n_loc * 2, 8);
  ix->n = r64(&R);
  if (!R.bad) ix->fbwt = (uint8_t *)rarr(&R, ix->n, 1);
  ix->first_location = r64(&R);
  rskip(&R, 5);
  if (r64(&R) != 256) R.bad = 1;
  r32(&R);
  for (int v = 0; v < 256 && !R.bad; ++v) {
    if (r64(&R) != 5) R.bad = 1;
    r32(&R);
    int c = base_code((char)v);
    for (int k = 0; k < 5 && !R.bad; ++k) {
      if (r64(&R) != 256) R.bad = 1;
      if (c >= 0) { if (fread(ix->occ_jbwt[c][k], 8, 256, R.f) != 256) R.bad = 1; }
      else rskip(&R, 256 * 8);
    }
  }
  ix->jbwt_len = r64(&R);
  if (!R.bad) ix->jbwt_seq = (uint8_t *)rarr(&R, ix->jbwt_len, 1);
  if (r64(&R) != 1024) R.bad = 1;
  rskip(&R, 1024);
  int bad = R.bad;
  fclose(R.f);
  if (bad) { snprintf(g_err, sizeof g_err, "%s: truncated or unexpected archive layout", path); return -1; }
  /* set_interval (aligner_table.hpp:198): powerV = log2(interval) */
  ix->powerV = 0;
  while ((1ull << ix->powerV) < ix->interval) ix->powerV++;
  return 0;
}

static int read_npos(sbo_index *ix, const char *path) {
  rd R = {fopen(path, "rb"), 0};
  if (!R.f) { snprintf(g_err, sizeof g_err, "cannot open %s", path); return -1; }
  unsigned char hdr[40];
  if (fread(hdr, 1, 40, R.f) != 40 || memcmp(hdr, ARCH_HDR, 40)) R.bad = 1;
  rskip(&R, 5);
  ix->n_npos = r64(&R);
  r32(&R);
  rskip(&R, 5);
  if (!R.bad) {
    ix->npos_key = (uint64_t *)malloc(ix->n_npos * 8 + 8);
    ix->npos_val = (uint64_t *)malloc(ix->n_npos * 8 + 8);
    for (uint64_t k = 0; k < ix->n_npos; ++k) { ix->npos_key[k] = r64(&R); ix->npos_val[k] = r64(&R); }
  }
  int bad = R.bad;
  fclose(R.f);
  if (bad) { snprintf(g_err, sizeof g_err, "%s: bad NposLen archive", path); return -1; }
  return 0;
}

/* readChrStartPos / readChrLen (binary-only): "name\tvalue\n" per line */
static int read_chr(sbo_index *ix, const char *start_path, const char *len_path) {
  size_t n = 0;
  char *d = slurp(start_path, &n);
  if (!d) { snprintf(g_err, sizeof g_err, "cannot open %s", start_path); return -1; }
  uint64_t cap = 16;
  ix->chr_start = (uint64_t *)malloc(cap * 8);
  ix->chr_name = (char **)malloc(cap * sizeof(char *));
  ix->n_chr = 0;
  char *save = NULL;
  for (char *line = strtok_r(d, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    char *tab = strrchr(line, '\t');
    if (!tab) continue;
    *tab = 0;
    if (ix->n_chr == cap) {
      cap *= 2;
      ix->chr_start = (uint64_t *)realloc(ix->chr_start, cap * 8);
      ix->chr_name = (char **)realloc(ix->chr_name, cap * sizeof(char *));
    }
    ix->chr_name[ix->n_chr] = strdup(line);
    ix->chr_start[ix->n_chr] = strtoull(tab + 1, NULL, 10);
    ix->n_chr++;
  }
  free(d);
  d = slurp(len_path, &n);
  if (!d) { snprintf(g_err, sizeof g_err, "cannot open %s", len_path); return -1; }
  ix->real_size = 0;
  save = NULL;
  for (char *line = strtok_r(d, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    char *tab = strrchr(line, '\t');
    if (tab) ix->real_size += strtoull(tab + 1, NULL, 10);
  }
  free(d);
  return 0;
}

sbo_index *sbo_index_load(const char *prefix_in) {
  char prefix[4000], path[4096], path2[4096];
  snprintf(prefix, sizeof prefix, "%s", prefix_in);
  size_t pl = strlen(prefix);
  if (pl == 0 || prefix[pl - 1] != '.') { prefix[pl] = '.'; prefix[pl + 1] = 0; } /* searcher.hpp:196-198 */
  sbo_index *ix = (sbo_index *)calloc(1, sizeof *ix);
  snprintf(path, sizeof path, "%st_table.bwt", prefix);
  if (read_table(ix, path)) { sbo_index_free(ix); return NULL; }
  snprintf(path, sizeof path, "%sNposLen.z", prefix);
  if (read_npos(ix, path)) { sbo_index_free(ix); return NULL; }
  snprintf(path, sizeof path, "%schrStart", prefix);
  snprintf(path2, sizeof path2, "%schrLen", prefix);
  if (read_chr(ix, path, path2)) { sbo_index_free(ix); return NULL; }
  return ix;
}

sbo_index *sbo_index_wrap(uint64_t interval, uint64_t n, uint64_t first_location, uint64_t real_size,
                          const uint64_t *c4, uint64_t *c_functions, uint64_t *occA, uint64_t *occC,
                          uint64_t *occG, uint64_t *occT, uint64_t occ_len, uint8_t *jbwt_seq, uint8_t *fbwt,
                          uint64_t *loc, uint64_t n_loc, uint64_t n_chr, const uint64_t *chr_start,
                          uint64_t n_npos, const uint64_t *npos_key, const uint64_t *npos_val) {
  sbo_index *ix = (sbo_index *)calloc(1, sizeof *ix);
  ix->borrowed = 1;
  ix->interval = interval;
  while ((1ull << ix->powerV) < interval) ix->powerV++;
  ix->n = n; ix->first_location = first_location; ix->real_size = real_size;
  for (int c = 0; c < 4; ++c) ix->c_function[(unsigned char)ACGT[c]] = c4[c];
  ix->c_functions = c_functions;
  ix->occ[0] = occA; ix->occ[1] = occC; ix->occ[2] = occG; ix->occ[3] = occT;
  ix->occ_len = occ_len;
  fill_occ_jbwt(ix);
  ix->jbwt_seq = jbwt_seq; ix->jbwt_len = n / 4 + 1;
  ix->fbwt = fbwt; ix->loc = loc; ix->n_loc = n_loc;
  ix->n_chr = n_chr;
  ix->chr_start = (uint64_t *)malloc(n_chr * 8 + 8);
  memcpy(ix->chr_start, chr_start, n_chr * 8);
  ix->chr_name = (char **)malloc(n_chr * sizeof(char *) + 8);
  for (uint64_t k = 0; k < n_chr; ++k) {
    char nm[32];
    snprintf(nm, sizeof nm, "chr%llu", (unsigned long long)(k + 1));
    ix->chr_name[k] = strdup(nm);
  }
  ix->n_npos = n_npos;
  ix->npos_key = (uint64_t *)malloc(n_npos * 8 + 8);
  ix->npos_val = (uint64_t *)malloc(n_npos * 8 + 8);
  memcpy(ix->npos_key, npos_key, n_npos * 8);
  memcpy(ix->npos_val, npos_val, n_npos * 8);
  return ix;
}

void sbo_index_free(sbo_index *ix) {
  if (!ix) return;
  if (!ix->borrowed) {
    free(ix->c_functions);
    for (int c = 0; c < 4; ++c) free(ix->occ[c]);
    free(ix->jbwt_seq); free(ix->fbwt); free(ix->loc);
  }
  for (uint64_t k = 0; k < ix->n_chr; ++k) free(ix->chr_name[k]);
  free(ix->chr_name); free(ix->chr_start); free(ix->npos_key); free(ix->npos_val);
  free(ix);
}

/* ------------------------------------------------------------------------ */
/* index build                                                               */
/* ------------------------------------------------------------------------ */
/* text = fwd + revcomp(fwd) + '$' (genome_pre_handler.hpp:210-228), kept as 2-bit codes */
typedef struct { const uint8_t *t; uint64_t m; /* m = 2G real bases; '$' at position m */ } text_t;

/* 32 bases starting at p as a big-endian base-4 number, positions >= m read as 0 */
static inline uint64_t key32(const text_t *T, uint64_t p) {
  uint64_t k = 0;
  uint64_t e = p + 32 <= T->m ? p + 32 : T->m;
  uint64_t q = p;
  for (; q < e; ++q) k = (k << 2) | T->t[q];
  if (q == p) return 0;
  if (q < p + 32) k <<= 2 * (p + 32 - q);
  return k;
}

static const text_t *g_T;
/* full suffix comparison, '$' smallest.  Any total suffix order is a valid sort_len-ordered sort
 * (split_sort.hpp / mkq_sort.hpp sort to depth sort_len and leave deeper ties unordered). */
static int suf_cmp_from(uint64_t a, uint64_t b, uint64_t off) {
  const text_t *T = g_T;
  if (a == b) return 0;
  for (;;) {
    uint64_t pa = a + off, pb = b + off;
    int a_out = pa + 32 > T->m, b_out = pb + 32 > T->m;
    uint64_t ka = pa <= T->m ? key32(T, pa) : 0, kb = pb <= T->m ? key32(T, pb) : 0;
    if (ka != kb) return ka < kb ? -1 : 1;
    if (a_out || b_out) return a > b ? -1 : 1; /* the shorter suffix meets '$' first */
    off += 32;
  }
}
typedef struct { uint64_t key, pos; } kp_t;
static int kp_cmp_deep(const void *x, const void *y) {
  return suf_cmp_from(((const kp_t *)x)->pos, ((const kp_t *)y)->pos, 32);
}

static void radix_sort_kp(kp_t *a, kp_t *tmp, uint64_t n, int bits) {
  for (int sh = 0; sh < bits; sh += 16) {
    uint64_t *cnt = (uint64_t *)calloc(65537, 8);
    for (uint64_t i = 0; i < n; ++i) cnt[((a[i].key >> sh) & 0xffff) + 1]++;
    for (int k = 0; k < 65536; ++k) cnt[k + 1] += cnt[k];
    for (uint64_t i = 0; i < n; ++i) tmp[cnt[(a[i].key >> sh) & 0xffff]++] = a[i];
    free(cnt);
    kp_t *s = a; a = tmp; tmp = s;
  }
  /* bits is a multiple of 32 here, so an even number of passes leaves the result in the original array */
}

sbo_index *sbo_index_from_text(const char *fwd, uint64_t g, uint64_t interval) {
  if (interval < 2 || (interval & (interval - 1)) || interval > 1024)
    FAIL("interval must be a power of two in 2..1024 (set_interval leaves powerV unset otherwise)");
  uint64_t m = 2 * g, n = m + 1;
  uint8_t *t = (uint8_t *)malloc(m + 64);
  for (uint64_t i = 0; i < g; ++i) {
    int b = base_code(fwd[i]);
    if (b < 0) { free(t); FAIL("non-ACGT base in text at %llu", (unsigned long long)i); }
    t[i] = (uint8_t)b;
  }
  for (uint64_t i = 0; i < g; ++i) t[g + i] = (uint8_t)(3 - t[g - 1 - i]);
  text_t T = {t, m};

  /* suffix array: suffixes 0..m (m is the lone '$') */
  kp_t *a = (kp_t *)malloc(n * sizeof(kp_t)), *tmp = (kp_t *)malloc(n * sizeof(kp_t));
  for (uint64_t p = 0; p < m; ++p) { a[p].key = key32(&T, p); a[p].pos = p; }
  radix_sort_kp(a, tmp, m, 64);
  free(tmp);
  g_T = &T;
  for (uint64_t i = 0; i < m;) {
    uint64_t j = i + 1;
    while (j < m && a[j].key == a[i].key) ++j;
    if (j - i > 1) {
      /* equal 32-base keys: suffixes shorter than 32 sort first (they met '$'), then deeper compare */
      qsort(a + i, j - i, sizeof(kp_t), kp_cmp_deep);
    }
    i = j;
  }
  /* NB kp_cmp_deep treats a suffix with fewer than 32 bases left correctly: key32 pads with 0 and
   * suf_cmp_from(.., 32) sees pa > m or a_out and orders the shorter one first. */

  sbo_index *ix = (sbo_index *)calloc(1, sizeof *ix);
  ix->interval = interval;
  while ((1ull << ix->powerV) < interval) ix->powerV++;
  ix->n = n;
  ix->real_size = g;
  ix->occ_len = n / interval + 1;
  for (int c = 0; c < 4; ++c) ix->occ[c] = (uint64_t *)calloc(ix->occ_len, 8);
  ix->jbwt_len = n / 4 + 1;
  ix->jbwt_seq = (uint8_t *)calloc(ix->jbwt_len, 1);
  ix->fbwt = (uint8_t *)malloc(n);
  ix->n_loc = 0;
  ix->loc = (uint64_t *)malloc((n / interval + 2) * 16);
  ix->c_functions = (uint64_t *)malloc((SBO_NKMER + 1) * 8);
  uint8_t *present = (uint8_t *)calloc(SBO_NKMER + 1, 1);
  fill_occ_jbwt(ix);

  uint64_t run[4] = {0, 0, 0, 0}, first_cnt[4] = {0, 0, 0, 0};
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t sa = i == 0 ? m : a[i - 1].pos; /* '$' suffix is SA[0] */
    if (i % interval == 0)
      for (int c = 0; c < 4; ++c) ix->occ[c][i / interval] = run[c];
    int bw;
    if (sa == 0) { ix->first_location = i; bw = 0; /* '$' stored as 'A' */ }
    else bw = t[sa - 1];
    run[bw]++;
    ix->jbwt_seq[i >> 2] |= (uint8_t)(bw << (6 - 2 * (i & 3)));
    ix->fbwt[i] = (uint8_t)(sa % interval);
    if (sa % interval == 0) { ix->loc[2 * ix->n_loc] = i; ix->loc[2 * ix->n_loc + 1] = sa; ix->n_loc++; }
    if (sa < m) first_cnt[t[sa]]++;
    /* c_functions rule: a suffix defines a bucket start iff it has 12 real bases
     * before '$', or it is the lone '$' suffix (code 0) */
    if (sa == m) { present[0] = 1; ix->c_functions[0] = i; }
    else if (sa + SBO_KMER <= m) {
      uint64_t code = 0;
      for (int k = 0; k < SBO_KMER; ++k) code = code * 4 + t[sa + k];
      if (!present[code]) { present[code] = 1; ix->c_functions[code] = i; }
    }
  }
  ix->c_functions[SBO_NKMER] = n;
  for (uint64_t code = SBO_NKMER; code-- > 0;)
    if (!present[code]) ix->c_functions[code] = ix->c_functions[code + 1];
  free(present);
  /* c_function (createAllTable, binary-only; recovered from the object code at .text+0x2675 and
   * +0x4064): c_function[first byte of suffix] = first SA index of that byte's run ('$' -> 0),
   * c_function[255] = n, then every zero entry 255..1 takes the next larger non-zero value;
   * entry 0 is left alone.  Hence c_function[1..'A'] = 1. */
  {
    uint64_t acc = 1;
    for (int c = 0; c < 4; ++c) {
      if (first_cnt[c]) ix->c_function[(unsigned char)ACGT[c]] = acc;
      acc += first_cnt[c];
    }
    ix->c_function[255] = n;
    uint64_t fill = 0;
    for (int v = 255; v >= 1; --v) {
      if (ix->c_function[v]) fill = ix->c_function[v];
      ix->c_function[v] = fill;
    }
  }
  free(a);
  free(t);
  return ix;
}

/* genome_pre_handler.hpp:13-132 Segment(std::string&, idx) + :244-310 parser_combo/updateNpos */
typedef struct { uint64_t off, len; } seg_t;
typedef struct { uint64_t *k, *v; uint64_t n, cap; } u64map; /* sorted by key, insert-if-absent */
static uint64_t map_insert(u64map *M, uint64_t key, uint64_t val) {
  uint64_t l = 0, h = M->n;
  while (l < h) { uint64_t mid = (l + h) >> 1; if (M->k[mid] < key) l = mid + 1; else h = mid; }
  if (l < M->n && M->k[l] == key) return l; /* std::map::insert keeps the existing value */
  if (M->n == M->cap) {
    M->cap = M->cap ? M->cap * 2 : 64;
    M->k = (uint64_t *)realloc(M->k, M->cap * 8);
    M->v = (uint64_t *)realloc(M->v, M->cap * 8);
  }
  memmove(M->k + l + 1, M->k + l, (M->n - l) * 8);
  memmove(M->v + l + 1, M->v + l, (M->n - l) * 8);
  M->k[l] = key; M->v[l] = val; M->n++;
  return l;
}

int sbo_build(const char *fasta_path, const char *prefix_in, uint64_t interval) {
  char prefix[4000], path[4096];
  snprintf(prefix, sizeof prefix, "%s", prefix_in);
  size_t pl = strlen(prefix);
  if (pl == 0 || prefix[pl - 1] != '.') { prefix[pl] = '.'; prefix[pl + 1] = 0; } /* indexer.hpp:140-142 */
  size_t fn = 0;
  char *fd = slurp(fasta_path, &fn);
  if (!fd) { snprintf(g_err, sizeof g_err, "cannot read %s", fasta_path); return -1; }
  snprintf(path, sizeof path, "%schrStart", prefix);
  FILE *fs = fopen(path, "wb");
  snprintf(path, sizeof path, "%schrLen", prefix);
  FILE *fl = fopen(path, "wb");
  if (!fs || !fl) { snprintf(g_err, sizeof g_err, "cannot write index files under %s", prefix); return -1; }
  char *seq = (char *)malloc(fn + 1); /* concatenated ACGT-only text */
  uint64_t g = 0, accumulated = 0;
  u64map NP = {0};
  /* fasta_reader_impl.hpp:65-110: header = whole trimmed line after '>', sequence = trimmed lines joined */
  size_t pos = 0;
  char *rec = (char *)malloc(fn + 1);
  while (pos < fn) {
    const char *name = NULL;
    size_t name_len = 0, rn = 0;
    int have_head = 0;
    while (pos < fn) {
      const char *nl = (const char *)memchr(fd + pos, '\n', fn - pos);
      size_t ll = nl ? (size_t)(nl - (fd + pos)) : fn - pos;
      const char *l = fd + pos;
      size_t tl = ll;
      trim(&l, &tl);
      if (tl && l[0] == '>') {
        if (have_head) break; /* next record: leave the line unread */
        have_head = 1;
        name = l + 1; name_len = tl - 1;
      } else {
        memcpy(rec + rn, l, tl);
        rn += tl;
      }
      pos += ll + (nl ? 1 : 0);
    }
    /* parser_combo: split into segments N*[ACGT]* */
    seg_t *segs = NULL;
    uint64_t nseg = 0, segcap = 0, idx = 0, len_no_n = 0;
    uint64_t g0 = g;
    while (idx < rn) {
      seg_t s = {0, 0};
      int fin = 0;
      while (!fin && idx < rn) {
        switch (rec[idx]) {
          case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u':
            fin = 1; break;
          case 'N': case 'n': ++s.off; ++idx; break;
          default: ++idx; break;
        }
      }
      fin = 0;
      while (!fin && idx < rn) {
        char c = rec[idx];
        switch (c) {
          case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': case 'U': case 'u':
            ++s.len; seq[g++] = (char)toupper((unsigned char)c); ++idx; break;
          case 'N': case 'n': fin = 1; break;
          default: ++idx; break;
        }
      }
      if (nseg == segcap) { segcap = segcap ? segcap * 2 : 16; segs = (seg_t *)realloc(segs, segcap * sizeof(seg_t)); }
      segs[nseg++] = s;
      len_no_n += s.len;
    }
    (void)g0;
    if (nseg == 0) { free(segs); snprintf(g_err, sizeof g_err, "empty FASTA record (undefined upstream)"); return -1; }
    /* run_handler callback: genome_pre_handler.hpp:190-203 */
    fprintf(fs, "%.*s\t%llu\n", (int)name_len, name ? name : "", (unsigned long long)accumulated);
    fprintf(fl, "%.*s\t%llu\n", (int)name_len, name ? name : "", (unsigned long long)len_no_n);
    accumulated += len_no_n;
    /* updateNpos: genome_pre_handler.hpp:279-310 */
    uint64_t si = 0, stop = nseg - 1, last;
    if (NP.n == 0) last = map_insert(&NP, 0, segs[0].off);
    else { NP.v[NP.n - 1] = segs[0].off; last = NP.n - 1; }
    while (si != stop) {
      uint64_t key = NP.k[last] + segs[si].len;
      uint64_t val = NP.v[last] + segs[si + 1].off;
      ++si;
      last = map_insert(&NP, key, val);
    }
    if (segs[si].len != 0) map_insert(&NP, NP.k[last] + segs[si].len, NP.v[last]);
    free(segs);
  }
  free(rec);
  fclose(fs);
  fclose(fl);
  /* NposLen.z: genome_pre_handler.hpp:231-236 (std::map<u64,u64> archive) */
  snprintf(path, sizeof path, "%sNposLen.z", prefix);
  FILE *fz = fopen(path, "wb");
  if (!fz) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
  fwrite(ARCH_HDR, 1, 40, fz);
  wz(fz, 5); w64(fz, NP.n); w32(fz, 0); wz(fz, 5);
  for (uint64_t k = 0; k < NP.n; ++k) { w64(fz, NP.k[k]); w64(fz, NP.v[k]); }
  fclose(fz);
  free(NP.k); free(NP.v);
  free(fd);
  for (uint64_t i = 0; i < g; ++i)
    if (seq[i] == 'U') { free(seq); snprintf(g_err, sizeof g_err, "'U' bases are not reverse-complemented upstream"); return -1; }
  sbo_index *ix = sbo_index_from_text(seq, g, interval);
  free(seq);
  if (!ix) return -1;
  snprintf(path, sizeof path, "%st_table.bwt", prefix);
  int rc = sbo_index_save(ix, path);
  sbo_index_free(ix);
  return rc;
}
