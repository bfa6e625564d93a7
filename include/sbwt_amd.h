/*
 * sbwt_amd.h -- C ABI of the MI355X (gfx950) implementation of sBWT's read-mapping hot path.
 *
 * The upstream program exposes no plugin / FFI interface; the seam this library replaces is the
 * per-read functor that Searcher::search hands to Job_distributer (src/aligner/searcher.hpp:346-372),
 * which calls Searcher::start_sbwt_match (searcher.hpp:386-458) over the tables loaded by
 * Searcher_impl::load_table (searcher.hpp:190-204).  Every entry point below cites the upstream
 * interface it stands in for.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions: every function returns 0 (SBWT_OK) or a negative SBWT_ERR_* code; the message for
 * the calling thread's last failure is sbwt_last_error().  Nothing throws or exits across the ABI.
 * Calls on distinct sbwt_ctx objects may run concurrently (this mirrors the concurrent functor
 * calls of the upstream thread pool); one sbwt_ctx serves one call at a time.
 */
#ifndef SBWT_AMD_H_
#define SBWT_AMD_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SBWT_OK 0
#define SBWT_ERR_IO (-1)            /* file missing / unreadable / unwritable */
#define SBWT_ERR_FORMAT (-2)        /* index files damaged (main.cpp:34-47,190-194) */
#define SBWT_ERR_HIP (-3)           /* HIP runtime failure, or no gfx950 device */
#define SBWT_ERR_ARG (-4)           /* bad argument */
#define SBWT_ERR_ILLEGAL_CHAR (-5)  /* read with a byte outside ACGT: upstream throws "illegal char" (searcher.hpp:399) and aborts */
#define SBWT_ERR_TOO_SHORT (-6)     /* read shorter than 12: upstream substr throws (searcher.hpp:404) and aborts */
#define SBWT_ERR_CAPACITY (-7)      /* caller-provided output buffer too small */
#define SBWT_ERR_NOMEM (-8)

typedef struct sbwt_index sbwt_index; /* the loaded tables: Aligner_table<BWT_Aligner> (aligner_table.hpp:49-302) */
typedef struct sbwt_ctx sbwt_ctx;     /* one stream + workspace: one worker of the upstream pool */

/* One reported alignment, in upstream emission order (push_position, searcher.hpp:504-539). */
typedef struct sbwt_hit {
  uint64_t pos0;   /* 0-based position in the chromosome, N offset added back (SAM POS = pos0 + 1) */
  uint32_t chr_id; /* index into the chromosome table (order of prefix.chrStart) */
  uint32_t is_rc;  /* upstream isRC: 1 -> FLAG 0 / SEQ as read; 0 -> FLAG 16 / SEQ reverse-complemented */
} sbwt_hit;

/* Per-read result of start_sbwt_match (searcher.hpp:386-458). */
typedef struct sbwt_read_result {
  uint64_t sa_lo, sa_hi; /* SA interval after the backward search (searcher.hpp:404-410) */
  uint64_t first_hit;    /* index of this read's first sbwt_hit in the hits buffer */
  int32_t nh;            /* number of hits that survived the filters = upstream return value / NH */
  uint32_t flags;        /* SBWT_RF_* */
} sbwt_read_result;
#define SBWT_RF_SUPPRESSED 1u /* raw interval larger than `limit` (searcher.hpp:415) */
#define SBWT_RF_TRUNCATED 2u  /* the list was cut at the raw interval size: it is longer (upstream's walk can report a row twice where a read
                               * of L bases has L + interval - 1 beyond the depth the index's suffix order is sorted to; such lists are
                               * reported whole, in slots behind all others) AND does not fit the record (16 bits of NH in 16 bytes) */

/* Compact per-read result (sbwt_map_batch_device_compact / sbwt_map_batch_compact): everything start_sbwt_match
 * yields for a read in 16 bytes, the unique hit included, so that a read costs one quarter of a 64-byte write and
 * 16 bytes of the host link instead of 48.
 *   w0 = sa_lo | (sa_hi - sa_lo) << 40
 *   w1 = payload (40 bits) | aux (16 bits) << 40 | meta (8 bits) << 56
 * meta & SBWT_CR_UNIQUE: NH = 1, payload = pos0, aux = chr_id, SBWT_CR_RC = is_rc.
 * meta & SBWT_CR_MULTI: payload = index of the read's first hit in hits[] (which holds ONLY the lists of multi-hit
 *   reads here), aux = NH (number of hits that survived the filters; the list is in upstream emission order).
 * neither: NH = 0.  SBWT_CR_SUPPRESSED / _TRUNCATED = SBWT_RF_*; SBWT_CR_INVALID: length < 12 or beyond the record.
 * meta & SBWT_CR_WIDE: the result does not fit (interval of 2^24 rows or more, chr_id above 65,535, a multi-hit
 *   read with more than 65,535 raw rows, a position of 2^40 or more): payload = index into wide[], which holds the
 *   plain record and hit (r.first_hit indexes hits[] as above).
 * meta & SBWT_CR_NO_SA (only with SBWT_MAP_HITS_ONLY): w0 is 0, the SA interval was not computed. */
typedef struct sbwt_compact_result {
  uint64_t w0, w1;
} sbwt_compact_result;
#define SBWT_CR_UNIQUE 0x01u
#define SBWT_CR_RC 0x02u
#define SBWT_CR_MULTI 0x04u
#define SBWT_CR_SUPPRESSED 0x08u
#define SBWT_CR_TRUNCATED 0x10u
#define SBWT_CR_INVALID 0x20u
#define SBWT_CR_WIDE 0x40u
#define SBWT_CR_NO_SA 0x80u
typedef struct sbwt_wide_result {
  sbwt_read_result r;
  sbwt_hit hit; /* the unique hit when r.nh == 1 and the read is not a multi-hit one */
} sbwt_wide_result;
/* mode bits of the compact entry points */
#define SBWT_MAP_HITS_ONLY 1u /* only the hits are wanted (what the SAM output needs): a read whose located search
                                 prefix and the 32 bases of text stored with it settle the outcome (the one hit, or a
                                 dead interval) is reported without its SA interval (SBWT_CR_NO_SA) and without the
                                 access that would fetch it.  Hits, NH and flags are unchanged. */

typedef struct sbwt_index_info {
  uint64_t n;              /* number of suffixes = 2 * real_size + 1 */
  uint64_t real_size;      /* Aligner_table::_realSize = sum of chrLen */
  uint64_t interval;       /* Aligner_table::interval */
  uint64_t first_location; /* SA index of text position 0 */
  uint64_t n_chr, n_npos, n_loc;
  uint64_t device_bytes;   /* HBM held by the device layout */
  int device;
  uint32_t kmer_order;     /* bases the direct search-prefix table memoises (12 = none: c_functions only) */
  uint32_t hash_order;     /* bases the hash table of located prefixes covers (0 = none) */
  uint32_t hash_text;      /* 1: its slots carry the 32 bases of text behind the prefix (SBWT_MAP_HITS_ONLY fast pass) */
} sbwt_index_info;

const char *sbwt_last_error(void);
/* Number of usable gfx950 devices; fails (does not fall back) when there is none. */
int sbwt_device_count(int *count);

/* ---- index ------------------------------------------------------------------------------- */

/* Replaces Searcher_impl::load_table (searcher.hpp:190-204 -> Aligner_table::readTable / readNPosLen /
 * readChrStartPos / readChrLen): reads prefix.{t_table.bwt,NposLen.z,chrStart,chrLen} in the on-disk
 * format `sbwt build` writes and lays the tables out in the HBM of `device`. */
int sbwt_index_load(const char *prefix, int device, sbwt_index **out);

/* Same device layout from reference-format tables already in host memory (what readTable would have
 * produced): used by the index builder and by callers that keep the tables themselves.
 * c_function4 = c_function['A','C','G','T']; occ[c] has n/interval+1 entries; location_table is
 * n_loc (i, SA[i]) pairs; chr_names may be NULL (names chr1..). */
int sbwt_index_from_tables(int device, uint64_t interval, uint64_t n, uint64_t first_location,
                           uint64_t real_size, const uint64_t *c_function4, const uint64_t *c_functions,
                           const uint64_t *occA, const uint64_t *occC, const uint64_t *occG,
                           const uint64_t *occT, const uint8_t *jbwt_seq, const uint8_t *fbwt,
                           const uint64_t *location_table, uint64_t n_loc, uint64_t n_chr,
                           const uint64_t *chr_start, const char *const *chr_names, uint64_t n_npos,
                           const uint64_t *npos_key, const uint64_t *npos_val, sbwt_index **out);

/* Frees the index and its device memory.  Every sbwt_ctx created on it must have been freed before: a ctx reads its
 * index until sbwt_ctx_free returns. */
void sbwt_index_free(sbwt_index *ix);
int sbwt_index_get_info(const sbwt_index *ix, sbwt_index_info *info);
/* chromosome name i (Aligner_table::chr_start_pos values); NULL when out of range */
const char *sbwt_index_chr_name(const sbwt_index *ix, uint64_t i);

/* Replaces Indexer::build (indexer.hpp:137-229: Genome_pre_handler::run_handler, the suffix sort,
 * Aligner_table::createAllTable and saveTable): FASTA -> prefix.{t_table.bwt,NposLen.z,chrStart,chrLen},
 * suffix ordering on the GPU.  sort_len is accepted for CLI compatibility; the order produced is a
 * full suffix order, which is a valid sort_len-order for every sort_len. */
int sbwt_build_index(const char *fasta_path, const char *prefix, uint64_t sort_len, uint64_t interval, int device);

/* Reference-format tables in host memory: what Aligner_table::readTable leaves in memory
 * (aligner_table.hpp:98-161).  Produced by sbwt_index_build_device for callers that want to write the
 * index files themselves or hand the tables to another consumer.  Free with sbwt_host_tables_free. */
typedef struct sbwt_host_tables {
  uint64_t interval, n, first_location, real_size, occ_len, n_loc;
  uint64_t c_function4[4];   /* c_function['A','C','G','T'] */
  uint64_t *c_functions;     /* 4^12 + 1 */
  uint64_t *occ[4];          /* occ_function['A','C','G','T'], occ_len entries each */
  uint64_t *location_table;  /* n_loc pairs (i, SA[i]) */
  uint8_t *jbwt_seq;         /* n/4 + 1 bytes */
  uint8_t *fbwt;             /* n bytes */
} sbwt_host_tables;

/* Replaces Indexer::build_index (indexer.hpp:165-229: suffix ordering + Aligner_table::createAllTable)
 * for a genome already on the device: d_fwd_packed is a DEVICE pointer to the concatenated ACGT-only
 * forward text, 2-bit codes in the read encoding below (16 bases per 32-bit word).  chr_* / npos_* are
 * the host tables Genome_pre_handler::run_handler would have written (genome_pre_handler.hpp:183-236).
 * Either output may be NULL. */
int sbwt_index_build_device(int device, const uint32_t *d_fwd_packed, uint64_t g, uint64_t interval,
                            uint64_t n_chr, const uint64_t *chr_start, const char *const *chr_names,
                            uint64_t n_npos, const uint64_t *npos_key, const uint64_t *npos_val,
                            sbwt_index **out_index, sbwt_host_tables **out_tables);
void sbwt_host_tables_free(sbwt_host_tables *t);
/* Aligner_table::saveTable (aligner_table.hpp:213): writes prefix.t_table.bwt byte-compatible with upstream */
int sbwt_host_tables_save(const sbwt_host_tables *t, const char *table_path);

/* ---- mapping ----------------------------------------------------------------------------- */

/* One worker: a HIP stream plus scratch sized for batches of up to max_reads reads.  Free it before its index. */
int sbwt_ctx_create(const sbwt_index *ix, uint64_t max_reads, sbwt_ctx **out);
void sbwt_ctx_free(sbwt_ctx *ctx);

/* How reads are shared out over devices or ranks: share k of `parts` is the contiguous range [*begin, *end) of
 * n_items; the ranges tile [0, n_items) in order of k and their sizes differ by at most one.  Used by
 * sbwt_map_fastq_multi for every batch and by the one-process-per-GPU driver (bench.py, sbwt_amd/shard.py) for the
 * whole input: shares in order of k = input order (job_distributer.hpp:53-67 hands out groups in reading order). */
int sbwt_shard_range(uint64_t n_items, uint32_t parts, uint32_t k, uint64_t *begin, uint64_t *end);

/* Read encoding.  Base j of read r is the 2-bit code (A=0,C=1,G=2,T=3) at bits [2*(j%16), 2*(j%16)+1]
 * of 32-bit word  r*words_per_read + j/16 ; words_per_read is a multiple of 4 (16-byte records).
 * sbwt_pack_reads packs ASCII reads (concatenated, offsets[n_reads+1]) and reports the first read the
 * upstream program would have aborted on (bad_read, SBWT_ERR_ILLEGAL_CHAR / SBWT_ERR_TOO_SHORT). */
uint32_t sbwt_words_per_read(uint32_t max_len);
int sbwt_pack_reads(const char *seqs, const uint64_t *offsets, uint64_t n_reads, uint32_t words_per_read,
                    uint32_t *packed, uint32_t *lens, uint64_t *bad_read);

/* Replaces the body of the functor at searcher.hpp:348-352 (start_sbwt_match, searcher.hpp:386-458:
 * reverse complement, 12-mer init :83-96, backward search :102-105, suppression :415, level-0 locate
 * :424-443, find_possible :460-503, push_position :504-539) for a whole batch.  All pointers are DEVICE
 * pointers on the index's device; nothing is copied to or from the host.  results[n_reads], hits[hits_cap].
 * limit / strand are `sbwt map -m / -s` (main.cpp:167-168).
 * Layout of hits[]: the hit of a read with exactly one surviving hit lives at the read's own index
 * (first_hit == read index), so hits_cap must be at least n_reads; the lists of multi-hit reads follow behind
 * the first n_reads slots, each contiguous and in upstream emission order (first_hit says where; the ORDER of the lists
 * behind the first n_reads slots is not part of the contract and may differ from call to call).  Multi-hit reads
 * of one batch with the same SA interval and length have the same list (upstream would compute it once per read,
 * searcher.hpp:424-503): it is located once and their records carry the SAME first_hit, so lists may alias --
 * treat hits[] as read-only per read.  Slots of reads without a hit, and the slots reserved for reads that share
 * another read's list, are not written.  *n_hits_host (host pointer, may be NULL) receives the number of slots
 * in use = n_reads + the multi-hit lists; on SBWT_ERR_CAPACITY it is what a retry needs. */
int sbwt_map_batch_device(sbwt_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_lens, uint64_t n_reads,
                          uint32_t words_per_read, int limit, int strand, sbwt_read_result *d_results,
                          sbwt_hit *d_hits, uint64_t hits_cap, uint64_t *n_hits_host);

/* sbwt_map_batch_device with the compact records above: d_out[n_reads], d_hits[hits_cap] for the lists of multi-hit
 * reads only (hits_cap may be 0 when there are none), d_wide[wide_cap] for results that do not fit a record (rare;
 * 1,024 slots are plenty outside adversarial inputs).  counts_host (host pointer, may be NULL) receives
 * {hit slots in use, wide slots in use}; on SBWT_ERR_CAPACITY they are what a retry needs.  Same search, locate
 * and filters as sbwt_map_batch_device (start_sbwt_match, searcher.hpp:386-539); `mode`: SBWT_MAP_* bits. */
int sbwt_map_batch_device_compact(sbwt_ctx *ctx, const uint32_t *d_packed, const uint32_t *d_lens, uint64_t n_reads,
                                  uint32_t words_per_read, int limit, int strand, uint32_t mode,
                                  sbwt_compact_result *d_out, sbwt_hit *d_hits, uint64_t hits_cap,
                                  sbwt_wide_result *d_wide, uint64_t wide_cap, uint64_t counts_host[2]);
/* Same with HOST buffers (20 bytes in and 16 bytes out per read across the link). */
int sbwt_map_batch_compact(sbwt_ctx *ctx, const uint32_t *packed, const uint32_t *lens, uint64_t n_reads,
                           uint32_t words_per_read, int limit, int strand, uint32_t mode, sbwt_compact_result *out,
                           sbwt_hit *hits, uint64_t hits_cap, sbwt_wide_result *wide, uint64_t wide_cap,
                           uint64_t counts[2]);
/* Host helper: compact records (+ wide[], + the lists in list_hits[]) -> the plain layout of sbwt_map_batch:
 * results[n_reads], hits[] with the unique hit of read r at hits[r] and the lists behind the first n_reads slots in
 * the order of list_hits (hits must hold n_reads + n_list_hits entries).  Reads flagged SBWT_CR_NO_SA get sa_lo =
 * sa_hi = 0. */
int sbwt_compact_expand(const sbwt_compact_result *compact, uint64_t n_reads, const sbwt_wide_result *wide,
                        uint64_t n_wide, const sbwt_hit *list_hits, uint64_t n_list_hits, sbwt_read_result *results,
                        sbwt_hit *hits);

/* sbwt_map_batch_device for callers that cannot allocate.  No counterpart upstream (its per-read vectors live on
 * the host heap, searcher.hpp:419-422); it exists for callers that must not reach hipMalloc: tight serving loops,
 * harnesses that forbid allocation, code that runs between the captures of a HIP graph.  (The call itself cannot be
 * captured INTO a graph: it synchronises `stream` and reads its counters back on the host to report n_hits and
 * SBWT_ERR_CAPACITY.)  No sbwt_ctx: the scratch of the kernels
 * (counters, the two read queues, per-wave totals, the depth-first stacks: 14.1 MiB) is a file-scope device array
 * that every GPU the library runs on holds until the process ends; its address and a 64-byte pinned read-back
 * buffer are set up per device by the call that creates an index there.  This call itself allocates and frees
 * nothing (no hipMalloc / hipFree / stream or event creation), on `stream` (a hipStream_t; NULL = the default
 * stream), which it synchronises before returning.  Same arguments, per-read records, per-read hit lists and error
 * codes as sbwt_map_batch_device (WHERE a list lies behind the first n_reads slots may differ: this path has no
 * grouping table, so reads with one interval get a list each instead of sharing one); at most 1,048,576 reads per call (SBWT_ERR_ARG beyond); no kernel timing
 * (sbwt_ctx_last_timing / _stats have no counterpart here).  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU race; calls on different GPUs, and sbwt_ctx calls beside it, do not interfere. */
int sbwt_map_batch_device_static(const sbwt_index *ix, void *stream, const uint32_t *d_packed,
                                 const uint32_t *d_lens, uint64_t n_reads, uint32_t words_per_read, int limit,
                                 int strand, sbwt_read_result *d_results, sbwt_hit *d_hits, uint64_t hits_cap,
                                 uint64_t *n_hits_host);

/* Same with HOST buffers (pinned staging + async copies on the ctx stream). */
int sbwt_map_batch(sbwt_ctx *ctx, const uint32_t *packed, const uint32_t *lens, uint64_t n_reads,
                   uint32_t words_per_read, int limit, int strand, sbwt_read_result *results,
                   sbwt_hit *hits, uint64_t hits_cap, uint64_t *n_hits);

/* Page-locked host memory for the buffers handed to sbwt_map_batch: the copies then run as direct DMA at the
 * link's rate instead of through the runtime's staging of pageable memory (DESIGN.md section 4 has both rates).
 * Upstream has no counterpart (its vectors never leave the host, searcher.hpp:346-372).  Free with
 * sbwt_host_free.  Any other host memory is accepted by sbwt_map_batch too. */
int sbwt_host_alloc(uint64_t bytes, void **out);
void sbwt_host_free(void *p);

/* Milliseconds the kernels of the last sbwt_map_batch* call on this ctx took, measured with HIP events
 * on the ctx stream: [0] whole device pass, [1] search+unique-locate kernel, [2] multi-hit locate. */
int sbwt_ctx_last_timing(const sbwt_ctx *ctx, float ms[3]);
/* The same and more: [0..2] as above, [3] the hits-only fast pass (k_settle) of a compact call, [4] the general
 * kernels over the reads it left pending ([1] = [3] + their search kernel); values beyond read 0. */
int sbwt_ctx_last_timing_ex(const sbwt_ctx *ctx, float *ms, uint32_t n);
/* Counters of the last sbwt_map_batch* call on this ctx: [0] random index accesses the search kernel needed
 * (one per 128-byte line of an index table; the unit of the gather roofline), [1] reads handed to
 * the multi-hit locate, [2] hit slots in use, [3] multi-hit reads that needed the depth-first walk.  No upstream counterpart: measurement only. */
int sbwt_ctx_last_stats(const sbwt_ctx *ctx, uint64_t stats[4]);
/* The same and more: the first n counters of [0..3] as above, [4] hash_tab lookups the search kernel made, [5] those
 * that found their search prefix in it (the hash hit rate of a workload), [6] index lines the hits-only
 * fast pass (k_settle) read, [7] reads it left to the general kernels, [8] multi-hit reads whose record
 * points at the list of another read of the batch with the same interval and length instead of a list of their own
 * (see first_hit), [9] of the reads in [3], those the level-by-level form of the walk located (k_multi_bfs; [11]: those put
 * together from memoised chains, k_multi_chain; the rest took the depth-first kernel; [12]: raw rows of the multi-hit
 * reads the locate kernels located themselves, a list shared by several reads counted once; [13]: sub-batches whose
 * pending pass was launched without waiting for the fast pass), [10] reads
 * the fast pass settled as mapping nowhere because their first hash_order bases occur nowhere in the text (a miss in a complete
 * hash_tab); counters beyond the ones the library knows read 0. */
int sbwt_ctx_last_stats_ex(const sbwt_ctx *ctx, uint64_t *stats, uint32_t n);
/* The random-access roofline of the device, measured over the index's own largest table ("a random
 * granule gather microbenchmark over a table of the same footprint on the same GPU"): every lane walks
 * chains_per_lane (1, 2 or 4) chains of chain_len dependent 16-byte loads at pseudo-random 128-byte lines;
 * workgroups_per_cu workgroups of 256 lanes per CU (0 = 8).  *accesses_per_s = line accesses per second (best of
 * three launches), *table_bytes (may be NULL) the footprint probed.  No upstream counterpart: measurement only. */
int sbwt_probe_random_access(const sbwt_index *ix, uint32_t chains_per_lane, uint32_t chain_len,
                             uint32_t workgroups_per_cu, double *accesses_per_s, uint64_t *table_bytes);
/* The gather as the mapping kernels meet it: n_accesses independent random 128-byte lines of the same table whose
 * addresses come from a streamed 16-byte input record each, a 16-byte result written per access (3.2 GB of temporary
 * device memory per 10^8 accesses).  No upstream counterpart: measurement only (bench.py, roofline.gather_roofline). */
int sbwt_probe_gather_stream(const sbwt_index *ix, uint64_t n_accesses, uint32_t workgroups_per_cu, double *accesses_per_s);
/* raw hipStream_t of the ctx, for callers that want to order their own work after a batch */
void *sbwt_ctx_stream(const sbwt_ctx *ctx);

/* Replaces Searcher::search (searcher.hpp:336-373) + Job_distributer<M_T>::distribute_jobs
 * (job_distributer.hpp:37-90): FASTQ file -> SAM file ("stdout"/"-" = standard output), records in
 * input order.  stats = {total reads, mapped, single, multi, SAM records} (searcher.hpp:353-370).
 * n_threads = host threads for parsing / SAM formatting (`sbwt map -n`). */
int sbwt_map_fastq(const sbwt_index *ix, const char *fastq_path, const char *sam_path, int limit, int strand,
                   int n_threads, uint64_t stats[5]);
/* The same over several GPUs: ixs[0..n_ix) are replicas of ONE index, each loaded on its own device
 * (sbwt_index_load(prefix, device, ..) once per device).  Every batch of reads is cut into n_ix contiguous shares,
 * one per device, mapped at the same time by one host thread and ctx per device; there is no exchange between the
 * devices, and the SAM records leave in input order whatever the number of devices.  Upstream's only parallel
 * driver is the thread pool of Job_distributer<M_T>::distribute_jobs (job_distributer.hpp:37-90, sized by
 * GlobalPool.ChangePoolSize, main.cpp:231); this is its multi-device analogue. */
int sbwt_map_fastq_multi(const sbwt_index *const *ixs, int n_ix, const char *fastq_path, const char *sam_path, int limit,
                         int strand, int n_threads, uint64_t stats[5]);

#ifdef __cplusplus
}
#endif
#endif
