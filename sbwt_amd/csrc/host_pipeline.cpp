// host_pipeline.cpp -- FASTQ -> SAM driver above the batch mapper.
// Stands in for Searcher::search (searcher.hpp:336-373), Job_distributer<M_T>::distribute_jobs
// (job_distributer.hpp:37-90), the FASTQ reader (fastq_reader_impl.hpp:67-111) and the SAM text of
// position2sam / Sam::operator<< (searcher.hpp:540-598, sam.hpp:131-140, tuple_utility.hpp:54-64).
// Records are written in input order (what upstream produces with -n 1; with -n > 1 upstream emits the
// same lines with 1000-read groups in arbitrary order).
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sbwt_amd.h"
#include "index_file.hpp"
#include "index_impl.hpp"
#include "pack.hpp"

namespace sbwt {

struct Span {
  const char* p;
  uint32_t n;
};
struct Record {
  Span name, seq, qual;  // trimmed header line (with its first character), sequence, quality
};

// boost::trim: std::isspace on both ends
static inline Span trimmed(const char* p, const char* e) {
  while (p < e && isspace((unsigned char)*p)) ++p;
  while (e > p && isspace((unsigned char)e[-1])) --e;
  return Span{p, (uint32_t)(e - p)};
}

// fastq_reader_impl.hpp:81-98: four getline calls; the record only counts when the stream is still
// good() after the fourth, i.e. when all four lines are newline-terminated
static bool next_record(const char*& p, const char* end, Record& r) {
  if (p >= end) return false;
  const char* l[5];
  l[0] = p;
  for (int k = 0; k < 4; ++k) {
    const char* nl = static_cast<const char*>(memchr(l[k], '\n', (size_t)(end - l[k])));
    if (!nl) return false;
    l[k + 1] = nl + 1;
  }
  r.name = trimmed(l[0], l[1] - 1);
  r.seq = trimmed(l[1], l[2] - 1);
  r.qual = trimmed(l[3], l[4] - 1);
  p = l[4];
  return true;
}

// growing text buffer without zero fill; room(k) guarantees k writable bytes at end()
struct TextBuf {
  char* p = nullptr;
  size_t n = 0, cap = 0;
  TextBuf() = default;
  TextBuf(const TextBuf&) = delete;
  TextBuf& operator=(const TextBuf&) = delete;
  TextBuf(TextBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
  TextBuf& operator=(TextBuf&& o) noexcept {
    if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; }
    return *this;
  }
  ~TextBuf() { free(p); }
  bool room(size_t k) {
    if (cap - n >= k) return true;
    size_t c = cap ? cap * 2 : (size_t)1 << 16;
    while (c - n < k) c *= 2;
    char* q = static_cast<char*>(realloc(p, c));
    if (!q) return false;
    p = q;
    cap = c;
    return true;
  }
};

static inline char* put_u64(char* o, uint64_t v) {
  char t[24];
  int k = 0;
  do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (k) *o++ = t[--k];
  return o;
}

// reverse complement of ASCII ACGT, eight bases per step: A<->T differ by 0x15, C<->G by 0x04, and bit 1 tells the
// pairs apart (A 0x41, T 0x54: clear; C 0x43, G 0x47: set).  Bytes outside ACGT never reach the formatter
// (pack_read rejects the read first).
static inline uint64_t comp8(uint64_t x) {
  const uint64_t m = (x >> 1) & 0x0101010101010101ull;  // 1 in every C/G byte
  return x ^ (0x1515151515151515ull - m * 0x11ull);      // 0x15 for A/T, 0x04 for C/G
}
static inline void revcomp(const char* s, uint32_t n, char* out) {
  uint32_t k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t x;
    memcpy(&x, s + n - 8 - k, 8);
    x = __builtin_bswap64(comp8(x));
    memcpy(out + k, &x, 8);
  }
  for (; k < n; ++k) {
    const unsigned char c = (unsigned char)s[n - 1 - k];
    out[k] = (char)(c ^ ((c & 2) ? 0x04 : 0x15));
  }
}
static inline void reverse_bytes(const char* s, uint32_t n, char* out) {
  uint32_t k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t x;
    memcpy(&x, s + n - 8 - k, 8);
    x = __builtin_bswap64(x);
    memcpy(out + k, &x, 8);
  }
  for (; k < n; ++k) out[k] = s[n - 1 - k];
}

// One device's share of a batch, as sbwt_map_batch_compact left it: records of reads [r0, r1) of the batch, the
// lists of its multi-hit reads, its wide records.
struct PartView {
  uint64_t r0, r1;
  const sbwt_compact_result* rec;  // indexed by read - r0
  const sbwt_hit* lists;
  const sbwt_wide_result* wide;
  uint64_t n_wide;
};

// NH and the hits of one read, whatever form its record takes
struct ReadHits {
  int32_t nh;
  sbwt_hit one;          // nh == 1 from a compact record
  const sbwt_hit* list;  // nh >= 1 otherwise
};
static inline ReadHits hits_of(const PartView& P, uint64_t r) {
  const sbwt_compact_result c = P.rec[r - P.r0];
  const uint32_t meta = (uint32_t)(c.w1 >> 56);
  const uint64_t payload = c.w1 & kPosMask;
  ReadHits h;
  h.nh = 0;
  h.list = nullptr;
  if (meta & SBWT_CR_WIDE) {
    if (payload >= P.n_wide) return h;
    const sbwt_wide_result& w = P.wide[payload];
    h.nh = w.r.nh;
    if (meta & SBWT_CR_MULTI) h.list = P.lists + w.r.first_hit;
    else { h.one = w.hit; h.list = nullptr; }
    return h;
  }
  if (meta & SBWT_CR_UNIQUE) {
    h.nh = 1;
    h.one.pos0 = payload;
    h.one.chr_id = (uint32_t)(c.w1 >> kPosBits) & 0xFFFFu;
    h.one.is_rc = (meta & SBWT_CR_RC) ? 1u : 0u;
  } else if (meta & SBWT_CR_MULTI) {
    h.nh = (int32_t)((c.w1 >> kPosBits) & 0xFFFFu);
    h.list = P.lists + payload;
  }
  return h;
}

// SAM text of reads [r0, r1) of a batch (position2sam / Sam::operator<<, searcher.hpp:540-598, sam.hpp:131-140);
// counts the reads into st (searcher.hpp:353-370) on the way
static bool format_range(const sbwt_index* ix, const Record* recs, const PartView* parts, int n_parts, uint64_t r0, uint64_t r1,
                         TextBuf& out, uint64_t st[5]) {
  size_t longest_chr = 0;
  for (const auto& nm : ix->chr.names) longest_chr = std::max(longest_chr, nm.size());
  std::vector<char> rc_seq, rev_qual;
  int part = 0;
  for (uint64_t r = r0; r < r1; ++r) {
    while (part + 1 < n_parts && r >= parts[part].r1) ++part;
    const ReadHits H = hits_of(parts[part], r);
    const int32_t nh = H.nh;
    st[0]++;
    if (nh <= 0) continue;
    st[4] += (uint64_t)nh;
    st[1]++;
    if (nh == 1) st[2]++; else st[3]++;
    const Record& q = recs[r];
    const uint32_t L = q.seq.n, QL = q.qual.n;
    bool have_rev = false;
    const size_t line_max = (size_t)q.name.n + longest_chr + L + QL + 96;
    if (!out.room(line_max * (size_t)nh)) return false;
    char* o = out.p + out.n;
    for (int32_t h = 0; h < nh; ++h) {
      const sbwt_hit& hit = H.list ? H.list[h] : H.one;
      if (!hit.is_rc && !have_rev) {  // searcher.hpp:551-571: SEQ reverse-complemented, QUAL reversed
        if (rc_seq.size() < L + 8) rc_seq.resize(L + 8);
        if (rev_qual.size() < QL + 8) rev_qual.resize(QL + 8);
        revcomp(q.seq.p, L, rc_seq.data());
        reverse_bytes(q.qual.p, QL, rev_qual.data());
        have_rev = true;
      }
      // every field, the last included, is followed by a tab (tuple_utility.hpp:54-64), then '\n'
      if (q.name.n > 1) { memcpy(o, q.name.p + 1, q.name.n - 1); o += q.name.n - 1; }  // Fastq::getName: substr(1)
      *o++ = '\t';
      if (hit.is_rc) *o++ = '0'; else { *o++ = '1'; *o++ = '6'; }  // searcher.hpp:551-593
      *o++ = '\t';
      const std::string& chr = ix->chr.names[hit.chr_id];
      memcpy(o, chr.data(), chr.size()); o += chr.size();
      *o++ = '\t';
      o = put_u64(o, hit.pos0 + 1);
      memcpy(o, "\t255\t", 5); o += 5;
      o = put_u64(o, L);
      memcpy(o, "M\t*\t0\t0\t", 8); o += 8;
      memcpy(o, hit.is_rc ? q.seq.p : rc_seq.data(), L); o += L;
      *o++ = '\t';
      memcpy(o, hit.is_rc ? q.qual.p : rev_qual.data(), QL); o += QL;
      *o++ = '\t';
      o = put_u64(o, (uint64_t)nh);
      *o++ = '\t';
      *o++ = '\n';
    }
    out.n = (size_t)(o - out.p);
  }
  return true;
}

}  // namespace sbwt

using namespace sbwt;

namespace sbwt {

// Newline census of the file in 1 MiB chunks: upstream's reader consumes the file four lines at a time from its
// start (fastq_reader_impl.hpp:81-98), so record k begins right after newline 4k; the census lets every host
// thread find the first record of its share without reading what comes before.
struct LineIndex {
  static constexpr uint64_t kChunk = 1ull << 20;
  const char* base = nullptr;
  uint64_t size = 0;
  std::vector<uint64_t> before;  // newlines before each chunk; last entry = total
  uint64_t total_lines() const { return before.empty() ? 0 : before.back(); }
  static uint64_t count_newlines(const char* a, const char* e) {
    uint64_t cnt = 0;
    for (const char* q = a; q < e; ++q) cnt += *q == '\n';
    return cnt;
  }
  // byte offset right after newline number `k` (k >= 1), or 0 for k == 0
  const char* after_newline(uint64_t k) const {
    if (k == 0) return base;
    // chunk holding newline k: before[c] < k <= before[c + 1]
    uint64_t c = (uint64_t)(std::lower_bound(before.begin(), before.end(), k) - before.begin()) - 1;
    uint64_t seen = before[c];
    const char* q = base + c * kChunk;
    const char* e = base + size;
    while (q < e) {
      const char* nl = static_cast<const char*>(memchr(q, '\n', (size_t)(e - q)));
      if (!nl) return e;
      if (++seen == k) return nl + 1;
      q = nl + 1;
    }
    return e;
  }
};

// page-locked host array (sbwt_host_alloc): the batch buffers cross PCIe as direct DMA
template <typename T>
struct PinnedArr {
  T* p = nullptr;
  uint64_t cap = 0;
  PinnedArr() = default;
  PinnedArr(const PinnedArr&) = delete;
  PinnedArr& operator=(const PinnedArr&) = delete;
  PinnedArr(PinnedArr&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  ~PinnedArr() { sbwt_host_free(p); }
  int ensure(uint64_t count) {  // contents are not kept
    if (count <= cap) return SBWT_OK;
    sbwt_host_free(p);
    p = nullptr;
    cap = 0;
    void* q = nullptr;
    if (int rc = sbwt_host_alloc(count * sizeof(T), &q)) return rc;
    p = static_cast<T*>(q);
    cap = count;
    return SBWT_OK;
  }
  T* data() { return p; }
  T& operator[](uint64_t k) { return p[k]; }
};

// n worker threads that run the shares of one stage after another: run(n_tasks, f) calls f(task) for every task and
// returns when all are done.  Several run() calls may be active at once (the stages of different batches); tasks
// are taken in FIFO order.
class WorkerPool {
 public:
  explicit WorkerPool(int n) {
    for (int t = 0; t < n; ++t) th_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::unique_lock<std::mutex> lk(m_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& t : th_) t.join();
  }
  struct Group {
    std::function<void(int)> f;
    int n = 0, next = 0, done = 0;
  };
  // starts the group; wait(g) blocks until its tasks are done
  std::shared_ptr<Group> start(int n_tasks, std::function<void(int)> f) {
    auto g = std::make_shared<Group>();
    g->f = std::move(f);
    g->n = n_tasks;
    std::unique_lock<std::mutex> lk(m_);
    if (n_tasks > 0) q_.push_back(g);
    cv_.notify_all();
    return g;
  }
  void wait(const std::shared_ptr<Group>& g) {
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return g->done == g->n; });
  }

 private:
  void loop() {
    for (;;) {
      std::shared_ptr<Group> g;
      int task = -1;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        g = q_.front();
        task = g->next++;
        if (g->next == g->n) q_.pop_front();
      }
      g->f(task);
      {
        std::unique_lock<std::mutex> lk(m_);
        if (++g->done == g->n) done_cv_.notify_all();
      }
    }
  }
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  std::deque<std::shared_ptr<Group>> q_;
  std::vector<std::thread> th_;
  bool stop_ = false;
};

// what one device contributes to the pipeline: its index, a ctx, page-locked result buffers per batch slot
struct DeviceLane {
  const sbwt_index* ix = nullptr;
  sbwt_ctx* ctx = nullptr;
};

constexpr int kSlots = 4;  // batches in flight: parse+pack | map | format | write

struct DevOut {  // results of one device for one batch slot
  PinnedArr<sbwt_hit> lists;
  PinnedArr<sbwt_wide_result> wide;
  uint64_t n_lists = 0, n_wide = 0;
};

struct BatchSlot {
  uint64_t r0 = 0, nr = 0, n_ok = 0;  // first record, records, records before the first offending one
  int bad = 0;                        // SBWT_ERR_ILLEGAL_CHAR / _TOO_SHORT of record r0 + n_ok
  uint32_t wpr = 4;
  std::vector<Record> recs;
  PinnedArr<uint32_t> packed, lens;
  PinnedArr<sbwt_compact_result> out;
  std::vector<DevOut> dev;
  std::vector<PartView> parts;
  std::vector<TextBuf> text;          // one per format share, reused from batch to batch
  std::vector<uint64_t> text_off;     // file offset of every share's text
  int rc = SBWT_OK;
  std::string err;
};

}  // namespace sbwt

extern "C" int sbwt_map_fastq_multi(const sbwt_index* const* ixs, int n_ix, const char* fastq_path, const char* sam_path, int limit,
                                    int strand, int n_threads, uint64_t stats[5]) {
  if (!ixs || n_ix < 1 || !fastq_path || !sam_path) {
    set_error("sbwt_map_fastq_multi: null pointer, or no index to map against (n_ix < 1)");
    return SBWT_ERR_ARG;
  }
  for (int d = 0; d < n_ix; ++d) {
    if (!ixs[d]) {
      set_error("sbwt_map_fastq_multi: null index");
      return SBWT_ERR_ARG;
    }
    if (ixs[d]->d.n != ixs[0]->d.n || ixs[d]->chr.names != ixs[0]->chr.names) {
      set_error("sbwt_map_fastq_multi: the indexes are not replicas of one index");
      return SBWT_ERR_ARG;
    }
  }
  if (n_threads < 1) n_threads = 1;
  std::string err;
  MappedFile fq;
  if (!fq.open(fastq_path, err)) {
    // an empty FASTQ maps nothing, it is not an error upstream
    FILE* probe = fopen(fastq_path, "rb");
    if (!probe) {
      set_error(err);
      return SBWT_ERR_IO;
    }
    fclose(probe);
  }
  const bool to_stdout = !strcmp(sam_path, "stdout") || !strcmp(sam_path, "-");  // main.cpp:202
  FILE* out = to_stdout ? stdout : fopen(sam_path, "wb");
  if (!out) {
    set_error(std::string("cannot create output file ") + sam_path);
    return SBWT_ERR_IO;
  }
  // The text of a batch is written by ONE thread of its own, share after share.  Buffered writes to one file serialise
  // in the kernel (the inode lock), so positional writes from all the workers only queue there -- and keep the whole
  // pool away from parsing and formatting meanwhile.  (Measured on the MI355X host, ext4: 1.16 GB of SAM in 0.12-0.18 s
  // either way, i.e. 7-9 GB/s, the floor of this path; a shared mapping of the output is twice as slow, fallocate and
  // page faults included.)  SBWT_SAM_WRITE=pool brings the positional writes by all workers back (measurement).
  struct stat sst;
  const char* write_knob = getenv("SBWT_SAM_WRITE");
  const bool parallel_write = !to_stdout && fstat(fileno(out), &sst) == 0 && S_ISREG(sst.st_mode) && write_knob && !strcmp(write_knob, "pool");
  const bool direct_write = !to_stdout && !parallel_write;  // write(2) on the descriptor, no stdio copy
  const bool timing = getenv("SBWT_TIMING") != nullptr;  // per-stage wall time on stderr
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  uint64_t st[5] = {0, 0, 0, 0, 0};
  int rc = SBWT_OK;
  uint64_t batch = 1u << 18;  // reads per batch and device: small enough that filling and draining the four stages is a tenth of a 8M-read file, large enough that a batch keeps a device busy
  if (const char* e = getenv("SBWT_FASTQ_BATCH")) batch = std::max<uint64_t>(64, strtoull(e, nullptr, 10));
  const uint64_t kBatch = batch * (uint64_t)n_ix;
  std::vector<DeviceLane> lanes((size_t)n_ix);
  for (int d = 0; d < n_ix && rc == SBWT_OK; ++d) {
    lanes[(size_t)d].ix = ixs[d];
    rc = sbwt_ctx_create(ixs[d], batch, &lanes[(size_t)d].ctx);
  }
  if (rc != SBWT_OK) {  // a ctx could not be made: sbwt_ctx_create has set the message; nothing of the file is read
    for (auto& L : lanes)
      if (L.ctx) sbwt_ctx_free(L.ctx);
    if (!to_stdout) fclose(out);
    return rc;
  }
  const char* p = fq.data() ? reinterpret_cast<const char*>(fq.data()) : nullptr;
  const uint64_t fsize = p ? fq.size() : 0;
  double t_stage[5] = {0, 0, 0, 0, 0};  // census, parse+pack, map, format, write (summed over batches; they overlap)
  {
    WorkerPool pool(n_threads);
    // ---- census
    LineIndex lines;
    lines.base = p;
    lines.size = fsize;
    const uint64_t nc = (fsize + LineIndex::kChunk - 1) / LineIndex::kChunk;
    lines.before.assign(nc + 1, 0);
    {
      const double t0 = now();
      const int n_tasks = (int)std::min<uint64_t>(nc, (uint64_t)n_threads * 8);
      auto g = pool.start(n_tasks, [&](int t) {
        for (uint64_t c = (uint64_t)t; c < nc; c += (uint64_t)n_tasks)
          lines.before[c + 1] = LineIndex::count_newlines(p + c * LineIndex::kChunk, p + std::min(fsize, (c + 1) * LineIndex::kChunk));
      });
      pool.wait(g);
      for (uint64_t c = 0; c < nc; ++c) lines.before[c + 1] += lines.before[c];
      t_stage[0] = now() - t0;
    }
    // a record counts only when all four of its lines are newline-terminated (fastq_reader_impl.hpp:94-98)
    const uint64_t n_records = lines.total_lines() / 4;
    const uint64_t n_batches = (n_records + kBatch - 1) / kBatch;
    std::vector<BatchSlot> slots((size_t)kSlots);
    for (auto& S : slots) {
      S.dev.resize((size_t)n_ix);
      S.text.resize((size_t)n_threads);
      S.text_off.assign((size_t)n_threads, 0);
    }
    uint32_t wpr_hint = 4;
    uint64_t file_off = 0;         // bytes of SAM text placed so far
    uint64_t stop_after = ~0ull;   // batch that holds the first offending read: nothing behind it is started
    int final_rc = SBWT_OK;
    std::string final_err;
    std::mutex err_m;
    auto fail = [&](int code, const std::string& msg) {
      std::lock_guard<std::mutex> lk(err_m);
      if (final_rc == SBWT_OK) { final_rc = code; final_err = msg; }
    };

    // ---- the four stages of one batch
    auto stage_parse = [&](BatchSlot& S) {  // parse + pack, n_threads shares
      const double t0 = now();
      const uint64_t nr = S.nr;
      S.recs.resize(nr);
      S.wpr = wpr_hint;
      if (S.packed.ensure(nr * S.wpr) || S.lens.ensure(nr) || S.out.ensure(nr)) { fail(SBWT_ERR_NOMEM, sbwt_last_error()); return; }
      std::vector<uint32_t> max_len((size_t)n_threads, 0);
      std::vector<uint64_t> bad_at((size_t)n_threads, ~0ull);
      std::vector<int> bad_rc((size_t)n_threads, 0);
      auto work = [&](int t, bool pack_only) {
        const uint64_t a = nr * (uint64_t)t / (uint64_t)n_threads, b = nr * (uint64_t)(t + 1) / (uint64_t)n_threads;
        if (a == b) return;
        const char* q = pack_only ? nullptr : lines.after_newline(4 * (S.r0 + a));
        const char* end = p + fsize;
        uint32_t mx = 0;
        const uint32_t wpr = S.wpr;
        for (uint64_t k = a; k < b; ++k) {
          if (!pack_only) next_record(q, end, S.recs[k]);  // cannot fail: the census counted its four newlines
          const Record& R = S.recs[k];
          if (R.seq.n > mx) mx = R.seq.n;
          S.lens[k] = R.seq.n;
          if (R.seq.n > wpr * 16u) continue;  // the batch is packed again with wider records
          // the earliest offending read wins, as the upstream serial loop would hit it first
          const int s = pack_read(R.seq.p, R.seq.n, wpr, &S.packed[k * wpr]);
          if (s != SBWT_OK && bad_at[(size_t)t] == ~0ull) { bad_at[(size_t)t] = k; bad_rc[(size_t)t] = s; }
        }
        max_len[(size_t)t] = mx;
      };
      pool.wait(pool.start(n_threads, [&](int t) { work(t, false); }));
      uint32_t longest = 0;
      for (uint32_t v : max_len) longest = std::max(longest, v);
      const uint32_t need = sbwt_words_per_read(longest);
      if (need > S.wpr) {  // a read longer than the records of the batch before: pack again, wider
        S.wpr = need;
        if (S.packed.ensure(nr * S.wpr)) { fail(SBWT_ERR_NOMEM, sbwt_last_error()); return; }
        std::fill(bad_at.begin(), bad_at.end(), ~0ull);
        pool.wait(pool.start(n_threads, [&](int t) { work(t, true); }));
      }
      wpr_hint = need;
      S.n_ok = nr;
      S.bad = 0;
      for (int t = 0; t < n_threads; ++t)
        if (bad_at[(size_t)t] != ~0ull) { S.n_ok = bad_at[(size_t)t]; S.bad = bad_rc[(size_t)t]; break; }
      t_stage[1] += now() - t0;
    };
    auto stage_map = [&](BatchSlot& S) {  // every device maps its contiguous share of the batch
      const double t0 = now();
      S.parts.assign((size_t)n_ix, PartView{});
      std::vector<std::thread> th;
      std::vector<int> rcs((size_t)n_ix, SBWT_OK);
      std::vector<std::string> errs((size_t)n_ix);
      auto work = [&](int d) {
        uint64_t a = 0, b = 0;
        sbwt_shard_range(S.n_ok, (uint32_t)n_ix, (uint32_t)d, &a, &b);
        DevOut& O = S.dev[(size_t)d];
        PartView& P = S.parts[(size_t)d];
        P.r0 = a; P.r1 = b; P.rec = S.out.data() + a; P.lists = nullptr; P.wide = nullptr; P.n_wide = 0;
        O.n_lists = O.n_wide = 0;
        if (a == b) return;
        if (O.lists.cap == 0 && (O.lists.ensure(1u << 16) || O.wide.ensure(1024))) { rcs[(size_t)d] = SBWT_ERR_NOMEM; errs[(size_t)d] = sbwt_last_error(); return; }
        uint64_t cnt[2] = {0, 0};
        int r = sbwt_map_batch_compact(lanes[(size_t)d].ctx, S.packed.data() + a * S.wpr, S.lens.data() + a, b - a, S.wpr, limit, strand,
                                       SBWT_MAP_HITS_ONLY, S.out.data() + a, O.lists.data(), O.lists.cap, O.wide.data(), O.wide.cap, cnt);
        // lists or wide records did not fit: again with what it asked for (a second time when the first count could not
        // yet include the lists that are longer than their intervals: multi_locate.h, extra_slots)
        for (int tries = 0; r == SBWT_ERR_CAPACITY && tries < 4; ++tries) {
          if (O.lists.ensure(cnt[0] + (cnt[0] >> 2) + 16) || O.wide.ensure(cnt[1] + 16)) { rcs[(size_t)d] = SBWT_ERR_NOMEM; errs[(size_t)d] = sbwt_last_error(); return; }
          r = sbwt_map_batch_compact(lanes[(size_t)d].ctx, S.packed.data() + a * S.wpr, S.lens.data() + a, b - a, S.wpr, limit, strand,
                                     SBWT_MAP_HITS_ONLY, S.out.data() + a, O.lists.data(), O.lists.cap, O.wide.data(), O.wide.cap, cnt);
        }
        if (r != SBWT_OK) { rcs[(size_t)d] = r; errs[(size_t)d] = sbwt_last_error(); return; }
        O.n_lists = cnt[0];
        O.n_wide = cnt[1];
        P.lists = O.lists.data();
        P.wide = O.wide.data();
        P.n_wide = cnt[1];
      };
      for (int d = 1; d < n_ix; ++d) th.emplace_back(work, d);
      work(0);
      for (auto& x : th) x.join();
      for (int d = 0; d < n_ix; ++d)
        if (rcs[(size_t)d] != SBWT_OK) fail(rcs[(size_t)d], errs[(size_t)d]);
      t_stage[2] += now() - t0;
    };
    std::vector<std::vector<uint64_t>> share_stats((size_t)n_threads, std::vector<uint64_t>(5, 0));
    auto stage_format = [&](BatchSlot& S) {
      const double t0 = now();
      std::vector<char> ok((size_t)n_threads, 1);
      pool.wait(pool.start(n_threads, [&](int t) {
        const uint64_t a = S.n_ok * (uint64_t)t / (uint64_t)n_threads, b = S.n_ok * (uint64_t)(t + 1) / (uint64_t)n_threads;
        TextBuf& B = S.text[(size_t)t];
        B.n = 0;
        ok[(size_t)t] = format_range(ixs[0], S.recs.data(), S.parts.data(), n_ix, a, b, B, share_stats[(size_t)t].data());
      }));
      if (std::find(ok.begin(), ok.end(), 0) != ok.end()) fail(SBWT_ERR_NOMEM, "out of host memory while formatting SAM text");
      for (int t = 0; t < n_threads; ++t) {  // batches are formatted in order: so are their places in the file
        S.text_off[(size_t)t] = file_off;
        file_off += S.text[(size_t)t].n;
      }
      t_stage[3] += now() - t0;
    };
    auto stage_write = [&](BatchSlot& S) {
      const double t0 = now();
      if (parallel_write) {
        const int fd = fileno(out);
        pool.wait(pool.start(n_threads, [&](int t) {
          const TextBuf& B = S.text[(size_t)t];
          size_t done = 0;
          while (done < B.n) {
            const ssize_t w = pwrite(fd, B.p + done, B.n - done, (off_t)(S.text_off[(size_t)t] + done));
            if (w <= 0) { fail(SBWT_ERR_IO, "write error on SAM output"); return; }
            done += (size_t)w;
          }
        }));
      } else if (direct_write) {
        const int fd = fileno(out);
        for (int t = 0; t < n_threads; ++t) {
          const TextBuf& B = S.text[(size_t)t];
          size_t done = 0;
          while (done < B.n) {
            const ssize_t w = write(fd, B.p + done, B.n - done);
            if (w <= 0) { fail(SBWT_ERR_IO, "write error on SAM output"); return; }
            done += (size_t)w;
          }
        }
      } else {
        for (int t = 0; t < n_threads; ++t) {
          const TextBuf& B = S.text[(size_t)t];
          if (B.n && fwrite(B.p, 1, B.n, out) != B.n) { fail(SBWT_ERR_IO, "write error on SAM output"); break; }
        }
      }
      t_stage[4] += now() - t0;
    };

    // ---- the pipeline: in every tick batch k is parsed and packed, batch k-1 mapped, batch k-2 formatted and batch
    // k-3 written, all at once (upstream: Job_distributer<M_T>::distribute_jobs, job_distributer.hpp:37-90, reads on
    // the calling thread and hands groups of 1000 reads to its pool; the output order there is the pool's, here the
    // input's)
    for (uint64_t tick = 0; rc == SBWT_OK && tick < n_batches + 3; ++tick) {
      std::vector<std::thread> stage_threads;
      const uint64_t last = std::min<uint64_t>(n_batches, stop_after == ~0ull ? n_batches : stop_after + 1);
      if (tick >= 3 && tick - 3 < last) stage_threads.emplace_back([&, tick] { stage_write(slots[(size_t)((tick - 3) % kSlots)]); });
      if (tick >= 2 && tick - 2 < last) stage_threads.emplace_back([&, tick] { stage_format(slots[(size_t)((tick - 2) % kSlots)]); });
      if (tick >= 1 && tick - 1 < last) stage_threads.emplace_back([&, tick] { stage_map(slots[(size_t)((tick - 1) % kSlots)]); });
      if (tick < last) {
        BatchSlot& S = slots[(size_t)(tick % kSlots)];
        S.r0 = tick * kBatch;
        S.nr = std::min(kBatch, n_records - S.r0);
        stage_parse(S);
        if (S.bad && stop_after == ~0ull) stop_after = tick;  // upstream processes every read before this one, then aborts
      }
      for (auto& t : stage_threads) t.join();
      if (final_rc != SBWT_OK) rc = final_rc;
      if (tick >= 3 && tick - 3 + 1 >= last) break;  // the last batch has been written
    }
    if (final_rc != SBWT_OK) set_error(final_err);
    for (const auto& s5 : share_stats)
      for (int k = 0; k < 5; ++k) st[k] += s5[(size_t)k];
    if (rc == SBWT_OK && stop_after != ~0ull) {
      const BatchSlot& S = slots[(size_t)(stop_after % kSlots)];
      char msg[200];
      snprintf(msg, sizeof msg, "read %llu: %s; upstream sbwt aborts here (searcher.hpp:%d)", (unsigned long long)(S.r0 + S.n_ok),
               S.bad == SBWT_ERR_ILLEGAL_CHAR ? "illegal char (base outside ACGT)" : "shorter than 12 bases",
               S.bad == SBWT_ERR_ILLEGAL_CHAR ? 399 : 404);
      set_error(msg);
      rc = S.bad;
    }
  }
  if (timing)
    fprintf(stderr, "[sbwt timing] %d device(s), %d host threads: census %.3f parse+pack %.3f map %.3f format %.3f write %.3f s (stages overlap), total %.3f s\n",
            n_ix, n_threads, t_stage[0], t_stage[1], t_stage[2], t_stage[3], t_stage[4], now() - t_begin);
  for (auto& L : lanes)
    if (L.ctx) sbwt_ctx_free(L.ctx);
  if (to_stdout) fflush(out);
  else if (fclose(out) != 0 && rc == SBWT_OK) { set_error("write error on SAM output"); rc = SBWT_ERR_IO; }
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}

extern "C" int sbwt_map_fastq(const sbwt_index* ix, const char* fastq_path, const char* sam_path, int limit, int strand,
                              int n_threads, uint64_t stats[5]) {
  if (!ix) return SBWT_ERR_ARG;
  const sbwt_index* one[1] = {ix};
  return sbwt_map_fastq_multi(one, 1, fastq_path, sam_path, limit, strand, n_threads, stats);
}
