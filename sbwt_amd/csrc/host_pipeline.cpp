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

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
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

struct RevComp {
  char t[256];
  RevComp() {
    memset(t, 'A', sizeof t);  // bytes outside ACGT never reach the formatter (pack_read rejects them)
    t[(unsigned char)'A'] = 'T'; t[(unsigned char)'C'] = 'G'; t[(unsigned char)'G'] = 'C'; t[(unsigned char)'T'] = 'A';
  }
};

static bool format_range(const sbwt_index* ix, const Record* recs, const sbwt_read_result* res, const sbwt_hit* hits,
                         uint64_t r0, uint64_t r1, TextBuf& out) {
  static const RevComp rc;
  size_t longest_chr = 0;
  for (const auto& nm : ix->chr.names) longest_chr = std::max(longest_chr, nm.size());
  std::vector<char> rc_seq, rev_qual;
  for (uint64_t r = r0; r < r1; ++r) {
    const int32_t nh = res[r].nh;
    if (nh <= 0) continue;
    const Record& q = recs[r];
    const uint32_t L = q.seq.n, QL = q.qual.n;
    bool have_rev = false;
    const size_t line_max = (size_t)q.name.n + longest_chr + L + QL + 96;
    for (int32_t h = 0; h < nh; ++h) {
      const sbwt_hit& H = hits[res[r].first_hit + (uint64_t)h];
      if (!H.is_rc && !have_rev) {  // searcher.hpp:551-571: SEQ reverse-complemented, QUAL reversed
        if (rc_seq.size() < L) rc_seq.resize(L);
        if (rev_qual.size() < QL) rev_qual.resize(QL);
        for (uint32_t k = 0; k < L; ++k) rc_seq[k] = rc.t[(unsigned char)q.seq.p[L - 1 - k]];
        for (uint32_t k = 0; k < QL; ++k) rev_qual[k] = q.qual.p[QL - 1 - k];
        have_rev = true;
      }
      if (!out.room(line_max)) return false;
      char* o = out.p + out.n;
      // every field, the last included, is followed by a tab (tuple_utility.hpp:54-64), then '\n'
      if (q.name.n > 1) { memcpy(o, q.name.p + 1, q.name.n - 1); o += q.name.n - 1; }  // Fastq::getName: substr(1)
      *o++ = '\t';
      if (H.is_rc) *o++ = '0'; else { *o++ = '1'; *o++ = '6'; }  // searcher.hpp:551-593
      *o++ = '\t';
      const std::string& chr = ix->chr.names[H.chr_id];
      memcpy(o, chr.data(), chr.size()); o += chr.size();
      *o++ = '\t';
      o = put_u64(o, H.pos0 + 1);
      memcpy(o, "\t255\t", 5); o += 5;
      o = put_u64(o, L);
      memcpy(o, "M\t*\t0\t0\t", 8); o += 8;
      memcpy(o, H.is_rc ? q.seq.p : rc_seq.data(), L); o += L;
      *o++ = '\t';
      memcpy(o, H.is_rc ? q.qual.p : rev_qual.data(), QL); o += QL;
      *o++ = '\t';
      o = put_u64(o, (uint64_t)nh);
      *o++ = '\t';
      *o++ = '\n';
      out.n = (size_t)(o - out.p);
    }
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
  void build(const char* p, uint64_t n, int n_threads) {
    base = p;
    size = n;
    const uint64_t nc = (n + kChunk - 1) / kChunk;
    before.assign(nc + 1, 0);
    auto work = [&](int t) {
      for (uint64_t c = (uint64_t)t; c < nc; c += (uint64_t)n_threads) {
        const char* a = p + c * kChunk;
        const char* e = p + std::min(n, (c + 1) * kChunk);
        uint64_t cnt = 0;
        for (const char* q = a; q < e; ++q) cnt += *q == '\n';
        before[c + 1] = cnt;
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (uint64_t c = 0; c < nc; ++c) before[c + 1] += before[c];
  }
  uint64_t total_lines() const { return before.empty() ? 0 : before.back(); }
  // byte offset right after newline number `k` (k >= 1), or 0 for k == 0
  const char* after_newline(uint64_t k) const {
    if (k == 0) return base;
    // chunk holding newline k: before[c] < k <= before[c + 1]
    uint64_t c = (uint64_t)(std::lower_bound(before.begin(), before.end(), k) - before.begin()) - 1;
    uint64_t seen = before[c];
    const char* q = base + c * kChunk;
    const char* e = base + size;
    for (; q < e; ++q)
      if (*q == '\n' && ++seen == k) return q + 1;
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

// SAM text leaves through one writer thread, in order, while the next batch is parsed and mapped
class AsyncWriter {
 public:
  explicit AsyncWriter(FILE* f) : f_(f), th_([this] { run(); }) {}
  ~AsyncWriter() { finish(); }
  void push(std::vector<TextBuf>&& bufs) {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [this] { return q_.size() < 2; });  // at most two batches of text in flight
    q_.push_back(std::move(bufs));
    cv_.notify_all();
  }
  bool finish() {
    {
      std::unique_lock<std::mutex> lk(m_);
      if (done_) return ok_;
      done_ = true;
      cv_.notify_all();
    }
    th_.join();
    return ok_;
  }

 private:
  void run() {
    for (;;) {
      std::vector<TextBuf> bufs;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        bufs = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
      }
      for (auto& b : bufs)
        if (b.n && fwrite(b.p, 1, b.n, f_) != b.n) ok_ = false;
    }
  }
  FILE* f_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::vector<TextBuf>> q_;
  bool done_ = false;
  bool ok_ = true;
  std::thread th_;
};

}  // namespace sbwt

extern "C" int sbwt_map_fastq(const sbwt_index* ix, const char* fastq_path, const char* sam_path, int limit,
                              int strand, int n_threads, uint64_t stats[5]) {
  if (!ix || !fastq_path || !sam_path) return SBWT_ERR_ARG;
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
  uint64_t st[5] = {0, 0, 0, 0, 0};
  const bool timing = getenv("SBWT_TIMING") != nullptr;  // per-phase wall time on stderr
  double t_phase[6] = {0, 0, 0, 0, 0, 0};                // census, parse, pack, map, format, wait for the writer
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int k) { const double t = now(); t_phase[k] += t - t_mark; t_mark = t; };
  const uint64_t kBatch = 4u << 20;
  sbwt_ctx* ctx = nullptr;
  int rc = sbwt_ctx_create(ix, kBatch, &ctx);
  const char* p = fq.data() ? reinterpret_cast<const char*>(fq.data()) : nullptr;
  LineIndex lines;
  if (p) lines.build(p, fq.size(), n_threads);
  lap(0);
  // a record counts only when all four of its lines are newline-terminated (fastq_reader_impl.hpp:94-98)
  const uint64_t n_records = lines.total_lines() / 4;
  std::vector<Record> recs;
  PinnedArr<uint32_t> packed, lens;
  PinnedArr<sbwt_read_result> res;
  PinnedArr<sbwt_hit> hits;
  bool write_ok = true;
  {
    AsyncWriter writer(out);
    for (uint64_t r0 = 0; rc == SBWT_OK && r0 < n_records; r0 += kBatch) {
      const uint64_t nr = std::min(kBatch, n_records - r0);
      recs.resize(nr);
      // parse: every thread finds the first record of its share through the census and reads on from there
      std::vector<uint32_t> max_len((size_t)n_threads, 0);
      auto parse_work = [&](int t) {
        const uint64_t a = nr * (uint64_t)t / (uint64_t)n_threads, b = nr * (uint64_t)(t + 1) / (uint64_t)n_threads;
        if (a == b) return;
        const char* q = lines.after_newline(4 * (r0 + a));
        const char* end = p + fq.size();
        uint32_t mx = 0;
        for (uint64_t k = a; k < b; ++k) {
          next_record(q, end, recs[k]);  // cannot fail: the census counted its four newlines
          if (recs[k].seq.n > mx) mx = recs[k].seq.n;
        }
        max_len[(size_t)t] = mx;
      };
      {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(parse_work, t);
        parse_work(0);
        for (auto& x : th) x.join();
      }
      lap(1);
      uint32_t longest = 0;
      for (uint32_t v : max_len) longest = std::max(longest, v);
      const uint32_t wpr = sbwt_words_per_read(longest);
      if (rc == SBWT_OK) rc = packed.ensure(nr * wpr);
      if (rc == SBWT_OK) rc = lens.ensure(nr);
      if (rc == SBWT_OK) rc = res.ensure(nr);
      if (rc != SBWT_OK) break;
      // pack on n_threads; the earliest offending read wins, as the upstream serial loop would hit it first
      std::vector<uint64_t> bad_at((size_t)n_threads, ~0ull);
      std::vector<int> bad_rc((size_t)n_threads, 0);
      auto pack_work = [&](int t) {
        const uint64_t a = nr * (uint64_t)t / (uint64_t)n_threads, b = nr * (uint64_t)(t + 1) / (uint64_t)n_threads;
        for (uint64_t k = a; k < b; ++k) {
          const int s = pack_read(recs[k].seq.p, recs[k].seq.n, wpr, &packed[k * wpr]);
          lens[k] = recs[k].seq.n;
          if (s != SBWT_OK && bad_at[(size_t)t] == ~0ull) { bad_at[(size_t)t] = k; bad_rc[(size_t)t] = s; }
        }
      };
      {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(pack_work, t);
        pack_work(0);
        for (auto& x : th) x.join();
      }
      lap(2);
      uint64_t n_ok = nr;
      int bad = 0;
      for (int t = 0; t < n_threads; ++t)
        if (bad_at[(size_t)t] != ~0ull) { n_ok = bad_at[(size_t)t]; bad = bad_rc[(size_t)t]; break; }
      // upstream processes (and prints) every read before the offending one, then aborts
      uint64_t n_hits = 0;
      if (n_ok) {
        if ((rc = hits.ensure(n_ok * 2)) != SBWT_OK) break;
        rc = sbwt_map_batch(ctx, packed.data(), lens.data(), n_ok, wpr, limit, strand, res.data(), hits.data(),
                            hits.cap, &n_hits);
        if (rc == SBWT_ERR_CAPACITY) {
          if ((rc = hits.ensure(n_hits + 16)) != SBWT_OK) break;
          rc = sbwt_map_batch(ctx, packed.data(), lens.data(), n_ok, wpr, limit, strand, res.data(), hits.data(),
                              hits.cap, &n_hits);
        }
        if (rc != SBWT_OK) break;
        lap(3);
        std::vector<TextBuf> bufs((size_t)n_threads);
        std::vector<char> fmt_ok((size_t)n_threads, 1);
        auto fmt_work = [&](int t) {
          const uint64_t a = n_ok * (uint64_t)t / (uint64_t)n_threads, b = n_ok * (uint64_t)(t + 1) / (uint64_t)n_threads;
          // one SAM line of a 50-bp read is about 150 bytes
          bufs[(size_t)t].room((size_t)(b - a) * 160 + 4096);
          fmt_ok[(size_t)t] = format_range(ix, recs.data(), res.data(), hits.data(), a, b, bufs[(size_t)t]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(fmt_work, t);
        fmt_work(0);
        for (auto& x : th) x.join();
        lap(4);
        if (std::find(fmt_ok.begin(), fmt_ok.end(), 0) != fmt_ok.end()) {
          set_error("out of host memory while formatting SAM text");
          rc = SBWT_ERR_NOMEM;
          break;
        }
        writer.push(std::move(bufs));
        for (uint64_t k = 0; k < n_ok; ++k) {  // searcher.hpp:353-370
          const int32_t nh = res[k].nh;
          st[0]++;
          st[4] += (uint64_t)(nh > 0 ? nh : 0);
          if (nh == 1) { st[1]++; st[2]++; } else if (nh > 1) { st[1]++; st[3]++; }
        }
      }
      if (bad) {
        char msg[200];
        snprintf(msg, sizeof msg, "read %llu: %s; upstream sbwt aborts here (searcher.hpp:%d)",
                 (unsigned long long)(r0 + n_ok),
                 bad == SBWT_ERR_ILLEGAL_CHAR ? "illegal char (base outside ACGT)" : "shorter than 12 bases",
                 bad == SBWT_ERR_ILLEGAL_CHAR ? 399 : 404);
        set_error(msg);
        rc = bad;
        break;
      }
    }
    write_ok = writer.finish();
    lap(5);
  }
  if (timing)
    fprintf(stderr, "[sbwt timing] census %.3f parse %.3f pack %.3f map %.3f format %.3f writer+stats %.3f s\n", t_phase[0],
            t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
  if (ctx) sbwt_ctx_free(ctx);
  if (!write_ok && rc == SBWT_OK) { set_error("write error on SAM output"); rc = SBWT_ERR_IO; }
  if (to_stdout) fflush(out); else if (fclose(out) != 0 && rc == SBWT_OK) { set_error("write error on SAM output"); rc = SBWT_ERR_IO; }
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}
