// host_pipeline.cpp -- FASTQ -> SAM driver above the batch mapper.
// Stands in for Searcher::search (searcher.hpp:336-373), Job_distributer<M_T>::distribute_jobs
// (job_distributer.hpp:37-90), the FASTQ reader (fastq_reader_impl.hpp:67-111) and the SAM text of
// position2sam / Sam::operator<< (searcher.hpp:540-598, sam.hpp:131-140, tuple_utility.hpp:54-64).
// Records are written in input order (what upstream produces with -n 1; with -n > 1 upstream emits the
// same lines with 1000-read groups in arbitrary order).
#include <ctype.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
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

static inline void put_u64(std::string& s, uint64_t v) {
  char t[24];
  int k = 0;
  do { t[k++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (k) s.push_back(t[--k]);
}

static void format_range(const sbwt_index* ix, const Record* recs, const sbwt_read_result* res, const sbwt_hit* hits,
                         uint64_t r0, uint64_t r1, std::string& out) {
  std::string rc_seq, rev_qual;
  for (uint64_t r = r0; r < r1; ++r) {
    const int32_t nh = res[r].nh;
    if (nh <= 0) continue;
    const Record& q = recs[r];
    const uint32_t L = q.seq.n;
    rc_seq.resize(L);
    for (uint32_t k = 0; k < L; ++k) {
      const char c = q.seq.p[L - 1 - k];
      rc_seq[k] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
    }
    rev_qual.assign(q.qual.p, q.qual.n);
    for (uint32_t a = 0, b = q.qual.n; a + 1 < b; ++a, --b) std::swap(rev_qual[a], rev_qual[b - 1]);
    for (int32_t h = 0; h < nh; ++h) {
      const sbwt_hit& H = hits[res[r].first_hit + (uint64_t)h];
      // every field, the last included, is followed by a tab (tuple_utility.hpp:54-64), then '\n'
      if (q.name.n > 1) out.append(q.name.p + 1, q.name.n - 1);  // Fastq::getName: substr(1)
      out.push_back('\t');
      out.append(H.is_rc ? "0" : "16");  // searcher.hpp:551-593
      out.push_back('\t');
      out.append(ix->chr.names[H.chr_id]);
      out.push_back('\t');
      put_u64(out, H.pos0 + 1);
      out.append("\t255\t");
      put_u64(out, L);
      out.append("M\t*\t0\t0\t");
      if (H.is_rc) out.append(q.seq.p, L); else out.append(rc_seq);
      out.push_back('\t');
      if (H.is_rc) out.append(q.qual.p, q.qual.n); else out.append(rev_qual);
      out.push_back('\t');
      put_u64(out, (uint64_t)nh);
      out.append("\t\n");
    }
  }
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

// SAM text leaves through one writer thread, in order, while the next batch is parsed and mapped
class AsyncWriter {
 public:
  explicit AsyncWriter(FILE* f) : f_(f), th_([this] { run(); }) {}
  ~AsyncWriter() { finish(); }
  void push(std::vector<std::string>&& bufs) {
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
      std::vector<std::string> bufs;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !q_.empty() || done_; });
        if (q_.empty()) return;
        bufs = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
      }
      for (auto& b : bufs)
        if (!b.empty() && fwrite(b.data(), 1, b.size(), f_) != b.size()) ok_ = false;
    }
  }
  FILE* f_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::vector<std::string>> q_;
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
  const uint64_t kBatch = 4u << 20;
  sbwt_ctx* ctx = nullptr;
  int rc = sbwt_ctx_create(ix, kBatch, &ctx);
  const char* p = fq.data() ? reinterpret_cast<const char*>(fq.data()) : nullptr;
  LineIndex lines;
  if (p) lines.build(p, fq.size(), n_threads);
  // a record counts only when all four of its lines are newline-terminated (fastq_reader_impl.hpp:94-98)
  const uint64_t n_records = lines.total_lines() / 4;
  std::vector<Record> recs;
  std::vector<uint32_t> packed, lens;
  std::vector<sbwt_read_result> res;
  std::vector<sbwt_hit> hits;
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
      uint32_t longest = 0;
      for (uint32_t v : max_len) longest = std::max(longest, v);
      const uint32_t wpr = sbwt_words_per_read(longest);
      packed.resize(nr * wpr);
      lens.resize(nr);
      res.resize(nr);
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
      uint64_t n_ok = nr;
      int bad = 0;
      for (int t = 0; t < n_threads; ++t)
        if (bad_at[(size_t)t] != ~0ull) { n_ok = bad_at[(size_t)t]; bad = bad_rc[(size_t)t]; break; }
      // upstream processes (and prints) every read before the offending one, then aborts
      uint64_t n_hits = 0;
      if (n_ok) {
        if (hits.size() < n_ok * 2) hits.resize(n_ok * 2);
        rc = sbwt_map_batch(ctx, packed.data(), lens.data(), n_ok, wpr, limit, strand, res.data(), hits.data(),
                            hits.size(), &n_hits);
        if (rc == SBWT_ERR_CAPACITY) {
          hits.resize(n_hits + 16);
          rc = sbwt_map_batch(ctx, packed.data(), lens.data(), n_ok, wpr, limit, strand, res.data(), hits.data(),
                              hits.size(), &n_hits);
        }
        if (rc != SBWT_OK) break;
        std::vector<std::string> bufs((size_t)n_threads);
        auto fmt_work = [&](int t) {
          const uint64_t a = n_ok * (uint64_t)t / (uint64_t)n_threads, b = n_ok * (uint64_t)(t + 1) / (uint64_t)n_threads;
          format_range(ix, recs.data(), res.data(), hits.data(), a, b, bufs[(size_t)t]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(fmt_work, t);
        fmt_work(0);
        for (auto& x : th) x.join();
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
  }
  if (ctx) sbwt_ctx_free(ctx);
  if (!write_ok && rc == SBWT_OK) { set_error("write error on SAM output"); rc = SBWT_ERR_IO; }
  if (to_stdout) fflush(out); else if (fclose(out) != 0 && rc == SBWT_OK) { set_error("write error on SAM output"); rc = SBWT_ERR_IO; }
  if (stats) memcpy(stats, st, sizeof st);
  return rc;
}
