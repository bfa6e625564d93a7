// capi.cpp -- extern "C" entry points for index handling and read packing (see include/sbwt_amd.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sbwt_amd.h"
#include "index_impl.hpp"
#include "pack.hpp"

namespace sbwt {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  g_error = buf;
  (void)hipGetLastError();  // clear the sticky error of the runtime
  return e == hipErrorOutOfMemory ? SBWT_ERR_NOMEM : SBWT_ERR_HIP;
}

// temporary device copy of a host array, freed on scope exit
struct TempDev {
  void* p = nullptr;
  ~TempDev() { if (p) hipFree(p); }
  int upload(const void* host, uint64_t bytes) {
    if (bytes == 0) bytes = 8;
    SBWT_HIP(hipMalloc(&p, bytes));
    if (host) SBWT_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    return 0;
  }
};

static int check_device(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_error("no HIP device available: this library has no CPU fallback");
    (void)hipGetLastError();
    return SBWT_ERR_HIP;
  }
  if (device < 0 || device >= count) {
    set_error("device ordinal out of range");
    return SBWT_ERR_ARG;
  }
  hipDeviceProp_t prop;
  SBWT_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error(std::string("device is ") + prop.gcnArchName + ", the kernels are built for gfx950 only");
    return SBWT_ERR_HIP;
  }
  SBWT_HIP(enter_device(device));
  return SBWT_OK;
}

// common tail of the two index constructors: raw host tables -> device layout
static int make_index(int device, uint64_t interval, uint64_t n, uint64_t first_location, uint64_t real_size,
                      const uint64_t c4[4], const void* c_functions, const void* const occ[4], uint64_t occ_len,
                      const void* jbwt, const void* fbwt, const void* loc, uint64_t n_loc, ChrTables&& chr,
                      sbwt_index** out) {
  if (int rc = check_device(device)) return rc;
  // SBWT_TIMING: where the seconds of a load go (stderr): file pages -> device, re-layout + prefix tables
  const bool timing = getenv("SBWT_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::unique_ptr<sbwt_index> ix(new (std::nothrow) sbwt_index);
  if (!ix) return SBWT_ERR_NOMEM;
  ix->device = device;
  ix->chr = std::move(chr);
  if (real_size * 2 + 1 != n) {
    set_error("index tables are inconsistent: n != 2 * sum(chrLen) + 1");
    return SBWT_ERR_FORMAT;
  }
  uint64_t* d_cf = nullptr;
  if (int rc = dev_alloc(ix.get(), &d_cf, kNumKmer + 1)) return rc;
  SBWT_HIP(hipMemcpy(d_cf, c_functions, (kNumKmer + 1) * 8, hipMemcpyHostToDevice));
  TempDev t_jbwt, t_fbwt, t_loc, t_occ[4];
  if (int rc = t_jbwt.upload(jbwt, n / 4 + 1)) return rc;
  if (int rc = t_fbwt.upload(fbwt, n)) return rc;
  if (int rc = t_loc.upload(loc, n_loc * 16)) return rc;
  RawDeviceTables raw;
  raw.interval = interval;
  raw.n = n;
  raw.first_location = first_location;
  raw.real_size = real_size;
  for (int c = 0; c < 4; ++c) raw.c_function4[c] = c4[c];
  raw.c_functions = d_cf;
  raw.jbwt = static_cast<const uint8_t*>(t_jbwt.p);
  raw.fbwt = static_cast<const uint8_t*>(t_fbwt.p);
  raw.loc = static_cast<const uint64_t*>(t_loc.p);
  raw.n_loc = n_loc;
  raw.occ_len = occ_len;
  for (int c = 0; c < 4; ++c) {
    if (!occ[c]) continue;
    if (int rc = t_occ[c].upload(occ[c], occ_len * 8)) return rc;
    raw.occ[c] = static_cast<const uint64_t*>(t_occ[c].p);
  }
  const double t1 = now();
  if (int rc = build_device_layout(ix.get(), raw)) return rc;
  if (int rc = upload_chr_tables(ix.get())) return rc;
  if (int rc = finish_index(ix.get())) return rc;
  if (timing)
    fprintf(stderr, "[sbwt timing] index on device %d: reference tables to the device %.2f s (the file's pages are read here), HBM layout "
            "%.2f s (see SBWT_DEBUG for its parts), %.1f GB held\n", device, t1 - t0, now() - t1, (double)ix->device_bytes / 1e9);
  *out = ix.release();
  return SBWT_OK;
}

}  // namespace sbwt

using namespace sbwt;

extern "C" const char* sbwt_last_error(void) { return g_error.c_str(); }

extern "C" int sbwt_device_count(int* count) {
  if (!count) return SBWT_ERR_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    *count = 0;
    set_error("no HIP device available: this library has no CPU fallback");
    (void)hipGetLastError();
    return SBWT_ERR_HIP;
  }
  *count = n;
  return SBWT_OK;
}

extern "C" int sbwt_index_load(const char* prefix_in, int device, sbwt_index** out) {
  if (!prefix_in || !out) return SBWT_ERR_ARG;
  *out = nullptr;
  std::string prefix(prefix_in);
  if (prefix.empty() || prefix.back() != '.') prefix += '.';  // searcher.hpp:196-198
  std::string err;
  MappedFile f;
  if (!f.open(prefix + "t_table.bwt", err)) {
    set_error(err);
    return SBWT_ERR_IO;
  }
  TableView v;
  if (!parse_table(f, v, err)) {
    set_error(err);
    return SBWT_ERR_FORMAT;
  }
  ChrTables chr;
  if (!read_chr_tables(prefix, chr, err)) {
    set_error(err);
    return SBWT_ERR_FORMAT;
  }
  uint64_t c4[4];
  static const int kAcgt[4] = {'A', 'C', 'G', 'T'};
  for (int c = 0; c < 4; ++c) memcpy(&c4[c], v.c_function + 8 * kAcgt[c], 8);
  const void* occ[4] = {v.occ[0], v.occ[1], v.occ[2], v.occ[3]};
  const uint64_t real_size = chr.real_size;
  return make_index(device, v.interval, v.n, v.first_location, real_size, c4, v.c_functions, occ, v.occ_len, v.jbwt,
                    v.fbwt, v.loc, v.n_loc, std::move(chr), out);
}

extern "C" int sbwt_index_from_tables(int device, uint64_t interval, uint64_t n, uint64_t first_location,
                                      uint64_t real_size, const uint64_t* c_function4, const uint64_t* c_functions,
                                      const uint64_t* occA, const uint64_t* occC, const uint64_t* occG,
                                      const uint64_t* occT, const uint8_t* jbwt_seq, const uint8_t* fbwt,
                                      const uint64_t* location_table, uint64_t n_loc, uint64_t n_chr,
                                      const uint64_t* chr_start, const char* const* chr_names, uint64_t n_npos,
                                      const uint64_t* npos_key, const uint64_t* npos_val, sbwt_index** out) {
  if (!out || !c_function4 || !c_functions || !jbwt_seq || !fbwt || (n_loc && !location_table) || !n_chr ||
      !chr_start || !n_npos || !npos_key || !npos_val) {
    set_error("sbwt_index_from_tables: null pointer");
    return SBWT_ERR_ARG;
  }
  *out = nullptr;
  if (interval < 2 || interval > 1024 || (interval & (interval - 1)) || n < 2 || first_location >= n) {
    set_error("sbwt_index_from_tables: interval must be a power of two in 2..1024 and first_location < n");
    return SBWT_ERR_ARG;
  }
  ChrTables chr;
  for (uint64_t k = 0; k < n_chr; ++k) {
    chr.start.push_back(chr_start[k]);
    chr.names.push_back(chr_names && chr_names[k] ? std::string(chr_names[k]) : "chr" + std::to_string(k + 1));
  }
  chr.real_size = real_size;
  chr.npos_key.assign(npos_key, npos_key + n_npos);
  chr.npos_val.assign(npos_val, npos_val + n_npos);
  if (chr.start[0] != 0 || chr.npos_key[0] != 0) {
    set_error("sbwt_index_from_tables: chr_start[0] and npos_key[0] must be 0");
    return SBWT_ERR_ARG;
  }
  const void* occ[4] = {occA, occC, occG, occT};
  return make_index(device, interval, n, first_location, real_size, c_function4, c_functions, occ, n / interval + 1,
                    jbwt_seq, fbwt, location_table, n_loc, std::move(chr), out);
}

extern "C" void sbwt_index_free(sbwt_index* ix) { delete ix; }

extern "C" int sbwt_index_get_info(const sbwt_index* ix, sbwt_index_info* info) {
  if (!ix || !info) return SBWT_ERR_ARG;
  info->n = ix->d.n;
  info->real_size = ix->d.real_size;
  info->interval = ix->d.interval;
  info->first_location = ix->d.first_location;
  info->n_chr = ix->chr.start.size();
  info->n_npos = ix->chr.npos_key.size();
  info->n_loc = ix->n_loc;
  info->device_bytes = ix->device_bytes;
  info->device = ix->device;
  info->kmer_order = ix->d.kmer_order;
  info->hash_order = ix->d.hash_order;
  info->hash_text = ix->d.hash_text;
  return SBWT_OK;
}

extern "C" const char* sbwt_index_chr_name(const sbwt_index* ix, uint64_t i) {
  if (!ix || i >= ix->chr.names.size()) return nullptr;
  return ix->chr.names[i].c_str();
}

extern "C" int sbwt_shard_range(uint64_t n_items, uint32_t parts, uint32_t k, uint64_t* begin, uint64_t* end) {
  if (parts == 0 || k >= parts || !begin || !end) {
    set_error("sbwt_shard_range: bad argument");
    return SBWT_ERR_ARG;
  }
  const uint64_t base = n_items / parts, extra = n_items % parts;
  *begin = (uint64_t)k * base + ((uint64_t)k < extra ? (uint64_t)k : extra);
  *end = *begin + base + ((uint64_t)k < extra ? 1 : 0);
  return SBWT_OK;
}

extern "C" uint32_t sbwt_words_per_read(uint32_t max_len) {
  uint32_t w = (max_len + 15) / 16;
  w = (w + 3) & ~3u;
  return w ? w : 4;
}

extern "C" int sbwt_pack_reads(const char* seqs, const uint64_t* offsets, uint64_t n_reads, uint32_t wpr,
                               uint32_t* packed, uint32_t* lens, uint64_t* bad_read) {
  if ((n_reads && (!seqs || !offsets || !packed || !lens)) || wpr == 0 || (wpr & 3u)) {
    set_error("sbwt_pack_reads: bad argument");
    return SBWT_ERR_ARG;
  }
  // the first offending read in input order decides the error, as the upstream serial reader would meet it first
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 16) nt = 16;
  if (n_reads < 4096) nt = 1;
  std::vector<uint64_t> first_bad(nt, ~0ull);
  std::vector<int> bad_code(nt, 0);
  auto work = [&](unsigned t) {
    const uint64_t r0 = n_reads * t / nt, r1 = n_reads * (t + 1) / nt;
    for (uint64_t r = r0; r < r1; ++r) {
      const char* s = seqs + offsets[r];
      const uint64_t len = offsets[r + 1] - offsets[r];
      const int bad = pack_read(s, len, wpr, packed + r * wpr);
      lens[r] = (uint32_t)len;
      if (bad && first_bad[t] == ~0ull) {
        first_bad[t] = r;
        bad_code[t] = bad;
      }
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t)
    if (first_bad[t] != ~0ull) {
      if (bad_read) *bad_read = first_bad[t];
      char msg[160];
      snprintf(msg, sizeof msg, "read %llu: %s", (unsigned long long)first_bad[t],
               bad_code[t] == SBWT_ERR_ILLEGAL_CHAR ? "illegal char (upstream aborts on non-ACGT bases)"
               : bad_code[t] == SBWT_ERR_TOO_SHORT  ? "shorter than 12 bases (upstream aborts)"
                                                    : "longer than words_per_read allows");
      set_error(msg);
      return bad_code[t];
    }
  return SBWT_OK;
}
