// index_file.cpp -- see index_file.hpp
#include "index_file.hpp"

#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <fstream>
#include <sstream>

#include "device_layout.h"

namespace sbwt {

// "serialization::archive", library version 10, little-endian 64-bit build
static const unsigned char kArchiveHeader[40] = {22,  0,   0,   0,   0,   0,   0,   0,   's', 'e', 'r', 'i', 'a', 'l',
                                                 'i', 'z', 'a', 't', 'i', 'o', 'n', ':', ':', 'a', 'r', 'c', 'h', 'i',
                                                 'v', 'e', 10,  0,   4,   8,   4,   8,   1,   0,   0,   0};

MappedFile::~MappedFile() {
  if (data_ && size_) munmap(const_cast<uint8_t*>(data_), size_);
}

bool MappedFile::open(const std::string& path, std::string& err) {
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    err = "cannot open " + path;
    return false;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) {
    ::close(fd);
    err = "cannot stat " + path;
    return false;
  }
  void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (p == MAP_FAILED) {
    err = "cannot map " + path;
    return false;
  }
  madvise(p, (size_t)st.st_size, MADV_SEQUENTIAL);
  data_ = static_cast<const uint8_t*>(p);
  size_ = (uint64_t)st.st_size;
  return true;
}

namespace {
struct Cursor {
  const uint8_t* p;
  uint64_t left;
  bool bad = false;
  uint64_t u64() {
    uint64_t v = 0;
    if (left < 8) { bad = true; return 0; }
    memcpy(&v, p, 8);
    p += 8; left -= 8;
    return v;
  }
  const uint8_t* take(uint64_t nbytes) {
    if (bad || left < nbytes) { bad = true; return nullptr; }
    const uint8_t* r = p;
    p += nbytes; left -= nbytes;
    return r;
  }
  // count * elem bytes with overflow guard
  const uint8_t* take_array(uint64_t count, uint64_t elem) {
    if (elem && count > left / elem) { bad = true; return nullptr; }
    return take(count * elem);
  }
};
int acgt_row(int v) { return v == 'A' ? 0 : v == 'C' ? 1 : v == 'G' ? 2 : v == 'T' ? 3 : -1; }
}  // namespace

bool parse_table(const MappedFile& f, TableView& v, std::string& err) {
  Cursor c{f.data(), f.size()};
  const uint8_t* h = c.take(40);
  if (!h || memcmp(h, kArchiveHeader, 40) != 0) {
    err = "t_table.bwt: not a Boost binary archive (library version 10)";
    return false;
  }
  v.interval = c.u64();
  if (c.u64() != 256) c.bad = true;
  v.c_function = c.take(256 * 8);
  if (c.u64() != kNumKmer + 1) c.bad = true;
  v.c_functions = c.take((kNumKmer + 1) * 8);
  c.take(5);                        // class info of vector<vector<u64>>
  if (c.u64() != 256) c.bad = true; // occ_function rows
  c.take(4);                        // item_version
  for (int row = 0; row < 256 && !c.bad; ++row) {
    uint64_t len = c.u64();
    const uint8_t* p = c.take_array(len, 8);
    int k = acgt_row(row);
    if (k >= 0) {
      if (k > 0 && len != v.occ_len) c.bad = true;
      v.occ_len = len;
      v.occ[k] = p;
    }
  }
  c.take(5);  // class info of vector<pair<u64,u64>>
  v.n_loc = c.u64();
  v.loc = c.take_array(v.n_loc, 16);
  v.n = c.u64();
  v.fbwt = c.take(v.n);
  v.first_location = c.u64();
  c.take(5);                        // class info of vector<vector<vector<u64>>>
  if (c.u64() != 256) c.bad = true; // occ_jbwt: rebuilt from its definition, not read
  c.take(4);
  for (int row = 0; row < 256 && !c.bad; ++row) {
    if (c.u64() != 5) c.bad = true;
    c.take(4);
    for (int k = 0; k < 5 && !c.bad; ++k) {
      if (c.u64() != 256) c.bad = true;
      c.take(256 * 8);
    }
  }
  v.jbwt_len = c.u64();
  v.jbwt = c.take(v.jbwt_len);
  if (c.u64() != 1024) c.bad = true;  // jbwt_idx_char
  c.take(1024);
  if (c.bad || c.left != 0) {
    err = "t_table.bwt: truncated or unexpected layout";
    return false;
  }
  // set_interval (aligner_table.hpp:198) only knows the powers of two 2..1024
  if (v.interval < 2 || v.interval > 1024 || (v.interval & (v.interval - 1))) {
    err = "t_table.bwt: interval is not a power of two in 2..1024";
    return false;
  }
  if (v.n < 2 || v.jbwt_len != v.n / 4 + 1 || v.occ_len != v.n / v.interval + 1 || v.first_location >= v.n) {
    err = "t_table.bwt: table sizes are inconsistent";
    return false;
  }
  return true;
}

static bool read_name_value(const std::string& path, std::vector<std::string>* names, std::vector<uint64_t>& vals,
                            std::string& err) {
  std::ifstream in(path);
  if (!in) {
    err = "cannot open " + path;
    return false;
  }
  std::string line;
  while (std::getline(in, line)) {
    size_t tab = line.rfind('\t');
    if (tab == std::string::npos) continue;
    if (names) names->push_back(line.substr(0, tab));
    vals.push_back(strtoull(line.c_str() + tab + 1, nullptr, 10));
  }
  return true;
}

bool read_chr_tables(const std::string& prefix, ChrTables& t, std::string& err) {
  if (!read_name_value(prefix + "chrStart", &t.names, t.start, err)) return false;
  std::vector<uint64_t> lens;
  if (!read_name_value(prefix + "chrLen", nullptr, lens, err)) return false;
  t.real_size = 0;
  for (uint64_t l : lens) t.real_size += l;
  if (t.names.empty() || t.start[0] != 0) {
    err = prefix + "chrStart: empty or does not start at 0";
    return false;
  }
  for (size_t k = 1; k < t.start.size(); ++k)
    if (t.start[k] < t.start[k - 1]) {
      err = prefix + "chrStart: not ascending";
      return false;
    }
  MappedFile f;
  if (!f.open(prefix + "NposLen.z", err)) return false;
  Cursor c{f.data(), f.size()};
  const uint8_t* h = c.take(40);
  if (!h || memcmp(h, kArchiveHeader, 40) != 0) c.bad = true;
  c.take(5);  // class info of std::map
  uint64_t cnt = c.u64();
  c.take(4);  // item_version
  c.take(5);  // class info of std::pair
  if (cnt > c.left / 16) c.bad = true;
  for (uint64_t k = 0; k < cnt && !c.bad; ++k) {
    t.npos_key.push_back(c.u64());
    t.npos_val.push_back(c.u64());
  }
  if (c.bad || c.left != 0 || t.npos_key.empty() || t.npos_key[0] != 0) {
    err = prefix + "NposLen.z: unexpected layout";
    return false;
  }
  return true;
}

TableWriter::~TableWriter() {
  if (f_) fclose(static_cast<FILE*>(f_));
}
bool TableWriter::open(const std::string& path, std::string& err) {
  path_ = path;
  f_ = fopen(path.c_str(), "wb");
  if (!f_) {
    err = "cannot write " + path;
    return false;
  }
  setvbuf(static_cast<FILE*>(f_), nullptr, _IOFBF, 8 << 20);
  return true;
}
void TableWriter::bytes(const void* p, uint64_t n) {
  if (n && fwrite(p, 1, n, static_cast<FILE*>(f_)) != n) bad_ = true;
}
void TableWriter::u64(uint64_t v) { bytes(&v, 8); }
void TableWriter::u32(uint32_t v) { bytes(&v, 4); }
void TableWriter::zeros(int n) {
  static const char z[16] = {0};
  bytes(z, (uint64_t)n);
}
void TableWriter::header() { bytes(kArchiveHeader, 40); }
bool TableWriter::close(std::string& err) {
  FILE* f = static_cast<FILE*>(f_);
  f_ = nullptr;
  if (fclose(f) != 0) bad_ = true;
  if (bad_) err = "write error on " + path_;
  return !bad_;
}

bool write_npos(const std::string& path, const std::vector<uint64_t>& key, const std::vector<uint64_t>& val,
                std::string& err) {
  TableWriter w;
  if (!w.open(path, err)) return false;
  w.header();
  w.zeros(5);
  w.u64(key.size());
  w.u32(0);
  w.zeros(5);
  for (size_t k = 0; k < key.size(); ++k) {
    w.u64(key[k]);
    w.u64(val[k]);
  }
  return w.close(err);
}

}  // namespace sbwt
