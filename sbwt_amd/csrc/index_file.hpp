// index_file.hpp -- reader/writer for the on-disk index `sbwt build` produces (format: SURVEY.md 8b).
// Upstream: Aligner_table::readTable / saveTable (aligner_table.hpp:213-221, bodies binary-only),
// readNPosLen / readChrStartPos / readChrLen (aligner_table.hpp:194-197),
// Genome_pre_handler::run_handler writers (genome_pre_handler.hpp:183-236).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace sbwt {

// read-only memory map of a whole file
class MappedFile {
 public:
  MappedFile() = default;
  ~MappedFile();
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
  bool open(const std::string& path, std::string& err);
  const uint8_t* data() const { return data_; }
  uint64_t size() const { return size_; }

 private:
  const uint8_t* data_ = nullptr;
  uint64_t size_ = 0;
};

// Views of the tables inside prefix.t_table.bwt (byte pointers: the archive does not align them).
struct TableView {
  uint64_t interval = 0, n = 0, first_location = 0;
  const uint8_t* c_function = nullptr;   // 256 x u64
  const uint8_t* c_functions = nullptr;  // (4^12 + 1) x u64
  const uint8_t* occ[4] = {nullptr, nullptr, nullptr, nullptr};  // rows 'A','C','G','T', occ_len x u64 each
  uint64_t occ_len = 0;
  const uint8_t* loc = nullptr;  // n_loc x (u64 i, u64 SA)
  uint64_t n_loc = 0;
  const uint8_t* fbwt = nullptr;  // n bytes
  const uint8_t* jbwt = nullptr;  // jbwt_len bytes, 4 bases per byte, base 0 in bits 7..6
  uint64_t jbwt_len = 0;
};

bool parse_table(const MappedFile& f, TableView& v, std::string& err);

struct ChrTables {
  std::vector<std::string> names;
  std::vector<uint64_t> start;   // chr_start_pos keys, file order (ascending)
  uint64_t real_size = 0;        // sum of chrLen
  std::vector<uint64_t> npos_key, npos_val;  // chr_umbiguous_starting_length
};
bool read_chr_tables(const std::string& prefix_dot, ChrTables& t, std::string& err);

// Streaming writer of prefix.t_table.bwt.  Pieces must be written in archive order.
class TableWriter {
 public:
  ~TableWriter();
  bool open(const std::string& path, std::string& err);
  void u64(uint64_t v);
  void u32(uint32_t v);
  void zeros(int n);
  void bytes(const void* p, uint64_t n);
  bool close(std::string& err);
  void header();

 private:
  void* f_ = nullptr;
  bool bad_ = false;
  std::string path_;
};
bool write_npos(const std::string& path, const std::vector<uint64_t>& key, const std::vector<uint64_t>& val,
                std::string& err);

}  // namespace sbwt
