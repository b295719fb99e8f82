// prost/common.hpp -- small host helpers shared by the solver classes
// (reference: include/prost/common.hpp, src/common.cu).
#ifndef PROST_COMMON_HPP_
#define PROST_COMMON_HPP_
#include <cstddef>
#include <cstdint>
#include <functional>
#include <list>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "prost/exception.hpp"

namespace prost {

using std::shared_ptr;
using std::vector;

std::string get_version();

/// num_in equally spaced values from start to end plus a trailing `end` (num_in + 1 entries;
/// reference src/common.cu:33-46 -- the callback schedule of Solver::Solve relies on the extra one)
template <typename T> std::list<double> linspace(T start_in, T end_in, int num_in);

/// CSR (n x m) -> CSC, not in place (reference src/common.cu:55-82)
template <typename T>
void csr2csc(int n, int m, int nz, const T* a, const int32_t* col_idx, const int32_t* row_start, T* csc_a,
             int32_t* row_idx, int32_t* col_start);

/// glibc's rand() stream for a given seed.  Problem::normest draws its start vector from
/// std::rand() with srand never called (reference src/problem.cu:435); every solve here uses the
/// stream a fresh process would see (seed 1), so repeated solves are reproducible.
class GlibcRand {
 public:
  explicit GlibcRand(unsigned seed = 1);
  int32_t next();
  /// out[i] = (T)next() / (T)RAND_MAX for i < n -- the same stream as n calls of next(), generated in one tight loop and
  /// narrowed on several host threads (normest draws one value per primal entry)
  template <typename T> void fill_unit(T* out, size_t n);

 private:
  std::vector<uint32_t> r_;
};

/// HIP stream all kernels of the calling thread are enqueued on (NULL stream by default)
void* CurrentStream();
void SetCurrentStream(void* stream);
/// throws prost::Exception(prost_hip_last_error()) if rc != 0
void CheckHip(int rc, const char* what);

/// fn(begin, end) over disjoint sub-ranges of [0, n) on up to 16 host threads (the one-off setup passes over 10^7..10^8
/// host entries are memory-bound loops); runs inline when n is small.  fn must only touch its own range.
void ParallelFor(size_t n, const std::function<void(size_t, size_t)>& fn);
/// number of sub-ranges ParallelFor(n, ...) uses, and the i-th of them (for two-phase scans with a carried value)
size_t ParallelChunks(size_t n);
void ParallelChunkRange(size_t n, size_t i, size_t& begin, size_t& end);

/// Device -> host copy of n elements through two pinned staging buffers (32 MiB each, kept for the life of the process): the
/// transfer of one chunk overlaps the host threads that move the previous chunk to `dst`, widening T -> D on the way (D = T:
/// plain copy).  `dst` may be pageable, untouched memory -- its pages are first touched by the copying threads.  Synchronous.
/// (400 MB of results at 4096^2: a pageable hipMemcpy into zero-filled std::vectors + a widening pass took 0.22 s.)
template <class D, class T> void DownloadAs(D* dst, const T* dev, size_t n);
/// The other direction for data that is GENERATED on the host: gen(p, len) fills consecutive pieces of the sequence straight
/// into the pinned staging buffers, each piece is on its way to `dev` while the next one is generated (no pageable copy of the
/// whole vector, no page faults on it).  Synchronous.
template <class T> void UploadGenerated(T* dev, size_t n, const std::function<void(T*, size_t)>& gen);

/// Wall-clock of a setup stage, printed to stderr at scope exit when the environment variable
/// PROST_TIMING is set (the stream is synchronised first so device work is attributed to its stage).
class StageTimer {
 public:
  explicit StageTimer(const char* name);
  ~StageTimer();
 private:
  const char* name_;
  double t0_;
  bool on_;
};

}  // namespace prost
#endif
