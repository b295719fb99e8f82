// CPU harness for two host-side passes of the sparse block (tests/test_host_sanitizers.py builds it with -fsanitize=address,undefined
// and with -fsanitize=thread): the row-pattern recognition of BlockSparse::Initialize (linop.cpp: BuildRowPatterns) and the
// multi-core form of csr2csc (common.cpp).  The two product sources are compiled into this translation unit as they are; the device
// entry points they reference are never called here and stay unresolved (-Wl,--unresolved-symbols=ignore-all).
#include "../../prost_amd/csrc/host/common.cpp"
#include "../../prost_amd/csrc/host/linop.cpp"
#include <cstdio>
#include <random>
using namespace prost;
int main() {
  // gradient-like stencil: rows r: (r,-1), (r+ny,+1) ; boundary rows empty
  for (int rep = 0; rep < 3; rep++) {
    const size_t nx = 300 + rep * 57, ny = 211 + rep, n = nx * ny;
    std::vector<int32_t> ptr(2 * n + 1, 0), ind; std::vector<double> val;
    for (size_t r = 0; r < 2 * n; r++) {
      const size_t p = r % n; const bool gx = r < n;
      const bool has = gx ? (p / ny + 1 < nx) : (p % ny + 1 < ny);
      if (has) { ind.push_back((int32_t)p); val.push_back(-1.0); ind.push_back((int32_t)(p + (gx ? ny : 1))); val.push_back(1.0); }
      ptr[r + 1] = (int32_t)ind.size();
    }
    HostRowPatterns<double> h;
    const bool ok = BuildRowPatterns<double>(2 * n, ptr, ind, val, h);
    // verify
    size_t bad = 0;
    if (ok) for (size_t r = 0; r < 2 * n; r++) {
      const int id = h.ids[r]; const int b = h.pptr[id], e = h.pptr[id + 1];
      if (e - b != ptr[r + 1] - ptr[r]) { bad++; continue; }
      for (int k = 0; k < e - b; k++) if (h.rel[b + k] + (int32_t)r != ind[ptr[r] + k] || h.val[b + k] != val[ptr[r] + k]) bad++;
    }
    std::printf("stencil %zu rows: ok=%d patterns=%zu bad=%zu\n", 2 * n, (int)ok, h.pptr.size() - 1, bad);
  }
  {   // unstructured: must refuse
    std::mt19937 g(1); const size_t n = 50000;
    std::vector<int32_t> ptr(n + 1, 0), ind; std::vector<double> val;
    for (size_t r = 0; r < n; r++) { for (int k = 0; k < 3; k++) { ind.push_back((int32_t)(g() % n)); val.push_back((double)(g() % 1000) / 7.0); } std::sort(ind.end() - 3, ind.end()); ptr[r + 1] = (int32_t)ind.size(); }
    HostRowPatterns<double> h;
    std::printf("unstructured: ok=%d\n", (int)BuildRowPatterns<double>(n, ptr, ind, val, h));
  }
  {   // csr2csc parallel vs sequential on a banded matrix with > 4M entries
    const size_t n = 1500000; std::vector<int32_t> ptr(n + 1, 0), ind; std::vector<double> val;
    for (size_t r = 0; r < n; r++) { for (int k = -1; k <= 1; k++) { long c = (long)r + k * 700; if (c >= 0 && c < (long)n) { ind.push_back((int32_t)c); val.push_back(0.5 + k + r % 7); } } ptr[r + 1] = (int32_t)ind.size(); }
    const int nz = (int)ind.size();
    std::vector<double> a1(nz), a2(nz); std::vector<int32_t> r1(nz), r2(nz), c1(n + 1), c2(n + 1);
    csr2csc<double>((int)n, (int)n, nz, val.data(), ind.data(), ptr.data(), a1.data(), r1.data(), c1.data());
    // sequential reference
    { std::vector<int32_t> fill(n + 1, 0); for (int i = 0; i < nz; i++) fill[ind[i] + 1]++; for (size_t c = 0; c < n; c++) fill[c + 1] += fill[c]; for (size_t c = 0; c <= n; c++) c2[c] = fill[c];
      for (size_t r = 0; r < n; r++) for (int32_t j = ptr[r]; j < ptr[r + 1]; j++) { const int32_t d = fill[ind[j]]++; r2[d] = (int32_t)r; a2[d] = val[j]; } }
    std::printf("csr2csc %d entries: equal=%d\n", nz, (int)(a1 == a2 && r1 == r2 && c1 == c2));
  }
  return 0;
}
