// tests/host/test_stdsort_port.cpp — a-loam_amd/csrc/aloam_stdsort.hpp against THIS toolchain's std::sort / std::partial_sort: the same permutation,
// element for element, on arrays shaped like pcl::VoxelGrid's index vector (few distinct keys, runs of equal keys, sorted / reversed / organ-pipe
// stretches, sizes around the insertion-sort threshold and up to 200 000).  The heap-sort arm of introsort (depth limit) is not reached by such
// data, so it is checked on its own against std::partial_sort(first, last, last), which is the call std::__introsort_loop makes.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../a-loam_amd/csrc/aloam_stdsort.hpp"

using aloam::stdsort::Entry;

static bool same(const std::vector<Entry>& a, const std::vector<Entry>& b) {
  for (size_t i = 0; i < a.size(); ++i) if (a[i].idx != b[i].idx || a[i].pt != b[i].pt) return false;
  return true;
}

int main() {
  std::mt19937 rng(12345);
  long long arrays = 0, elements = 0;
  int stack[aloam::stdsort::kStackInts];
  auto check = [&](std::vector<Entry> v, const char* what) {
    std::vector<Entry> a = v, b = v;
    std::sort(a.begin(), a.end(), [](const Entry& x, const Entry& y) { return x.idx < y.idx; });
    aloam::stdsort::sort(b.data(), (int)b.size(), stack);
    if (!same(a, b)) { std::printf("MISMATCH std::sort n=%zu (%s)\n", v.size(), what); std::exit(1); }
    if (v.size() >= 2) {
      std::vector<Entry> c = v, d = v;
      std::partial_sort(c.begin(), c.end(), c.end(), [](const Entry& x, const Entry& y) { return x.idx < y.idx; });
      aloam::stdsort::heap_sort(d.data(), (int)d.size());
      if (!same(c, d)) { std::printf("MISMATCH heap sort n=%zu (%s)\n", v.size(), what); std::exit(1); }
    }
    {   // the data-parallel formulation: large ranges partitioned by lists, chunks finished one by one (in reverse order: any order must do)
      std::vector<Entry> e = v;
      std::vector<int> work(3 * (v.size() / 8 + 80)), fpos(v.size() + 2), lpos(v.size() + 2);
      std::vector<int> chunks;
      for (int big : {24, 300}) {
        e = v; chunks.clear();
        aloam::stdsort::sort_by_chunks(e.data(), (int)e.size(), big, work.data(), fpos.data(), lpos.data(), [&](int f, int l, int d) { chunks.push_back(f); chunks.push_back(l); chunks.push_back(d); });
        for (size_t c = 0; c < chunks.size(); c += 3) aloam::stdsort::finish_chunk(e.data(), chunks[c], chunks[c + 1], chunks[c + 2], stack);
        if (!same(a, e)) { std::printf("MISMATCH chunked sort n=%zu big=%d (%s)\n", v.size(), big, what); std::exit(1); }
      }
    }
    ++arrays; elements += (long long)v.size();
  };
  auto make = [&](int n, int distinct, int shape) {
    std::vector<Entry> v(n);
    int run = 0; unsigned cur = 0;
    for (int i = 0; i < n; ++i) {
      unsigned k;
      switch (shape) {
        case 0: k = rng() % (unsigned)distinct; break;                                    // random, heavy duplication
        case 1: if (run == 0) { cur = rng() % (unsigned)distinct; run = 1 + rng() % 9; } --run; k = cur; break;   // runs of equal keys (ring-ordered points)
        case 2: k = (unsigned)((long long)i * distinct / n); break;                           // already sorted with duplicates
        case 3: k = (unsigned)((long long)(n - 1 - i) * distinct / n); break;                 // reversed
        case 4: k = (unsigned)(i < n / 2 ? i : n - i) % (unsigned)distinct; break;            // organ pipe
        default: k = (unsigned)((i * 7919u) % (unsigned)distinct); break;                    // strided
      }
      v[i] = Entry{k, (unsigned)i};
    }
    return v;
  };
  for (int n = 0; n <= 70; ++n)
    for (int rep = 0; rep < 400; ++rep) check(make(n, 1 + (int)(rng() % 8), (int)(rng() % 6)), "small");
  const int sizes[] = {100, 257, 600, 1000, 2048, 5000, 12345, 40000, 65536, 200000};
  for (int n : sizes)
    for (int rep = 0; rep < (n <= 5000 ? 300 : 12); ++rep) {
      const int distinct = 1 + (int)(rng() % (unsigned)(rep % 3 == 0 ? 8 : (rep % 3 == 1 ? n / 3 + 1 : n + 1)));
      check(make(n, distinct, (int)(rng() % 6)), "large");
    }
  std::printf("aloam_stdsort == std::sort / std::partial_sort on %lld arrays, %lld elements\n", arrays, elements);
  return 0;
}
