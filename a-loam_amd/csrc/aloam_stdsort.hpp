// a-loam_amd/csrc/aloam_stdsort.hpp — the order libstdc++'s std::sort leaves EQUAL keys in, reproduced step by step.
//
// pcl::VoxelGrid::applyFilter sorts its (voxel index, point index) pairs with std::sort on the voxel index alone
// (pcl/filters/impl/voxel_grid.hpp; reference call sites src/scanRegistration.cpp:402-405, src/laserMapping.cpp:543-549,793-799) and then sums
// the members of a voxel in the order the sort left them.  std::sort is not stable, f32 addition is not associative, so the last bits of every
// centroid of three or more points depend on the exact sequence of swaps of the sort — introsort as libstdc++ (GCC 4.9 .. 13, bits/stl_algo.h,
// bits/stl_heap.h) writes it: median-of-three to the front, unguarded Hoare partition, recursion on the right part and iteration on the left until
// 16 elements or the depth limit 2 floor(log2 n) (then heap sort of the range), one final insertion sort.  This header is that sequence, written out for an
// array of 8-byte entries compared by their first word, callable on the host (tests/host/test_stdsort_port.cpp checks it against the real
// std::sort / std::partial_sort of this toolchain on millions of arrays with heavy duplication) and on the device (one lane runs it: the reference
// order is a sequential object).  Used only when a context asks for the reference's summation order (aloam_set_voxel_sum_order); the default path
// sums in input order and never includes this file's code in a hot kernel.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ALOAM_SS_HD __host__ __device__ inline
#else
#define ALOAM_SS_HD inline
#endif

namespace aloam {
namespace stdsort {

struct Entry { unsigned idx, pt; };                                // cloud_point_index_idx: operator< compares idx only
ALOAM_SS_HD bool less(const Entry& a, const Entry& b) { return a.idx < b.idx; }
ALOAM_SS_HD void swap_entries(Entry* a, int i, int j) { const Entry t = a[i]; a[i] = a[j]; a[j] = t; }

// std::__adjust_heap + std::__push_heap (bits/stl_heap.h)
ALOAM_SS_HD void adjust_heap(Entry* first, int hole, int len, Entry value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

// std::__partial_sort(first, last, last): __heap_select (= __make_heap, the selection loop is empty) + __sort_heap
ALOAM_SS_HD void heap_sort(Entry* first, int len) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      const Entry value = first[parent];
      adjust_heap(first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const Entry value = first[last];                               // __pop_heap(first, last, last)
    first[last] = first[0];
    adjust_heap(first, 0, last, value);
  }
}

// std::__unguarded_partition_pivot on [first, last) of a: returns the cut
ALOAM_SS_HD int partition_pivot(Entry* a, int first, int last) {
  const int mid = first + (last - first) / 2;
  const int x = first + 1, y = mid, z = last - 1;                   // __move_median_to_first(first, first + 1, mid, last - 1)
  if (less(a[x], a[y])) {
    if (less(a[y], a[z])) swap_entries(a, first, y);
    else if (less(a[x], a[z])) swap_entries(a, first, z);
    else swap_entries(a, first, x);
  } else if (less(a[x], a[z])) swap_entries(a, first, x);
  else if (less(a[y], a[z])) swap_entries(a, first, z);
  else swap_entries(a, first, y);
  int f = first + 1, l = last;                                      // __unguarded_partition(first + 1, last, pivot = first)
  while (true) {
    while (less(a[f], a[first])) ++f;
    --l;
    while (less(a[first], a[l])) --l;
    if (!(f < l)) return f;
    swap_entries(a, f, l);
    ++f;
  }
}

ALOAM_SS_HD void unguarded_linear_insert(Entry* a, int last) {
  const Entry val = a[last];
  int next = last - 1;
  while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
  a[last] = val;
}
ALOAM_SS_HD void insertion_sort(Entry* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      const Entry val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];             // move_backward(first, i, i + 1)
      a[first] = val;
    } else unguarded_linear_insert(a, i);
  }
}

// std::sort(a, a + n, less).  `stack` holds the pending right-hand ranges of __introsort_loop's recursion (first, last, depth limit): at most
// 2 floor(log2 n) + 1 of them.
constexpr int kThreshold = 16, kStackInts = 3 * 72;
ALOAM_SS_HD void sort(Entry* a, int n, int* stack) {
  if (n <= 0) return;
  int lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;                                // std::__lg(n)
  int sp = 0;
  stack[0] = 0; stack[1] = n; stack[2] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    const int first = stack[3 * sp];
    int last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
    // __introsort_loop(first, last, depth): libstdc++ recurses into [cut, last) and then loops on [first, cut).  The two ranges are disjoint
    // and nothing reads across them before the final insertion sort, so WHEN each is processed changes no swap: the right part is pushed
    // (with the decremented depth limit, as the recursive call gets it) and the loop goes on with the left one.
    int f = first;
    while (last - f > kThreshold) {
      if (depth == 0) { heap_sort(a + f, last - f); break; }
      --depth;
      const int cut = partition_pivot(a, f, last);
      stack[3 * sp] = cut; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; ++sp;   // the right part, with the decremented limit
      last = cut;
    }
  }
  if (n > kThreshold) {                                             // __final_insertion_sort
    insertion_sort(a, 0, kThreshold);
    for (int i = kThreshold; i != n; ++i) unguarded_linear_insert(a, i);
  } else insertion_sort(a, 0, n);
}

// ---- the same partition as DATA-PARALLEL steps (what a workgroup does for a large range; reference_order_kernels.hip) ----------------------------------
// __unguarded_partition alternates two scans: f runs up to the next element >= pivot, l runs down to the next element <= pivot, they swap, until
// they meet.  Until they meet both scans only ever look at elements no swap has touched, so the i-th stop of f is the i-th position (ascending,
// from first + 1) whose ORIGINAL key is >= pivot, F_i, and the i-th stop of l the i-th position (descending, from last - 1; `first` itself, the
// pivot, is the last stop l can reach) whose original key is <= pivot, L_i.  Pairs (F_i, L_i) are swapped while F_i < L_i - m of them, and since F
// ascends while L descends all F_i of swapped pairs lie left of all their L_i: the swaps are independent.  After the m-th swap f runs on to F_(m+1)
// unless it meets the element it has just put at L_m first (>= pivot by construction): cut = min(F_(m+1), L_m).  Two prefix counts, a rank match
// and a scatter - partition_by_lists() below is that formulation run sequentially, so that the host test can hold it against partition_pivot()
// (and both against std::sort); the device runs the same steps with wave ballots and block scans.
//   fpos / lpos: scratch for the two position lists (last - first entries each are enough).
ALOAM_SS_HD void median_to_first(Entry* a, int first, int last) {
  const int mid = first + (last - first) / 2;
  const int x = first + 1, y = mid, z = last - 1;
  if (less(a[x], a[y])) {
    if (less(a[y], a[z])) swap_entries(a, first, y);
    else if (less(a[x], a[z])) swap_entries(a, first, z);
    else swap_entries(a, first, x);
  } else if (less(a[x], a[z])) swap_entries(a, first, x);
  else if (less(a[y], a[z])) swap_entries(a, first, z);
  else swap_entries(a, first, y);
}
ALOAM_SS_HD int cut_from_lists(const int* fpos, int nf, const int* lpos, int nl, int m) {
  const int fnext = m < nf ? fpos[m] : 0x7fffffff;                 // F_(m+1) (0-based lists)
  const int lm = m > 0 ? lpos[m - 1] : 0x7fffffff;                 // L_m
  return fnext < lm ? fnext : lm;
}
ALOAM_SS_HD int partition_by_lists(Entry* a, int first, int last, int* fpos, int* lpos) {
  median_to_first(a, first, last);
  const unsigned pv = a[first].idx;
  int nf = 0, nl = 0;
  for (int p = first + 1; p < last; ++p) if (a[p].idx >= pv) fpos[nf++] = p;
  for (int p = last - 1; p > first; --p) if (a[p].idx <= pv) lpos[nl++] = p;
  lpos[nl++] = first;                                               // the pivot: the stop l cannot pass
  int m = 0;
  while (m < nf && m < nl && fpos[m] < lpos[m]) ++m;
  for (int i = 0; i < m; ++i) swap_entries(a, fpos[i], lpos[i]);
  return cut_from_lists(fpos, nf, lpos, nl, m);
}

// std::sort in the shape the device gives it: ranges of `big` elements and more are partitioned by lists (a workgroup), smaller ones become CHUNKS
// that one lane finishes on its own - introsort loop with the depth limit the range has inherited, then the insertion sort, which never moves an
// element across a cut (everything left of a cut is <= everything right of it), so finishing chunk by chunk IS the final insertion sort.
//   work: 3 ints per pending range (first, last, depth): at most (last - first) / big * 2 + 72 of them; chunk_fn(first, last, depth) is called once
//   per chunk, in any order.
template <class ChunkFn>
ALOAM_SS_HD void sort_by_chunks(Entry* a, int n, int big, int* work, int* fpos, int* lpos, ChunkFn&& chunk_fn) {
  if (n <= 0) return;
  int lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  int sp = 0;
  work[0] = 0; work[1] = n; work[2] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    const int first = work[3 * sp], last = work[3 * sp + 1], depth = work[3 * sp + 2];
    if (last - first < big || last - first <= kThreshold || depth == 0) { chunk_fn(first, last, depth); continue; }
    const int cut = partition_by_lists(a, first, last, fpos, lpos);
    work[3 * sp] = first; work[3 * sp + 1] = cut; work[3 * sp + 2] = depth - 1; ++sp;
    work[3 * sp] = cut; work[3 * sp + 1] = last; work[3 * sp + 2] = depth - 1; ++sp;
  }
}
// one chunk, by one lane: __introsort_loop(first, last, depth) and the insertion sort of the stretch
ALOAM_SS_HD void finish_chunk(Entry* a, int first, int last, int depth, int* stack) {
  int sp = 0;
  stack[0] = first; stack[1] = last; stack[2] = depth; sp = 1;
  while (sp > 0) {
    --sp;
    int f = stack[3 * sp], l = stack[3 * sp + 1], d = stack[3 * sp + 2];
    while (l - f > kThreshold) {
      if (d == 0) { heap_sort(a + f, l - f); break; }
      --d;
      const int cut = partition_pivot(a, f, l);
      stack[3 * sp] = cut; stack[3 * sp + 1] = l; stack[3 * sp + 2] = d; ++sp;
      l = cut;
    }
  }
  insertion_sort(a, first, last);
}

}  // namespace stdsort
}  // namespace aloam
