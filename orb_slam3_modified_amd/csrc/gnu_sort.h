// Sequential re-statement of libstdc++'s std::sort (GCC 11: introsort with median-of-3 pivot, depth limit
// 2*floor(log2 n), heapsort fallback, 16-element insertion-sort threshold) for use inside a HIP kernel.
//
// Why this exists: the reference's quadtree (src/ORBextractor.cc:697-701) sorts the expandable nodes with
// std::sort and a comparator on (point count, UL.x) only.  Nodes that tie on both are ordered by the
// *internals* of libstdc++'s introsort, and that order decides which nodes are split before the feature
// quota is reached and in which order children enter the node list — i.e. it changes the returned keypoints
// (SURVEY.md F9).  Bit-exact keypoints therefore need the same permutation, not just "a" sorted order.
//
// Elements are 64-bit words: the high 32 bits are the sort key (the only thing the comparator sees), the low
// 32 bits are a payload that travels with the element.  tests/test_gnu_sort.py checks this header against the
// host's std::sort on adversarial inputs (many ties, long runs, organ-pipe, killer sequences that hit the
// heapsort fallback).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_SORT_HD __host__ __device__ inline
#else
#define ORBX_SORT_HD static inline
#endif

namespace orbx_sort {

typedef uint64_t elem_t;

ORBX_SORT_HD bool lessk(elem_t a, elem_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }

template <typename P>
ORBX_SORT_HD void swap_at(P a, int i, int j) {
  elem_t t = a[i];
  a[i] = a[j];
  a[j] = t;
}

// std::__move_median_to_first(result, a, b, c)
template <typename P>
ORBX_SORT_HD void move_median_to_first(P v, int result, int a, int b, int c) {
  if (lessk(v[a], v[b])) {
    if (lessk(v[b], v[c])) swap_at(v, result, b);
    else if (lessk(v[a], v[c])) swap_at(v, result, c);
    else swap_at(v, result, a);
  } else if (lessk(v[a], v[c])) swap_at(v, result, a);
  else if (lessk(v[b], v[c])) swap_at(v, result, c);
  else swap_at(v, result, b);
}

// std::__unguarded_partition(first, last, pivot)
template <typename P>
ORBX_SORT_HD int unguarded_partition(P v, int first, int last, int pivot) {
  while (true) {
    while (lessk(v[first], v[pivot])) ++first;
    --last;
    while (lessk(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    swap_at(v, first, last);
    ++first;
  }
}

// std::__push_heap
template <typename P>
ORBX_SORT_HD void push_heap(P v, int first, int hole, int top, elem_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && lessk(v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}

// std::__adjust_heap
template <typename P>
ORBX_SORT_HD void adjust_heap(P v, int first, int hole, int len, elem_t value) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (lessk(v[first + second], v[first + (second - 1)])) second--;
    v[first + hole] = v[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    v[first + hole] = v[first + (second - 1)];
    hole = second - 1;
  }
  push_heap(v, first, hole, top, value);
}

// std::__partial_sort(first, last, last) == __heap_select(first,last,last) + __sort_heap(first,last)
template <typename P>
ORBX_SORT_HD void heap_sort(P v, int first, int last) {
#ifdef ORBX_SORT_COUNT_HEAP
  ++ORBX_SORT_COUNT_HEAP;
#endif
  int len = last - first;
  if (len >= 2) {  // __make_heap
    int parent = (len - 2) / 2;
    while (true) {
      elem_t value = v[first + parent];
      adjust_heap(v, first, parent, len, value);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {  // __sort_heap / __pop_heap
    --last;
    elem_t value = v[last];
    v[last] = v[first];
    adjust_heap(v, first, 0, last - first, value);
  }
}

// std::__unguarded_linear_insert
template <typename P>
ORBX_SORT_HD void unguarded_linear_insert(P v, int last) {
  elem_t val = v[last];
  int next = last - 1;
  while (lessk(val, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = val;
}

// std::__insertion_sort
template <typename P>
ORBX_SORT_HD void insertion_sort(P v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (lessk(v[i], v[first])) {
      elem_t val = v[i];
      for (int k = i; k > first; --k) v[k] = v[k - 1];  // move_backward
      v[first] = val;
    } else {
      unguarded_linear_insert(v, i);
    }
  }
}

// std::sort(first, last, comp) on v[0..n)
template <typename P>
ORBX_SORT_HD void gnu_sort(P v, int n) {
  if (n <= 0) return;
  // std::__introsort_loop, recursion on the right part replaced by an explicit stack (depth <= 2*lg n <= 64)
  int stack_first[64], stack_last[64], stack_depth[64];
  int sp = 0;
  int lg = 31 - __builtin_clz((unsigned)n);
  stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = lg * 2;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort(v, first, last);
        break;
      }
      --depth;
      int mid = first + (last - first) / 2;
      move_median_to_first(v, first, first + 1, mid, last - 1);
      int cut = unguarded_partition(v, first + 1, last, first);
      // libstdc++ recurses into [cut,last) first and then loops on [first,cut).  The two sub-ranges are
      // disjoint and each is processed by the same deterministic procedure, so the order in which they are
      // visited does not change the result; push the right part and continue with the left.
      stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = depth;
      ++sp;
      last = cut;
    }
  }
  // std::__final_insertion_sort
  if (n > 16) {
    insertion_sort(v, 0, 16);
    for (int i = 16; i != n; ++i) unguarded_linear_insert(v, i);
  } else {
    insertion_sort(v, 0, n);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Data-parallel formulation of the same permutation (what k_quadtree runs, one wave per segment):
//
//  * one Hoare partition step of libstdc++ (`__unguarded_partition(first+1, last, first)`) is a closed form of the
//    ORIGINAL segment: with a[1] < a[2] < ... the positions in (first, last) whose key is >= the pivot key and
//    b[1] > b[2] > ... the positions whose key is <= the pivot key (followed by the sentinel `first`), the loop swaps
//    exactly the pairs (a[j], b[j]) for j < J, J = first j with a[j] >= b[j], and returns
//    cut = a[J] if a[J] < b[J-1] (b[0] = last) else b[J-1]: between two swaps the pointers only scan untouched
//    positions, and the only touched positions they can stop at are the previous swap partners.
//  * `__final_insertion_sort` is a stable sort, and after the introsort loop every element already sits inside its
//    final (<= 16 element, or heap-sorted) segment, so it equals a stable rank inside each such segment.
//
// gnu_sort_model is the sequential statement of that formulation (tests/support/check_gnu_sort.cpp checks it against
// std::sort next to gnu_sort); partition_closed_form is shared with the device code.
template <typename P, typename I>
ORBX_SORT_HD int partition_closed_form(P v, int first, int last, I a, I r) {
  // a, r: scratch index arrays with room for last - first entries
  const uint32_t kp = (uint32_t)(v[first] >> 32);
  int nL = 0, nR = 0;
  for (int i = first + 1; i < last; i++) {
    const uint32_t k = (uint32_t)(v[i] >> 32);
    if (k >= kp) a[nL++] = i;
    if (k <= kp) r[nR++] = i;
  }
  const int lim = nL < nR + 1 ? nL : nR + 1;
  int cnt = 0;
  for (int j = 1; j <= lim; j++) {
    const int bj = j <= nR ? (int)r[nR - j] : first;
    if ((int)a[j - 1] < bj) cnt++;
  }
  for (int j = 1; j <= cnt; j++) swap_at(v, (int)a[j - 1], (int)r[nR - j]);
  const int J = cnt + 1;
  const int rprev = J == 1 ? last : (int)r[nR - (J - 1)];
  return (J <= nL && (int)a[J - 1] < rprev) ? (int)a[J - 1] : rprev;
}

template <typename P>
ORBX_SORT_HD void gnu_sort_model(P v, int n, int* idx_a, int* idx_r, int* seg_lo, int* seg_hi, elem_t* tmp, int* work) {
  // scratch: idx_a, idx_r, seg_lo, seg_hi, tmp hold n entries; work holds 6 * (n / 8 + 2) ints (pending segments)
  if (n <= 0) return;
  for (int i = 0; i < n; i++) { seg_lo[i] = 0; seg_hi[i] = n; }
  if (n > 16) {
    // level-synchronous introsort loop: every pending segment takes one partition step per round
    const int cap = n / 8 + 2;  // segments of one round are disjoint and longer than 16
    int *cur_f = work, *cur_l = work + cap, *cur_d = work + 2 * cap, ncur = 1;
    int *nxt_f = work + 3 * cap, *nxt_l = work + 4 * cap, *nxt_d = work + 5 * cap;
    cur_f[0] = 0; cur_l[0] = n; cur_d[0] = 2 * (31 - __builtin_clz((unsigned)n));
    while (ncur > 0) {
      int nn = 0;
      for (int s = 0; s < ncur; s++) {
        const int f = cur_f[s], l = cur_l[s], d = cur_d[s];
        if (d == 0) {
          heap_sort(v, f, l);
          for (int i = f; i < l; i++) { seg_lo[i] = i; seg_hi[i] = i + 1; }  // already in final order
          continue;
        }
        move_median_to_first(v, f, f + 1, f + (l - f) / 2, l - 1);
        const int cut = partition_closed_form(v, f, l, idx_a, idx_r);
        const int cf[2] = {f, cut}, cl[2] = {cut, l};
        for (int c = 0; c < 2; c++) {
          if (cl[c] - cf[c] > 16) { nxt_f[nn] = cf[c]; nxt_l[nn] = cl[c]; nxt_d[nn] = d - 1; nn++; }
          else for (int i = cf[c]; i < cl[c]; i++) { seg_lo[i] = cf[c]; seg_hi[i] = cl[c]; }
        }
      }
      for (int s = 0; s < nn; s++) { cur_f[s] = nxt_f[s]; cur_l[s] = nxt_l[s]; cur_d[s] = nxt_d[s]; }
      ncur = nn;
    }
  }
  // stable rank inside every final segment
  for (int i = 0; i < n; i++) {
    const uint32_t k = (uint32_t)(v[i] >> 32);
    int rank = seg_lo[i];
    for (int j = seg_lo[i]; j < seg_hi[i]; j++) {
      const uint32_t kj = (uint32_t)(v[j] >> 32);
      rank += (kj < k) || (kj == k && j < i);
    }
    tmp[rank] = v[i];
  }
  for (int i = 0; i < n; i++) v[i] = tmp[i];
}

}  // namespace orbx_sort
