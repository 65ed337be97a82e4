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

}  // namespace orbx_sort
