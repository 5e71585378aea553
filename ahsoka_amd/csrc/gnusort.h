// gnusort.h -- the sorting ALGORITHM std::sort runs in libstdc++ (introsort:
// median-of-3 quicksort to depth 2*floor(lg n), heapsort below that, final
// insertion sort with threshold 16), written out so that device code can
// reproduce the permutation the reference gets on ties.
//
// Why: the reference orders reads with ReadSet::sort() -- std::sort with a
// comparator that looks at the first variant position only
// (src/alignmentstoreadset.cpp:297) -- and orders clusters per position with
// std::sort on coverage only (:707-727).  Both comparators leave ties, and
// std::sort is not stable, so read indices and cluster order depend on the
// exact sequence of comparisons and moves.  This is an
// independent implementation of the published algorithm, iterative (explicit
// stack) so it runs on one GPU lane; tests/test_gnusort.py pins it against
// the real std::sort of this toolchain on tie-heavy and adversarial inputs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AHK_HDN __host__ __device__
#else
#define AHK_HDN
#endif

namespace ahk {

template <class T, class Less>
AHK_HDN inline void gs_unguarded_linear_insert(T* a, int last, Less less) {
  T val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <class T, class Less>
AHK_HDN inline void gs_insertion_sort(T* a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      T val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];
      a[first] = val;
    } else {
      gs_unguarded_linear_insert(a, i, less);
    }
  }
}

template <class T, class Less>
AHK_HDN inline void gs_adjust_heap(T* a, int first, int hole, int len, T value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a[first + child], a[first + child - 1])) child--;
    a[first + hole] = a[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[first + hole] = a[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less(a[first + parent], value)) {
    a[first + hole] = a[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[first + hole] = value;
}

template <class T, class Less>
AHK_HDN inline void gs_heapsort(T* a, int first, int last, Less less) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      T v = a[first + parent];
      gs_adjust_heap(a, first, parent, len, v, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    T v = a[l];
    a[l] = a[first];
    gs_adjust_heap(a, first, 0, l - first, v, less);
  }
}

template <class T>
AHK_HDN inline void gs_swap(T& x, T& y) {
  T t = x;
  x = y;
  y = t;
}

template <class T, class Less>
AHK_HDN inline int gs_partition_pivot(T* a, int first, int last, Less less) {
  int mid = first + (last - first) / 2;
  // move median of a[first+1], a[mid], a[last-1] to a[first]
  int ia = first + 1, ib = mid, ic = last - 1;
  if (less(a[ia], a[ib])) {
    if (less(a[ib], a[ic])) gs_swap(a[first], a[ib]);
    else if (less(a[ia], a[ic])) gs_swap(a[first], a[ic]);
    else gs_swap(a[first], a[ia]);
  } else if (less(a[ia], a[ic])) gs_swap(a[first], a[ia]);
  else if (less(a[ib], a[ic])) gs_swap(a[first], a[ic]);
  else gs_swap(a[first], a[ib]);
  int lo = first + 1, hi = last;
  for (;;) {
    while (less(a[lo], a[first])) ++lo;
    --hi;
    while (less(a[first], a[hi])) --hi;
    if (!(lo < hi)) return lo;
    gs_swap(a[lo], a[hi]);
    ++lo;
  }
}

// sorts a[0..n)
template <class T, class Less>
AHK_HDN inline void gnu_sort(T* a, int n, Less less) {
  if (n <= 0) return;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) depth++;
  depth *= 2;
  // introsort loop with an explicit stack of pending [first,last) ranges; the
  // recursive call is on the RIGHT part, the loop continues on the left, so
  // right parts are stacked and processed after the left spine -- the order
  // does not matter for the result (the ranges are disjoint).
  int stk_first[64], stk_last[64], stk_depth[64];
  int sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = depth;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk_first[sp], last = stk_last[sp], d = stk_depth[sp];
    while (last - first > 16) {
      if (d == 0) {
        gs_heapsort(a, first, last, less);
        break;
      }
      --d;
      int cut = gs_partition_pivot(a, first, last, less);
      stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = d;
      ++sp;
      last = cut;
    }
  }
  if (n > 16) {
    gs_insertion_sort(a, 0, 16, less);
    for (int i = 16; i != n; ++i) gs_unguarded_linear_insert(a, i, less);
  } else {
    gs_insertion_sort(a, 0, n, less);
  }
}

struct LessHi16 {
  AHK_HDN bool operator()(uint32_t x, uint32_t y) const { return (x >> 16) < (y >> 16); }
};

}  // namespace ahk
