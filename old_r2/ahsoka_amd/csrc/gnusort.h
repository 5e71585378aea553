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
// stack); tests/test_gnusort.py pins it against the real std::sort of this
// toolchain on tie-heavy and adversarial inputs.
//
// The algorithm is written against an ARRAY ACCESSOR (get(i) / set(i, v)), so the
// same code sorts a plain array (host test hook, small device arrays) or an array
// held in the vector registers of one wavefront (WaveRegArray, k_readset.hip:
// element i in lane i & 63 of register i >> 6, reached with v_readlane /
// v_writelane).  The sort is inherently sequential -- the tie order IS the
// sequence of comparisons -- so what can be made fast is the access: ~10 cycles
// per element in registers against ~100 for a dependent LDS read.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AHK_HDN __host__ __device__
#else
#define AHK_HDN
#endif

namespace ahk {

template <class T>
struct PtrArray {
  T* p;
  AHK_HDN T get(int i) const { return p[i]; }
  AHK_HDN void set(int i, T v) { p[i] = v; }
};

template <class A, class Less>
AHK_HDN inline void gs_unguarded_linear_insert(A& a, int last, Less less) {
  const auto val = a.get(last);
  int next = last - 1;
  while (less(val, a.get(next))) {
    a.set(last, a.get(next));
    last = next;
    --next;
  }
  a.set(last, val);
}

template <class A, class Less>
AHK_HDN inline void gs_insertion_sort(A& a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a.get(i), a.get(first))) {
      const auto val = a.get(i);
      for (int k = i; k > first; --k) a.set(k, a.get(k - 1));
      a.set(first, val);
    } else {
      gs_unguarded_linear_insert(a, i, less);
    }
  }
}

template <class A, class T, class Less>
AHK_HDN inline void gs_adjust_heap(A& a, int first, int hole, int len, T value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(a.get(first + child), a.get(first + child - 1))) child--;
    a.set(first + hole, a.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.set(first + hole, a.get(first + child - 1));
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && less(a.get(first + parent), value)) {
    a.set(first + hole, a.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(first + hole, value);
}

template <class A, class Less>
AHK_HDN inline void gs_heapsort(A& a, int first, int last, Less less) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const auto v = a.get(first + parent);
      gs_adjust_heap(a, first, parent, len, v, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    const auto v = a.get(l);
    a.set(l, a.get(first));
    gs_adjust_heap(a, first, 0, l - first, v, less);
  }
}

template <class A>
AHK_HDN inline void gs_swap(A& a, int i, int j) {
  const auto t = a.get(i);
  a.set(i, a.get(j));
  a.set(j, t);
}

template <class A, class Less>
AHK_HDN inline int gs_partition_pivot(A& a, int first, int last, Less less) {
  const int mid = first + (last - first) / 2;
  // move median of a[first+1], a[mid], a[last-1] to a[first]
  const int ia = first + 1, ib = mid, ic = last - 1;
  {
    const auto va = a.get(ia), vb = a.get(ib), vc = a.get(ic);
    if (less(va, vb)) {
      if (less(vb, vc)) gs_swap(a, first, ib);
      else if (less(va, vc)) gs_swap(a, first, ic);
      else gs_swap(a, first, ia);
    } else if (less(va, vc)) gs_swap(a, first, ia);
    else if (less(vb, vc)) gs_swap(a, first, ic);
    else gs_swap(a, first, ib);
  }
  const auto pivot = a.get(first);  // (a[first] is not touched by the partition loop)
  int lo = first + 1, hi = last;
  for (;;) {
    while (less(a.get(lo), pivot)) ++lo;
    --hi;
    while (less(pivot, a.get(hi))) --hi;
    if (!(lo < hi)) return lo;
    gs_swap(a, lo, hi);
    ++lo;
  }
}

// sorts a[0..n); st0/st1/st2: three scratch arrays of 64 ints (the stack of pending ranges)
template <class A, class S, class Less>
AHK_HDN inline void gnu_sort_acc(A& a, int n, Less less, S& st_first, S& st_last, S& st_depth) {
  if (n <= 0) return;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) depth++;
  depth *= 2;
  // introsort loop with an explicit stack of pending [first,last) ranges; the
  // recursive call is on the RIGHT part, the loop continues on the left, so
  // right parts are stacked and processed after the left spine -- the order
  // does not matter for the result (the ranges are disjoint).
  int sp = 0;
  st_first.set(0, 0);
  st_last.set(0, n);
  st_depth.set(0, depth);
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = st_first.get(sp), last = st_last.get(sp), d = st_depth.get(sp);
    while (last - first > 16) {
      if (d == 0) {
        gs_heapsort(a, first, last, less);
        break;
      }
      --d;
      const int cut = gs_partition_pivot(a, first, last, less);
      st_first.set(sp, cut);
      st_last.set(sp, last);
      st_depth.set(sp, d);
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

// plain array form
template <class T, class Less>
AHK_HDN inline void gnu_sort(T* a, int n, Less less) {
  int stk[3][64];
  PtrArray<T> arr{a};
  PtrArray<int> s0{stk[0]}, s1{stk[1]}, s2{stk[2]};
  gnu_sort_acc(arr, n, less, s0, s1, s2);
}

// ---------------------------------------------------------------------------
// The same permutation, computed in a form that parallelises (gnu_sort_par_ref is
// its sequential statement, pinned against std::sort by tests/test_gnusort.py; the
// wavefront version is in k_readset.hip).  Two facts about the algorithm above:
//
//  * The unguarded Hoare partition of [first+1,last) around the pivot a[first]
//    only ever looks at ORIGINAL elements until its pointers cross: with
//    l_1 < l_2 < ... the positions whose element is not less than the pivot and
//    r_1 > r_2 > ... those whose element is not greater, it swaps (l_k, r_k) for
//    k = 1..K, K = #{k : l_k < r_k} (a prefix, the sequences are monotone), and
//    returns cut = min(l_{K+1}, r_K) (missing terms = +infinity).  Positions and K
//    follow from prefix counts; the swaps touch disjoint positions.
//  * After partitioning, every element left of a cut is <= every element right of
//    it, and the final insertion sort moves an element left only past STRICTLY
//    greater ones: no element ever crosses a cut.  The final pass is therefore an
//    independent stable sort of every leaf range (<= 16 elements, or a range the
//    depth limit handed to heapsort, which is sorted already): each element's
//    place is first + #smaller in the leaf + #equal to its left.
template <class T, class Less>
AHK_HDN inline int gs_partition_pivot_ranked(T* a, int first, int last, Less less, uint16_t* posL, uint16_t* posR) {
  const int mid = first + (last - first) / 2;
  const int ia = first + 1, ib = mid, ic = last - 1;
  {
    const T va = a[ia], vb = a[ib], vc = a[ic];
    int pick;
    if (less(va, vb)) pick = less(vb, vc) ? ib : (less(va, vc) ? ic : ia);
    else pick = less(va, vc) ? ia : (less(vb, vc) ? ic : ib);
    const T t = a[first];
    a[first] = a[pick];
    a[pick] = t;
  }
  const T pivot = a[first];
  int nL = 0, nR = 0;
  for (int i = first + 1; i < last; i++)
    if (!less(a[i], pivot)) posL[nL++] = (uint16_t)i;
  for (int j = last - 1; j > first; j--)
    if (!less(pivot, a[j])) posR[nR++] = (uint16_t)j;
  int K = 0;
  for (int k = 0; k < nL && k < nR; k++) K += posL[k] < posR[k] ? 1 : 0;
  for (int k = 0; k < K; k++) {
    const T t = a[posL[k]];
    a[posL[k]] = a[posR[k]];
    a[posR[k]] = t;
  }
  const int c1 = K < nL ? (int)posL[K] : 0x7fffffff, c2 = K > 0 ? (int)posR[K - 1] : 0x7fffffff;
  return c1 < c2 ? c1 : c2;
}

// sorts a[0..n), n <= 65535; posL/posR: n entries each; cutbit: bit i set = a leaf starts at i
// ((n + 31) / 32 words); tmp: n entries
template <class T, class Less>
AHK_HDN inline void gnu_sort_par_ref(T* a, int n, Less less, uint16_t* posL, uint16_t* posR, uint32_t* cutbit, T* tmp) {
  if (n <= 0) return;
  for (int w = 0; w < (n + 31) / 32; w++) cutbit[w] = 0;
  cutbit[0] |= 1u;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) depth++;
  depth *= 2;
  int stk[3][64];
  int sp = 0;
  stk[0][0] = 0; stk[1][0] = n; stk[2][0] = depth;
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stk[0][sp], last = stk[1][sp], d = stk[2][sp];
    while (last - first > 16) {
      if (d == 0) {
        PtrArray<T> arr{a};
        gs_heapsort(arr, first, last, less);  // (sorted: the final pass leaves it as it is)
        break;
      }
      --d;
      const int cut = gs_partition_pivot_ranked(a, first, last, less, posL, posR);
      cutbit[cut >> 5] |= 1u << (cut & 31);
      stk[0][sp] = cut; stk[1][sp] = last; stk[2][sp] = d;
      ++sp;
      last = cut;
    }
  }
  // final pass: a stable sort of every leaf
  for (int i = 0; i < n; i++) {
    int f = i;
    while (!((cutbit[f >> 5] >> (f & 31)) & 1u)) f--;
    int l = i + 1;
    while (l < n && !((cutbit[l >> 5] >> (l & 31)) & 1u)) l++;
    int rank = f;
    for (int j = f; j < l; j++) rank += (less(a[j], a[i]) || (j < i && !less(a[i], a[j]))) ? 1 : 0;
    tmp[rank] = a[i];
  }
  for (int i = 0; i < n; i++) a[i] = tmp[i];
}

struct LessHi16 {
  AHK_HDN bool operator()(uint32_t x, uint32_t y) const { return (x >> 16) < (y >> 16); }
};

}  // namespace ahk
