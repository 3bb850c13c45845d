// Exact emulation of GCC libstdc++ std::sort (introsort) as a permutation.
//
// Why this exists: three of the four std::sort calls inside the reference's
// OverlapDetector::getSeqOverlaps (reference src/sequence/overlap.cpp:201-204,
// :269-275, :331-334, :432-434) sort with duplicate keys, std::sort is not
// stable, and the order it leaves equal keys in changes which OverlapRange
// records come out (SURVEY.md §7 "hard part 1").  Bit-parity therefore needs
// the very permutation GCC 11's std::sort produces.  This header restates that
// ALGORITHM (not its source) for any random-access "accessor":
//
//   introsort loop: while (n > 16) { if depth budget 2*floor(log2 n) is spent
//   -> heapsort the segment; pivot = median of (first+1, mid, last-1) swapped
//   into first; unguarded Hoare partition of [first+1,last) around *first;
//   recurse on the right part, continue on the left } ; then one insertion
//   sort pass (guarded for the first 16 elements, unguarded after).
//
// It is compiled three ways: by g++ for the CPU unit test against the real
// std::sort (tests/test_introsort.py), by g++ into the host shim, and by hipcc
// as device code for the per-target-group sorts of the chaining kernel.
//
// Accessor concept:  typedef T;  T load(int i);  void store(int i, const T&);
//                    bool less(const T& a, const T& b);
#pragma once

#if defined(__HIPCC__)
#define FG_HD __host__ __device__ __forceinline__
#else
#define FG_HD inline
#endif

namespace fgsort {

template <class A>
FG_HD void swap_at(A& a, int i, int j)
{
	typename A::T x = a.load(i);
	typename A::T y = a.load(j);
	a.store(i, y);
	a.store(j, x);
}

// --- heap fallback (std::__partial_sort(first, last, last)) ---------------
template <class A>
FG_HD void push_heap_(A& a, int first, int hole, int top, const typename A::T& value)
{
	int parent = (hole - 1) / 2;
	while (hole > top && a.less(a.load(first + parent), value))
	{
		a.store(first + hole, a.load(first + parent));
		hole = parent;
		parent = (hole - 1) / 2;
	}
	a.store(first + hole, value);
}

template <class A>
FG_HD void adjust_heap_(A& a, int first, int hole, int len, const typename A::T& value)
{
	const int top = hole;
	int child = hole;
	while (child < (len - 1) / 2)
	{
		child = 2 * (child + 1);
		if (a.less(a.load(first + child), a.load(first + child - 1))) --child;
		a.store(first + hole, a.load(first + child));
		hole = child;
	}
	if ((len & 1) == 0 && child == (len - 2) / 2)
	{
		child = 2 * (child + 1);
		a.store(first + hole, a.load(first + child - 1));
		hole = child - 1;
	}
	push_heap_(a, first, hole, top, value);
}

template <class A>
FG_HD void heap_sort_(A& a, int first, int last)
{
	const int len = last - first;
	if (len >= 2)
	{
		int parent = (len - 2) / 2;
		while (true)
		{
			typename A::T v = a.load(first + parent);
			adjust_heap_(a, first, parent, len, v);
			if (parent == 0) break;
			--parent;
		}
	}
	int end = last;
	while (end - first > 1)
	{
		--end;
		typename A::T v = a.load(end);
		a.store(end, a.load(first));
		adjust_heap_(a, first, 0, end - first, v);
	}
}

// --- pivot + partition ------------------------------------------------------
template <class A>
FG_HD void median_to_first_(A& a, int result, int ia, int ib, int ic)
{
	typename A::T va = a.load(ia), vb = a.load(ib), vc = a.load(ic);
	int pick;
	if (a.less(va, vb))
	{
		if (a.less(vb, vc)) pick = ib;
		else if (a.less(va, vc)) pick = ic;
		else pick = ia;
	}
	else if (a.less(va, vc)) pick = ia;
	else if (a.less(vb, vc)) pick = ic;
	else pick = ib;
	swap_at(a, result, pick);
}

template <class A>
FG_HD int partition_pivot_(A& a, int first, int last)
{
	int mid = first + (last - first) / 2;
	median_to_first_(a, first, first + 1, mid, last - 1);
	const typename A::T pivot = a.load(first);
	int lo = first + 1, hi = last;
	while (true)
	{
		while (a.less(a.load(lo), pivot)) ++lo;
		--hi;
		while (a.less(pivot, a.load(hi))) --hi;
		if (!(lo < hi)) return lo;
		swap_at(a, lo, hi);
		++lo;
	}
}

FG_HD int floor_log2_(int n)
{
	int r = 0;
	while (n > 1) { n >>= 1; ++r; }
	return r;
}

// quicksort phase only: leaves segments of <= 16 elements unsorted inside but
// mutually ordered (or fully heap-sorted where the depth budget ran out).
// STACK must hold >= 3 * (floor(log2 n) + 2) ints.
template <class A>
FG_HD void introsort_loop(A& a, int first, int last, int* stack)
{
	const int THRESH = 16;
	int sp = 0;
	int depth = 2 * floor_log2_(last - first);
	while (true)
	{
		while (last - first > THRESH)
		{
			if (depth == 0)
			{
				heap_sort_(a, first, last);
				break;
			}
			--depth;
			int cut = partition_pivot_(a, first, last);
			// both halves are independent: keep the smaller, stack the larger
			if (cut - first < last - cut)
			{
				stack[sp++] = cut; stack[sp++] = last; stack[sp++] = depth;
				last = cut;
			}
			else
			{
				stack[sp++] = first; stack[sp++] = cut; stack[sp++] = depth;
				first = cut;
			}
		}
		if (sp == 0) break;
		depth = stack[--sp]; last = stack[--sp]; first = stack[--sp];
	}
}

template <class A>
FG_HD void unguarded_linear_insert_(A& a, int last)
{
	typename A::T v = a.load(last);
	int next = last - 1;
	while (a.less(v, a.load(next)))
	{
		a.store(last, a.load(next));
		last = next;
		--next;
	}
	a.store(last, v);
}

template <class A>
FG_HD void insertion_sort_(A& a, int first, int last)
{
	if (first == last) return;
	for (int i = first + 1; i != last; ++i)
	{
		typename A::T v = a.load(i);
		if (a.less(v, a.load(first)))
		{
			for (int j = i; j > first; --j) a.store(j, a.load(j - 1));
			a.store(first, v);
		}
		else unguarded_linear_insert_(a, i);
	}
}

template <class A>
FG_HD void final_insertion_sort(A& a, int first, int last)
{
	const int THRESH = 16;
	if (last - first > THRESH)
	{
		insertion_sort_(a, first, first + THRESH);
		for (int i = first + THRESH; i != last; ++i) unguarded_linear_insert_(a, i);
	}
	else insertion_sort_(a, first, last);
}

// full std::sort(first, last) emulation
template <class A>
FG_HD void sort(A& a, int first, int last, int* stack)
{
	if (first == last) return;
	introsort_loop(a, first, last, stack);
	final_insertion_sort(a, first, last);
}

const int STACK_INTS = 3 * 34;	// enough for n < 2^31

} // namespace fgsort
