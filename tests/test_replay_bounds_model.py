"""The two bounds that let the parity mode replay std::partial_sort's heap over a sliver of the E values (csrc/cdist.hip),
checked on a pure-Python model of libstdc++'s heap algorithms: the replay over [0, P) + the id-sorted tail must leave the
same K pairs in the same order as the replay over all E pairs (what ATen's topk runs, reference pt.py:583).

  single engine   P = the id by which K candidates (value <= bound) have been seen; tail = the candidates behind P
  row partitions  t* = the smallest of the ranks' (K+1)-th best values; C = gathered keys <= t*; P = one past the K-th
                  smallest id in C; tail = the keys of C behind P (knn_merge_cdist_kernel)
No GPU needed."""
import random


def adjust_heap(first, hole, length, x):
    """bits/stl_heap.h __adjust_heap(first, hole, length, x) with the comparator on values alone, then __push_heap."""
    top = hole
    child = hole
    while child < (length - 1) // 2:
        child = 2 * (child + 1)
        if first[child][0] < first[child - 1][0]:
            child -= 1
        first[hole] = first[child]
        hole = child
    if length % 2 == 0 and child == (length - 2) // 2:
        child = 2 * (child + 1)
        first[hole] = first[child - 1]
        hole = child - 1
    parent = (hole - 1) // 2
    while hole > top and first[parent][0] < x[0]:
        first[hole] = first[parent]
        hole = parent
        parent = (hole - 1) // 2
    first[hole] = x


def partial_sort(pairs, K, heap=None):
    """std::partial_sort(first, first + K, last) on (value, id) pairs: __heap_select + __sort_heap.  `pairs` in the order
    they are visited; heap: continue from a heap another call left (the replay's second leg)."""
    if heap is None:
        heap = list(pairs[:K])
        for parent in range((K - 2) // 2, -1, -1):          # __make_heap
            adjust_heap(heap, parent, K, heap[parent])
        rest = pairs[K:]
    else:
        rest = pairs
    for p in rest:                                           # __heap_select: i enters iff *i < *first
        if p[0] < heap[0][0]:
            adjust_heap(heap, 0, K, p)                       # __pop_heap(first, middle, i)
    return heap


def sort_heap(heap):
    h = list(heap)
    for last in range(len(h) - 1, 0, -1):                    # __pop_heap(first, last, last)
        x = h[last]
        h[last] = h[0]
        adjust_heap(h, 0, last, x)
    return h


def test_prefix_and_tail_of_one_candidate_list_replay_like_all_values():
    rng = random.Random(11)
    for trial in range(300):
        E, K = rng.randrange(200, 3000), rng.randrange(2, 17)
        span = rng.choice([6, 40, 100000])                   # few distinct values: ties everywhere
        vals = [rng.randrange(span) for _ in range(E)]
        pairs = [(v, i) for i, v in enumerate(vals)]
        want = sort_heap(partial_sort(pairs, K))
        tau = sorted(vals)[min(E - 1, K + rng.randrange(0, 60))]          # the candidate list: every value <= tau
        cand = [p for p in pairs if p[0] <= tau]
        assert len(cand) >= K
        P = cand[K - 1][1] + 1                               # K candidates seen: the heap's maximum is <= tau from here on
        P = min(E, max(P, K))
        tail = [p for p in cand if p[1] >= P]
        got = sort_heap(partial_sort(tail, K, heap=partial_sort(pairs[:P], K)))
        assert got == want, (trial, E, K, span)


def test_bound_from_the_keys_gathered_over_row_partitions():
    rng = random.Random(5)
    for trial in range(300):
        world = rng.randrange(2, 9)
        K = rng.randrange(2, 13)
        E = rng.randrange(world * (K + 1) * 2, 4000)
        span = rng.choice([5, 30, 100000])
        vals = [rng.randrange(span) for _ in range(E)]
        pairs = [(v, i) for i, v in enumerate(vals)]
        want = sort_heap(partial_sort(pairs, K))
        owner = [rng.randrange(world) for _ in range(E)]     # hashed ownership
        gathered, tstar = [], None
        for r in range(world):
            own = sorted(p for p in pairs if owner[p[1]] == r)       # (value, id): the rank's best first
            keys = own[:K + 1]
            gathered += keys
            if len(own) > K + 1 or len(keys) == K + 1:               # a rank with more edges than keys: its (K+1)-th best bounds the rest
                b = keys[K][0] if len(keys) == K + 1 else None
                if b is not None:
                    tstar = b if tstar is None else min(tstar, b)
        if tstar is None:
            continue                                          # every rank sent all it has: nothing to bound (tiny graphs)
        C = sorted((p for p in gathered if p[0] <= tstar), key=lambda p: p[1])
        assert len(C) >= K + 1
        P = max(min(E, C[K - 1][1] + 1), K)
        tail = [p for p in C if p[1] >= P]
        got = sort_heap(partial_sort(tail, K, heap=partial_sort(pairs[:P], K)))
        assert got == want, (trial, world, E, K, span)
