"""PyTorch-CPU restatement of the reference's iteration (SURVEY.md 8d "CPU baseline beside it"): the same
torch ops as graphem_rapids/backends/embedder_pytorch.py (gathers, torch.norm, index_add_, cdist + topk,
boolean-mask compaction, mean / unbiased std) WITHOUT its MemoryManager / gc.collect wrappers, which are
98 % of the reference's own CPU wall time at small n (SURVEY headline fact 5).

TEST INFRASTRUCTURE ONLY: used by bench.py's cpu_baseline leg ("torch_cpu") and checked against the golden
vectors in tests/test_oracle_golden.py.  Written from the reference's description in SURVEY.md 8a, not copied.
"""
import torch


def spring_forces(pos, edges, L_min=1.0, k_attr=0.2):
    """pt.py:595-636."""
    p1, p2 = pos[edges[:, 0]], pos[edges[:, 1]]
    diff = p2 - p1
    dist = torch.norm(diff, dim=1, keepdim=True) + 1e-6
    f = (-k_attr * (dist - L_min)) * (diff / dist)
    F = torch.zeros_like(pos)
    F.index_add_(0, edges[:, 0], f)
    F.index_add_(0, edges[:, 1], -f)
    return F


def knn_midpoints(mid, sampled, k):
    """pt.py:381-424, 543-593: cdist + topk(k+1), column 0 dropped."""
    d = torch.cdist(mid[sampled], mid, p=2)
    _, idx = torch.topk(d, k + 1, dim=1, largest=False)
    return idx[:, 1:]


def _orient(a, b, c):
    return (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])


def intersection_forces(pos, edges, knn, sampled, k_inter=0.5):
    """pt.py:638-774."""
    F = torch.zeros_like(pos)
    i = sampled.unsqueeze(1).expand(-1, knn.shape[1]).reshape(-1)
    j = knn.reshape(-1)
    keep = i < j
    i, j = i[keep], j[keep]
    if i.numel() == 0:
        return F
    e1, e2 = edges[i], edges[j]
    share = (e1[:, 0] == e2[:, 0]) | (e1[:, 0] == e2[:, 1]) | (e1[:, 1] == e2[:, 0]) | (e1[:, 1] == e2[:, 1])
    e1, e2 = e1[~share], e2[~share]
    if e1.shape[0] == 0:
        return F
    p1, p2, q1, q2 = pos[e1[:, 0]], pos[e1[:, 1]], pos[e2[:, 0]], pos[e2[:, 1]]
    hit = (_orient(p1, p2, q1) * _orient(p1, p2, q2) < 0) & (_orient(q1, q2, p1) * _orient(q1, q2, p2) < 0)
    if not hit.any():
        return F
    e1, e2 = e1[hit], e2[hit]
    p1, p2, q1, q2 = p1[hit], p2[hit], q1[hit], q2[hit]
    c = (p1 + p2 + q1 + q2) / 4.0
    for x, ids in ((p1, e1[:, 0]), (p2, e1[:, 1]), (q1, e2[:, 0]), (q2, e2[:, 1])):
        diff = x - c
        dist = torch.norm(diff, dim=1, keepdim=True) + 1e-6
        F.index_add_(0, ids, k_inter * diff / dist ** 2)
    return F


def step(pos, edges, sampled, k, L_min=1.0, k_attr=0.2, k_inter=0.5):
    """pt.py:776-806 on torch tensors (pos float32 (n, D), edges int64 (E, 2), sampled int64 (S,))."""
    Fs = spring_forces(pos, edges, L_min, k_attr)
    mid = (pos[edges[:, 0]] + pos[edges[:, 1]]) / 2.0
    knn = knn_midpoints(mid, sampled, k)
    Fi = intersection_forces(pos, edges, knn, sampled, k_inter)
    new = pos + (Fs + Fi)
    new = new - new.mean(dim=0)
    return new / (new.std(dim=0) + 1e-6)
