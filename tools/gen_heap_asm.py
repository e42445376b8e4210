#!/usr/bin/env python3
"""Writes graphem-rapids_amd/csrc/cdist_heap_asm.h: the replace-the-maximum step of the K <= 16 heap of the parity mode's
replay (csrc/cdist.hip, cdist_heap_scalar) as ONE block of gfx950 scalar instructions over 16 pinned SGPR pairs.

libstdc++'s __adjust_heap(first, 0, len, x) + __push_heap, written as a decision tree: every heap index is a register
name, so an element that enters costs 7 instructions per level it passes (is there a child, which child, is it below x:
s_cselect_b64 moves the child up or lands x) -- the compiler's rendering of the same tree over two register arrays copied
the whole heap between register sets at every merge (NOTES.md round 4: 526 cycles per element).

usage: python tools/gen_heap_asm.py          rewrites the header in place (the header is committed)
       python tools/gen_heap_asm.py --check  checks the block against a port of __adjust_heap and the committed header
                                             against what would be written (tests/test_heap_asm_generator.py)
"""
import os
import sys

BASE = 40      # heap element i lives in s[BASE + 2 i : BASE + 2 i + 1]: id in the low half, value key in the high half
KMAX = 16


def reg(i):
    return "s[%d:%d]" % (BASE + 2 * i, BASE + 2 * i + 1)


def hi(i):
    return "s%d" % (BASE + 2 * i + 1)


def internal(i):   # may have a child for some len <= KMAX
    return 2 * i + 1 <= KMAX - 1


def after_move(i):   # where control goes once the hole is at i
    return (".Lgh_n%d_%%=" if internal(i) else ".Lgh_c%d_%%=") % i


out = []


def emit(s):
    out.append(s)


# Layout: a taken branch costs an instruction-buffer refill (tens of cycles; an instruction ~4), so the tree is laid out
# depth first -- the right child's block follows its parent's move directly, the left one follows the landing pad of the
# one conditional branch -- and every climb is a straight line from its starting index to the root, left by one branch.
# The block walks DOWN only.  libstdc++ sinks the hole to the bottom along the larger children w_1 >= w_2 >= ... (heap
# order) and lets x climb back while its parent is smaller; the result is w_1 .. w_j one level up and x below them, j = the
# number of path elements that are not below x -- so the walk can stop at the first larger child that IS below x (the ones
# under it are no larger), which is what one s_cselect_b64 per level does: the child moves up, or x lands and the block ends.
# Layout: a taken branch costs an instruction-buffer refill (~28 cycles measured; an instruction ~4), so the tree is laid
# out depth first: the right child's block follows its parent's directly, the left one follows the landing pad of the one
# conditional branch of the level.
def node(i):
    l, r = 2 * i + 1, 2 * i + 2
    if not internal(i):
        emit("s_mov_b64 %s, %%[x]" % reg(i))
        emit("s_branch .Lgh_done_%=")
        return
    if r <= KMAX - 1:
        emit("s_cmp_gt_u32 %%[len], %d" % r)              # two children?
        emit("s_cbranch_scc0 .Lgh_s%d_%%=" % i)
        emit("s_cmp_lt_u32 %s, %s" % (hi(r), hi(l)))      # the right one unless it is smaller than the left
        emit("s_cbranch_scc1 .Lgh_l%d_%%=" % i)
        emit("s_cmp_lt_u32 %s, %%[xv]" % hi(r))           # below x: x lands here
        emit("s_cselect_b64 %s, %%[x], %s" % (reg(i), reg(r)))
        emit("s_cbranch_scc1 .Lgh_done_%=")
        node(r)
        emit(".Lgh_l%d_%%=:" % i)
        emit("s_cmp_lt_u32 %s, %%[xv]" % hi(l))
        emit("s_cselect_b64 %s, %%[x], %s" % (reg(i), reg(l)))
        emit("s_cbranch_scc1 .Lgh_done_%=")
        node(l)
        emit(".Lgh_s%d_%%=:" % i)
    emit("s_cmp_gt_u32 %%[len], %d" % l)                  # a single child (len even), or none
    emit("s_cbranch_scc0 .Lgh_p%d_%%=" % i)
    emit("s_cmp_lt_u32 %s, %%[xv]" % hi(l))
    emit("s_cselect_b64 %s, %%[x], %s" % (reg(i), reg(l)))
    emit("s_cbranch_scc1 .Lgh_done_%=")
    emit("s_mov_b64 %s, %%[x]" % reg(l))
    emit("s_branch .Lgh_done_%=")
    emit(".Lgh_p%d_%%=:" % i)
    emit("s_mov_b64 %s, %%[x]" % reg(i))
    emit("s_branch .Lgh_done_%=")


node(0)
while out[-1].startswith("s_branch .Lgh_done"):
    out.pop()
emit(".Lgh_done_%=:")


def run_block(h, length, x):
    """Interprets the generated block (the six instructions it uses) on heap h (list of (value, id)); -> taken branches."""
    labels = {s[:-1]: k for k, s in enumerate(out) if s.endswith(":")}
    regs = {reg(i): h[i] for i in range(KMAX)}
    his = {hi(i): i for i in range(KMAX)}

    def val(tok):
        if tok == "%[xv]":
            return x[0]
        if tok == "%[len]":
            return length
        if tok in his:
            return regs[reg(his[tok])][0]
        return int(tok)
    pc, scc, taken, steps = 0, 0, 0, 0
    while pc < len(out):
        ins = out[pc]
        pc += 1
        if ins.endswith(":"):
            continue
        steps += 1
        op, rest = ins.split(" ", 1)
        a = [t.strip() for t in rest.split(", ")]
        if op == "s_cmp_gt_u32":
            scc = int(val(a[0]) > val(a[1]))
        elif op == "s_cmp_lt_u32":
            scc = int(val(a[0]) < val(a[1]))
        elif op == "s_mov_b64":
            regs[a[0]] = x if a[1] == "%[x]" else regs[a[1]]
        elif op == "s_cselect_b64":
            src = a[1] if scc else a[2]
            regs[a[0]] = x if src == "%[x]" else regs[src]
        elif op in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1"):
            if op == "s_branch" or (op == "s_cbranch_scc0") == (scc == 0):
                pc = labels[a[0]]
                taken += 1
        else:
            raise ValueError(ins)
    for i in range(KMAX):
        h[i] = regs[reg(i)]
    return taken, steps


def adjust_heap(first, length, x):
    """libstdc++ bits/stl_heap.h __adjust_heap(first, 0, length, x, less-on-values) followed by __push_heap."""
    hole = child = 0
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
    while hole > 0 and first[parent][0] < x[0]:
        first[hole] = first[parent]
        hole = parent
        parent = (hole - 1) // 2
    first[hole] = x


def self_check():
    import random
    rng = random.Random(7)
    tk = st = n = 0
    for length in range(1, KMAX + 1):
        for trial in range(400):
            span = rng.choice([2, 3, 5, 1000])   # few distinct values: ties everywhere
            vals = sorted((rng.randrange(span) for _ in range(length)), reverse=True)
            h = [(v, 100 + k) for k, v in enumerate(vals)]   # a sorted array is a heap; then churn it
            h += [(0xFFFFFFFF, 0)] * (KMAX - length)
            g = list(h)
            for step in range(12):
                x = (rng.randrange(span + 1), 1000 + step)
                a, b = run_block(h, length, x)
                adjust_heap(g, length, x)
                assert h == g, (length, trial, step)
                tk += a
                st += b
                n += 1
    return "self-check: %d replacements identical to __adjust_heap; %.1f instructions, %.1f taken branches each" % (n, st / n, tk / n)


HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphem-rapids_amd", "csrc", "cdist_heap_asm.h")


def header_text():
    t = []
    t.append("// GENERATED by tools/gen_heap_asm.py -- do not edit.  The K <= %d heap of csrc/cdist.hip in pinned scalar registers:\n" % KMAX)
    t.append("// element i = %s ... (id low, value key high); one block replaces the maximum by [x] ([xv] = its value key) in a\n" % reg(0))
    t.append("// heap of [len] elements exactly as libstdc++'s __adjust_heap(first, 0, len, x) + __push_heap do.\n")
    t.append("#pragma once\n")
    t.append("#define GH_CD_HEAP_BASE %d\n" % BASE)
    t.append("#define GH_CD_HEAP_REPLACE_ASM \\\n")
    for line in out:
        t.append('    "%s\\n" \\\n' % line)
    t.append('    ""\n')
    t.append("#define GH_CD_HEAP_OPERANDS(h) \\\n    ")
    t.append(", ".join('"+{%s}"((h)[%d])' % (reg(i), i) for i in range(KMAX)))
    t.append("\n")
    return "".join(t)


if __name__ == "__main__":
    print(self_check())
    if "--check" in sys.argv[1:]:
        same = open(HEADER).read() == header_text()
        print("committed header", "matches" if same else "DIFFERS from", "the generator's output")
        sys.exit(0 if same else 1)
    with open(HEADER, "w") as f:
        f.write(header_text())
    print("wrote", os.path.normpath(HEADER), len(out), "lines")
