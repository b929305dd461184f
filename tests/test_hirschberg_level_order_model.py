"""The level-by-level default aligner (genomeworks_amd/csrc/gwhip_myers.hip, hirschberg_levels_kernel) does not pop a range
stack like the reference (hirschberg_myers_gpu.cu:575-644: left half pushed first, so the right half is aligned first and
the path comes out back to front); it splits all parts of a level at once and afterwards appends the terminals of all
levels in descending (query begin, target begin) order. This is a model of that claim alone -- the order of the terminals --
on random split trees with empty halves on either side, single characters and leaves of every size: plain Python, no
device, no oracle."""
import random


def terminal(qn, tn):
    # the case order of the kernels: empty side -> run, short query with a matrix that fits -> leaf, single character
    if tn == 0 or qn == 0:
        return True
    if 2 <= qn < 63:
        return True
    return qn == 1


def depth_first(q, t, rng_seed):
    rng = random.Random(rng_seed)
    split = {}
    out, stack = [], [(0, q, 0, t)]
    while stack:
        qb, qe, tb, te = stack.pop()
        if qe - qb == 0 and te - tb == 0:
            continue
        if terminal(qe - qb, te - tb):
            out.append((qb, qe, tb, te))
            continue
        qm = qb + (qe - qb) // 2
        tm = split.setdefault((qb, qe, tb, te), rng.choice([tb, te, rng.randint(tb, te), rng.randint(tb, te)]))
        stack.append((qb, qm, tb, tm))   # left first: popped last
        stack.append((qm, qe, tm, te))
    return out, split


def level_by_level(q, t, split):
    level, terms = [(0, q, 0, t)], []
    while level:
        nxt = []
        for qb, qe, tb, te in level:
            if qe - qb == 0 and te - tb == 0:
                continue
            if terminal(qe - qb, te - tb):
                terms.append((qb, qe, tb, te))
                continue
            qm, tm = qb + (qe - qb) // 2, split[(qb, qe, tb, te)]
            nxt += [(qb, qm, tb, tm), (qm, qe, tm, te)]
        level = nxt
    return sorted(terms, key=lambda p: (p[0], p[2]), reverse=True)


def test_terminals_in_descending_query_target_order_are_the_stack_order():
    rng = random.Random(5)
    for case in range(400):
        q = rng.choice([1, 2, 62, 63, 64, 125, 126, 500, 1000, 2047, 2048])
        t = rng.choice([0, 1, q // 3, q, q + q // 8, 2300])
        if q == 0 and t == 0:
            continue
        df, split = depth_first(q, t, case)
        assert level_by_level(q, t, split) == df
        keys = [(p[0], p[2]) for p in df]
        assert len(set(keys)) == len(keys)  # the key is unique among terminals with an extent
