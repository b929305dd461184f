"""Scalar model of the consensus kernel's first heaviest-bundle pass (generate_consensus_lds, genomeworks_amd/csrc/
poa_graph_device.h: bundle_pass_chunked) next to the plain recurrence it replaces (cudapoa_generate_consensus.cuh:120-170, the
node-by-node bundle_pass): 64 sorted positions at a time, every lane decides its node's in-edge by the weights, predecessors
inside the chunk are recognised by markers (-2 - lane) in the score table, sums along the chains inside the chunk by pointer
jumping, nodes with a tie that involves an in-chunk predecessor (or with more than three in-edges) stop the chains that run
through them and take the plain step once the lanes before them are final. Random graphs with few distinct weights (ties
everywhere), long chains, bubbles and wide nodes; scores, predecessors and the end of the bundle must be identical."""
import random

WAVE = 64
SEEN = {"stopped_nodes": 0, "jump_rounds": 0, "longest_jump": 0, "chains_through_a_stopped_node": 0}


def plain_pass(order, in_edges):
    n = len(order)
    scores, pred = [0] * n, [-1] * n
    max_score, max_id = -1, 0
    for node in order:
        best_w, best, best_score = -1, -1, -1
        for begin, w in in_edges[node]:
            sc = scores[begin]
            if best_w < w or (best_w == w and best_score <= sc):
                best_w, best, best_score = w, begin, sc
        score = best_w + best_score if best != -1 else best_w
        scores[node], pred[node] = score, best
        if max_score <= score:
            max_score, max_id = score, node
    return scores, pred, max_id


def node_step(node, in_edges, scores):
    best_w, best, best_score = -1, -1, -1
    for begin, w in in_edges[node]:
        sc = scores[begin]
        if best_w < w or (best_w == w and best_score <= sc):
            best_w, best, best_score = w, begin, sc
    return best, (best_w + best_score if best != -1 else best_w)


def chunked_pass(order, in_edges):
    n = len(order)
    scores, pred = [0] * n, [-1] * n
    lane_max, lane_arg = [-2] * WAVE, [0] * WAVE
    for base in range(0, n, WAVE):
        lanes = range(min(WAVE, n - base))
        node = [order[base + l] for l in lanes]
        for l in lanes:
            scores[node[l]] = -2 - l  # markers, before anybody looks a predecessor up
        val, ref, waiting, slow, best = {}, {}, {}, {}, {}
        for l in lanes:
            edges = in_edges[node[l]]
            best_w, b, best_score, s = -1, -1, -1, len(edges) > 3
            for begin, w in edges[:3]:
                sc = scores[begin]
                s = s or (best_w == w and (sc <= -2 or best_score <= -2))
                if best_w < w or (best_w == w and best_score <= sc):
                    best_w, b, best_score = w, begin, sc
            inside = best_score <= -2
            slow[l], best[l] = s, b
            val[l] = 0 if s else (best_w if inside else (best_w + best_score if b != -1 else best_w))
            ref[l] = (-2 - best_score) if (inside and not s) else l
            waiting[l] = inside or s
        stoppers = sorted(l for l in lanes if slow[l])
        SEEN["stopped_nodes"] += len(stoppers)
        while True:
            rounds = 0
            while True:  # pointer jumping: every lane reads the state of the lane it points at, all at once
                hop = [l for l in lanes if waiting[l] and l not in stoppers and ref[l] not in stoppers]
                if not hop:
                    break
                rounds += 1
                SEEN["jump_rounds"] += 1
                SEEN["longest_jump"] = max(SEEN["longest_jump"], rounds)
                assert rounds <= 6, "a chain inside 64 lanes needs at most six doublings"
                snap = {l: (val[l], ref[l], waiting[l]) for l in lanes}
                for l in hop:
                    v, r, w = snap[ref[l]]
                    val[l] += v
                    waiting[l] = w
                    if w:
                        ref[l] = r
            if not stoppers:
                break
            j = stoppers.pop(0)
            for l in lanes:
                if l < j:
                    assert not waiting[l], "a lane in front of the first stopped node is not final"
                    scores[node[l]] = val[l]
            bj, sj = node_step(node[j], in_edges, scores)
            best[j], val[j], waiting[j] = bj, sj, False
            for l in lanes:
                if l != j and waiting[l] and ref[l] == j and l not in stoppers:
                    SEEN["chains_through_a_stopped_node"] += 1
                    val[l] += sj
                    waiting[l] = False
        for l in lanes:
            assert not waiting[l]
            scores[node[l]], pred[node[l]] = val[l], best[l]
            if lane_max[l] <= val[l]:
                lane_max[l], lane_arg[l] = val[l], ((base + l) << 12) | node[l]
    m = max(lane_max)
    arg = max(lane_arg[l] if lane_max[l] == m else -1 for l in range(WAVE))
    return scores, pred, arg & 0xfff


def random_graph(rng, n, weights, p_extra, p_wide):
    order = list(range(n))
    rng.shuffle(order)  # node ids in sorted order: position k holds node order[k]
    in_edges = [[] for _ in range(n)]
    for k in range(1, n):
        preds = set()
        if rng.random() < 0.93:
            preds.add(k - 1)
        while rng.random() < p_extra and len(preds) < k:
            preds.add(max(0, k - 1 - int(rng.expovariate(0.15))))
        if rng.random() < p_wide:
            for _ in range(rng.randrange(3, 7)):
                preds.add(rng.randrange(k))
        edges = [(order[p], rng.choice(weights)) for p in preds]
        rng.shuffle(edges)
        in_edges[order[k]] = edges
    return order, in_edges


def test_chunked_pass_equals_the_plain_recurrence():
    rng = random.Random(20260926)
    for case in range(300):
        n = rng.choice([1, 2, 63, 64, 65, 127, 200, 500, 1500])
        weights = rng.choice([[1], [1, 2], [1, 2, 3], list(range(1, 33))])
        order, in_edges = random_graph(rng, n, weights, p_extra=rng.choice([0.05, 0.3, 0.6]), p_wide=rng.choice([0.0, 0.02, 0.1]))
        assert chunked_pass(order, in_edges) == plain_pass(order, in_edges), (case, n, weights)
    # the cases reached what the kernel's special paths are for
    assert SEEN["stopped_nodes"] > 1000 and SEEN["chains_through_a_stopped_node"] > 1000 and SEEN["longest_jump"] == 6, SEEN
