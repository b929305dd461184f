"""CPU model of the MSA kernel's wave-wide racon order (genomeworks_amd/csrc/poa_graph_device.h: topsort_racon_wave,
node_id_to_msa_pos_wave) against the serial routine it replaces (topsort_racon / node_id_to_msa_pos, i.e. the reference's
raconTopologicalSortDeviceUtil and its MSA column assignment): same depth-first walk, but the pushes of a visit are found
"by ballot" over all edge / alignment slots at once and written in slot order, marks and check flags are one byte per node,
and the MSA column of a node is a prefix count of the check flags over the sorted order."""
import random


def serial(n, in_edges, aligned):
    marks, check = [0] * n, [1] * n
    order, stack = [], []
    for i in range(n):
        if marks[i] != 0:
            continue
        stack.append(i)
        while stack:
            node = stack[-1]
            valid = True
            if marks[node] != 2:
                for b in in_edges[node]:
                    if marks[b] != 2:
                        stack.append(b)
                        valid = False
                if check[node]:
                    for a in aligned[node]:
                        if marks[a] != 2:
                            stack.append(a)
                            check[a] = 0
                            valid = False
                if valid:
                    marks[node] = 2
                    if check[node]:
                        order.append(node)
                        order.extend(aligned[node])
                else:
                    marks[node] = 1
            if valid:
                stack.pop()
    pos, col, rank = [0] * n, 0, 0
    while rank < len(order):
        node = order[rank]
        pos[node] = col
        for _ in aligned[node]:
            rank += 1
            pos[order[rank]] = col
        col += 1
        rank += 1
    return order, pos, col


def wave(n, in_edges, aligned):
    state = [4] * n  # marks [0:2), check [2]
    order, stack = [], []
    for i in range(n):
        if state[i] & 3:
            continue
        stack = [i]
        while stack:
            node = stack[-1]
            st = state[node]
            valid = True
            if (st & 3) != 2:
                check = bool(st & 4)
                push_e = [b for b in in_edges[node] if (state[b] & 3) != 2]            # ballot + prefix popcount: slot order
                push_a = [a for a in aligned[node] if check and (state[a] & 3) != 2]
                stack.extend(push_e)
                stack.extend(push_a)
                for a in push_a:
                    state[a] &= ~4
                valid = not push_e and not push_a
                if valid:
                    state[node] = (st & ~3) | 2
                    if check:
                        order.append(node)
                        order.extend(aligned[node])
                else:
                    state[node] = (st & ~3) | 1
            if valid:
                stack.pop()
    pos, col = [0] * n, 0
    for base in range(0, len(order), 64):  # one wavefront: ballot of "opens a column", inclusive prefix count
        chunk = order[base:base + 64]
        opens = [bool(state[v] & 4) for v in chunk]
        for lane, v in enumerate(chunk):
            pos[v] = col + sum(opens[:lane + 1]) - 1
        col += sum(opens)
    return order, pos, col


def random_poa_graph(rng, backbone, reads):
    """A POA-shaped DAG: a backbone chain, then per read substitutions (a new node aligned to the column's nodes) and
    insertions (a new node between two existing ones). Node ids grow in creation order, as in the kernels."""
    n = backbone
    in_edges = [[i - 1] if i else [] for i in range(n)]
    aligned = [[] for _ in range(n)]
    column = [[i] for i in range(n)]  # aligned groups along the backbone
    for _ in range(reads):
        prev = None
        for c in range(backbone):
            r = rng.random()
            if r < 0.08:      # deletion: skip the column
                continue
            if r < 0.2:       # substitution: a new node aligned with everything in the column
                node = n
                n += 1
                in_edges.append([])
                aligned.append(list(column[c]))
                for other in column[c]:
                    aligned[other].append(node)
                column[c].append(node)
            else:
                node = rng.choice(column[c])
            if prev is not None and prev not in in_edges[node]:
                in_edges[node].append(prev)
            prev = node
            if rng.random() < 0.05:  # insertion behind this column
                ins = n
                n += 1
                in_edges.append([prev])
                aligned.append([])
                prev = ins
    return n, in_edges, aligned


def test_wave_model_equals_serial_order_and_columns():
    rng = random.Random(11)
    for trial in range(60):
        n, in_edges, aligned = random_poa_graph(rng, rng.choice([5, 40, 150, 400]), rng.choice([1, 3, 8, 20]))
        o1, p1, c1 = serial(n, in_edges, aligned)
        o2, p2, c2 = wave(n, in_edges, aligned)
        assert o1 == o2 and sorted(o1) == list(range(n))
        assert p1 == p2 and c1 == c2
