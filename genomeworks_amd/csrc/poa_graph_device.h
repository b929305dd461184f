// poa_graph_device.h -- graph merge, topological sorts, consensus and MSA on the device.
// Semantics per SURVEY.md section 8a rows P5-P8 (reference: cudapoa_add_alignment.cuh:65-285,
// cudapoa_topsort.cuh:45-197, cudapoa_generate_consensus.cuh:35-283, cudapoa_generate_msa.cuh:35-125).
// These phases are order-defining (node ids, edge slots, queue order, tie breaks), so the exact
// sequential order is kept; they run on lane 0 of the window's wavefront.
#pragma once
#include "poa_device.h"

namespace gwhip
{

// addAlignmentToGraph: walks the alignment back to front (= read order).
template <typename IdT, bool MSA>
__device__ uint8_t add_alignment_to_graph(int32_t& new_node_count, const GraphView<IdT>& g, int32_t node_count,
                                          int32_t alignment_length, const int32_t* alignment_graph,
                                          const uint8_t* read, const int32_t* alignment_read,
                                          const int8_t* base_weights, IdT* sequence_begin_nodes_ids, uint16_t s,
                                          uint32_t max_sequences_per_poa, uint32_t max_limit_nodes_per_window)
{
    int32_t head_node_id = -1;
    int32_t curr_node_id = -1;
    uint16_t prev_weight = 0;
    for (int32_t pos = alignment_length - 1; pos >= 0; pos--)
    {
        int32_t read_pos = alignment_read[pos];
        if (read_pos == -1) continue;
        int8_t node_weight    = base_weights[read_pos];
        uint8_t read_base     = read[read_pos];
        int32_t graph_node_id = alignment_graph[pos];
        if (graph_node_id == -1)
        {
            curr_node_id = node_count++;
            if ((uint32_t)node_count >= max_limit_nodes_per_window) return kNodeCountExceeded;
            g.nodes[curr_node_id]                = read_base;
            g.outgoing_edge_count[curr_node_id]  = 0;
            g.incoming_edge_count[curr_node_id]  = 0;
            g.node_alignment_count[curr_node_id] = 0;
            g.coverage[curr_node_id]             = 0;
        }
        else
        {
            uint8_t graph_base = g.nodes[graph_node_id];
            if (graph_base == read_base)
                curr_node_id = graph_node_id;
            else
            {
                uint16_t num_aligned    = g.node_alignment_count[graph_node_id];
                int32_t aligned_node_id = -1;
                for (int32_t n = 0; n < num_aligned; n++)
                {
                    int32_t aid = g.node_alignments[(int64_t)graph_node_id * kAligns + n];
                    if (g.nodes[aid] == read_base)
                    {
                        aligned_node_id = aid;
                        break;
                    }
                }
                if (aligned_node_id != -1)
                    curr_node_id = aligned_node_id;
                else
                {
                    curr_node_id = node_count++;
                    if ((uint32_t)node_count >= max_limit_nodes_per_window) return kNodeCountExceeded;
                    g.nodes[curr_node_id]                = read_base;
                    g.outgoing_edge_count[curr_node_id]  = 0;
                    g.incoming_edge_count[curr_node_id]  = 0;
                    g.node_alignment_count[curr_node_id] = 0;
                    g.coverage[curr_node_id]             = 0;
                    int32_t new_alignments               = 0;
                    for (int32_t n = 0; n < num_aligned; n++)
                    {
                        int32_t aid        = g.node_alignments[(int64_t)graph_node_id * kAligns + n];
                        uint16_t aid_count = g.node_alignment_count[aid];
                        g.node_alignments[(int64_t)aid * kAligns + aid_count]               = (IdT)curr_node_id;
                        g.node_alignment_count[aid]                                         = aid_count + 1;
                        g.node_alignments[(int64_t)curr_node_id * kAligns + new_alignments] = (IdT)aid;
                        new_alignments++;
                    }
                    g.node_alignments[(int64_t)graph_node_id * kAligns + num_aligned]   = (IdT)curr_node_id;
                    g.node_alignment_count[graph_node_id]                               = num_aligned + 1;
                    g.node_alignments[(int64_t)curr_node_id * kAligns + new_alignments] = (IdT)graph_node_id;
                    new_alignments++;
                    g.node_alignment_count[curr_node_id] = (uint16_t)new_alignments;
                }
            }
        }
        if (MSA && read_pos == 0) *sequence_begin_nodes_ids = (IdT)curr_node_id;
        if (head_node_id != -1)
        {
            bool edge_exists  = false;
            uint16_t in_count = g.incoming_edge_count[curr_node_id];
            for (int32_t e = 0; e < in_count; e++)
            {
                if (g.incoming_edges[(int64_t)curr_node_id * kEdges + e] == head_node_id)
                {
                    edge_exists = true;
                    g.incoming_edge_w[(int64_t)curr_node_id * kEdges + e] += (uint16_t)(prev_weight + node_weight);
                }
            }
            if (!edge_exists)
            {
                g.incoming_edges[(int64_t)curr_node_id * kEdges + in_count]  = (IdT)head_node_id;
                g.incoming_edge_w[(int64_t)curr_node_id * kEdges + in_count] = (uint16_t)(prev_weight + node_weight);
                g.incoming_edge_count[curr_node_id]                          = in_count + 1;
                uint16_t out_count                                           = g.outgoing_edge_count[head_node_id];
                g.outgoing_edges[(int64_t)head_node_id * kEdges + out_count] = (IdT)curr_node_id;
                if (MSA)
                {
                    g.out_cov_cnt[(int64_t)head_node_id * kEdges + out_count] = 1;
                    g.out_cov[((int64_t)head_node_id * kEdges + out_count) * max_sequences_per_poa] = s;
                }
                g.outgoing_edge_count[head_node_id] = out_count + 1;
                if (out_count + 1 >= kEdges || in_count + 1 >= kEdges) return kEdgeCountExceeded;
            }
            else if (MSA)
            {
                uint16_t out_count = g.outgoing_edge_count[head_node_id];
                for (int32_t e = 0; e < out_count; e++)
                {
                    if (g.outgoing_edges[(int64_t)head_node_id * kEdges + e] == curr_node_id)
                    {
                        uint16_t cc = g.out_cov_cnt[(int64_t)head_node_id * kEdges + e];
                        g.out_cov[((int64_t)head_node_id * kEdges + e) * max_sequences_per_poa + cc] = s;
                        g.out_cov_cnt[(int64_t)head_node_id * kEdges + e]                            = cc + 1;
                        break;
                    }
                }
            }
        }
        head_node_id = curr_node_id;
        g.coverage[head_node_id]++;
        prev_weight = (uint16_t)node_weight;
    }
    new_node_count = node_count;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Wave-parallel graph merge, bit-identical to add_alignment_to_graph.
//
// In a global alignment every read position rp = 0..L-1 appears exactly once, so the merge is a map over rp:
//   A. classify rp (reuse the aligned graph node / reuse one of its aligned nodes / create a node), read-only;
//   B. number the new nodes by an ordered prefix sum over rp (= the serial creation order);
//   C. create nodes and cross-link aligned-node lists;
//   D. for every consecutive pair (rp-1, rp) bump or append the edge, bump coverage.
// Different rp touch different nodes as long as no two nodes on the alignment path are aligned to each other
// (they never are in a well-formed POA); phase A verifies exactly that and returns -1, before anything was
// modified, so the caller can fall back to the serial routine. Error precedence (first failing rp, node
// overflow before edge overflow inside one rp) is reproduced.
// Scratch: gnode[L], curr[L] (NodeT) and a node bitset `onpath` -- in LDS when the graph fits there (16-bit ids),
// otherwise in the window's score matrix, which is dead between the traceback and the next forward pass.
// ------------------------------------------------------------------------------------------------
template <typename IdT, bool MSA, typename NodeT = int16_t>
__device__ __forceinline__ int32_t add_alignment_parallel(int32_t& new_node_count, const GraphView<IdT>& g,
                                                          int32_t node_count, int32_t alen, const int32_t* ag,
                                                          const int32_t* ar, const uint8_t* read,
                                                          const int8_t* base_weights, int32_t read_length,
                                                          IdT* sequence_begin_nodes_ids, uint16_t s,
                                                          uint32_t max_sequences_per_poa, int32_t max_nodes,
                                                          NodeT* gnode, NodeT* curr, uint32_t* onpath, int lane,
                                                          int32_t dbg = 0, uint64_t* prof_acc = nullptr)
{
    const int32_t L = read_length;
    const int32_t prof_sel = (dbg >> 16) & 7; // profiling (GWHIP_DEBUG bits 16-18): 1 = load, 2 = classify, 3 = create, 4 = edges
    uint64_t prof_t = prof_sel ? clock64() : 0;
    auto prof_mark = [&](int32_t which) {
        if (prof_sel)
        {
            const uint64_t now = clock64();
            if (prof_sel == which && prof_acc) *prof_acc += now - prof_t;
            prof_t = now;
        }
    };
    for (int32_t i = lane; i < (node_count + 31) / 32; i += kWave) onpath[i] = 0;
    for (int32_t i = lane; i < L; i += kWave) gnode[i] = -1;
    wave_sync();
    int32_t covered = 0;
    for (int32_t base = 0; base < alen; base += 4 * kWave) // four chunks per HBM round trip
    {
        int32_t rp[4], gn[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int32_t k = base + u * kWave + lane;
            rp[u] = k < alen ? ar[k] : -1;
            gn[u] = k < alen ? ag[k] : -1; // independent of rp: all loads are in flight together
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            if (rp[u] >= 0)
            {
                gnode[rp[u]] = (NodeT)gn[u];
                if (gn[u] >= 0) atomicOr(&onpath[gn[u] >> 5], 1u << (gn[u] & 31));
            }
            covered += __popcll(__ballot(rp[u] >= 0));
        }
    }
    wave_sync();
    // The map over read positions needs a complete alignment (every read position exactly once); a degenerate
    // traceback result (reference quirk: a walk that finds no predecessor at its first step) goes the serial way.
    if (covered != L) return -1;
    prof_mark(1);

    // ---- A + B: classify, detect conflicts, number new nodes in rp order ----
    int32_t running   = 0;
    bool conflict     = false;
    int32_t rp_nodeerr = INT32_MAX;
    // Four 64-position chunks per iteration share the HBM round trips: read base and graph node -> the node's base,
    // its aligned-node count and first aligned node -> that node's base. Longer aligned-node lists (rare) loop on.
    // New node ids are numbered chunk by chunk, in read-position order.
    for (int32_t base = 0; base < L; base += 4 * kWave)
    {
        int32_t gn[4], nb[4], na[4], a0[4], nb0[4];
        uint8_t rb[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int32_t rp = base + u * kWave + lane;
            gn[u] = rp < L ? (int32_t)gnode[rp] : -1;
            rb[u] = rp < L ? read[rp] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int32_t gi = max(gn[u], 0);
            nb[u] = g.nodes[gi];
            na[u] = g.node_alignment_count[gi];
            a0[u] = g.node_alignments[(int64_t)gi * kAligns];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            nb0[u] = g.nodes[(gn[u] >= 0 && na[u] > 0 && (uint32_t)a0[u] < (uint32_t)node_count) ? a0[u] : 0];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int32_t rp = base + u * kWave + lane;
            bool is_new      = false;
            int32_t cur      = -1;
            if (rp < L)
            {
                if (gn[u] < 0)
                    is_new = true;
                else if (nb[u] == (int32_t)rb[u])
                    cur = gn[u];
                else
                {
                    for (int32_t n = 0; n < na[u]; n++)
                    {
                        const int32_t aid = n == 0 ? a0[u] : (int32_t)g.node_alignments[(int64_t)gn[u] * kAligns + n];
                        if ((onpath[aid >> 5] >> (aid & 31)) & 1u) conflict = true;
                        const int32_t ab = n == 0 ? nb0[u] : (int32_t)g.nodes[aid];
                        if (cur < 0 && ab == (int32_t)rb[u]) cur = aid;
                    }
                    if (cur < 0) is_new = true;
                }
            }
            const unsigned long long m = __ballot(is_new);
            if (is_new)
            {
                cur = node_count + running + __popcll(m & ((1ull << lane) - 1));
                if (cur + 1 >= max_nodes) rp_nodeerr = min(rp_nodeerr, rp);
            }
            if (rp < L) curr[rp] = (NodeT)cur;
            running += __popcll(m);
        }
    }
    if (__any(conflict)) return -1;
    for (int off = 32; off > 0; off >>= 1) rp_nodeerr = min(rp_nodeerr, __shfl_xor(rp_nodeerr, off));
    wave_sync();

    prof_mark(2);
    const int32_t first_new = node_count;
    auto in_count_of  = [&](int32_t n) -> int32_t { return n >= first_new ? 0 : (int32_t)g.incoming_edge_count[n]; };
    auto out_count_of = [&](int32_t n) -> int32_t { return n >= first_new ? 0 : (int32_t)g.outgoing_edge_count[n]; };

    if (rp_nodeerr != INT32_MAX)
    {
        // node overflow at rp_nodeerr: an edge overflow at an earlier rp would have been reported first
        int32_t rp_edgeerr = INT32_MAX;
        for (int32_t rp = 1 + lane; rp < rp_nodeerr; rp += kWave)
        {
            const int32_t head = curr[rp - 1], cur = curr[rp];
            const int32_t ic   = in_count_of(cur);
            bool exists        = false;
            for (int32_t e = 0; e < ic; e++)
                if (g.incoming_edges[(int64_t)cur * kEdges + e] == head) exists = true;
            if (!exists && (out_count_of(head) + 1 >= kEdges || ic + 1 >= kEdges)) rp_edgeerr = min(rp_edgeerr, rp);
        }
        for (int off = 32; off > 0; off >>= 1) rp_edgeerr = min(rp_edgeerr, __shfl_xor(rp_edgeerr, off));
        return rp_edgeerr != INT32_MAX ? (int32_t)kEdgeCountExceeded : (int32_t)kNodeCountExceeded;
    }

    // ---- C: create nodes, cross-link aligned-node lists ----
    for (int32_t rp = lane; rp < L; rp += kWave)
    {
        const int32_t cur = curr[rp];
        if (cur < first_new) continue;
        const int32_t gn          = gnode[rp];
        g.nodes[cur]               = read[rp];
        g.outgoing_edge_count[cur] = 0;
        g.incoming_edge_count[cur] = 0;
        g.coverage[cur]            = 0;
        int32_t na_new             = 0;
        if (gn >= 0)
        {
            const int32_t na = g.node_alignment_count[gn];
            for (int32_t n = 0; n < na; n++)
            {
                const int32_t aid  = g.node_alignments[(int64_t)gn * kAligns + n];
                const int32_t acnt = g.node_alignment_count[aid];
                g.node_alignments[(int64_t)aid * kAligns + acnt] = (IdT)cur;
                g.node_alignment_count[aid]                      = (uint16_t)(acnt + 1);
                g.node_alignments[(int64_t)cur * kAligns + n]    = (IdT)aid;
            }
            g.node_alignments[(int64_t)gn * kAligns + na]  = (IdT)cur;
            g.node_alignment_count[gn]                     = (uint16_t)(na + 1);
            g.node_alignments[(int64_t)cur * kAligns + na] = (IdT)gn;
            na_new                                         = na + 1;
        }
        g.node_alignment_count[cur] = (uint16_t)na_new;
    }
    wave_sync();
    prof_mark(3);

    // ---- D: edges and coverage ----
    // Read positions are independent here (a node is the current node of one position and the head of one
    // position), so four 64-position chunks go through each HBM round trip together: first every load of the four
    // chunks (coverage, in-degree, head's out-degree, the first four in-edge slots with their weights -- slots past
    // the in-degree hold stale values and are masked), then their stores.
    int32_t rp_edgeerr = INT32_MAX;
    constexpr int kUD = 4, kBatch = 4;
    for (int32_t rp0 = lane; rp0 < L; rp0 += kUD * kWave)
    {
        int32_t cur[kUD], head[kUD], ic[kUD], oc_head[kUD], be[kUD][kBatch];
        uint16_t cov[kUD], w[kUD], bw[kUD][kBatch];
#pragma unroll
        for (int u = 0; u < kUD; u++)
        {
            const int32_t rp = min(rp0 + u * kWave, L - 1); // clamped lanes load valid addresses and store nothing
            cur[u]  = curr[rp];
            head[u] = rp > 0 ? (int32_t)curr[rp - 1] : cur[u];
            w[u]    = rp > 0 ? (uint16_t)((uint16_t)base_weights[rp - 1] + base_weights[rp]) : (uint16_t)0;
        }
#pragma unroll
        for (int u = 0; u < kUD; u++)
        {
            cov[u]     = g.coverage[cur[u]]; // zeroed in pass C for new nodes
            ic[u]      = in_count_of(cur[u]);
            oc_head[u] = out_count_of(head[u]); // heads are distinct across rp, nobody else bumps it
#pragma unroll
            for (int e = 0; e < kBatch; e++)
            {
                be[u][e] = g.incoming_edges[(int64_t)cur[u] * kEdges + e];
                bw[u][e] = g.incoming_edge_w[(int64_t)cur[u] * kEdges + e];
            }
        }
#pragma unroll
        for (int u = 0; u < kUD; u++)
        {
            const int32_t rp = rp0 + u * kWave;
            if (rp >= L) continue;
            if (rp > 0)
            {
                bool exists = false;
#pragma unroll
                for (int e = 0; e < kBatch; e++)
                    if (e < ic[u] && be[u][e] == head[u])
                    {
                        exists = true;
                        g.incoming_edge_w[(int64_t)cur[u] * kEdges + e] = (uint16_t)(bw[u][e] + w[u]);
                    }
                for (int32_t e = kBatch; e < ic[u]; e++) // longer in-edge lists loop on
                {
                    if (g.incoming_edges[(int64_t)cur[u] * kEdges + e] == head[u])
                    {
                        exists = true;
                        g.incoming_edge_w[(int64_t)cur[u] * kEdges + e] += w[u];
                    }
                }
                if (!exists)
                {
                    g.incoming_edges[(int64_t)cur[u] * kEdges + ic[u]]  = (IdT)head[u];
                    g.incoming_edge_w[(int64_t)cur[u] * kEdges + ic[u]] = w[u];
                    g.incoming_edge_count[cur[u]]                       = (uint16_t)(ic[u] + 1);
                    const int32_t oc                                    = oc_head[u];
                    g.outgoing_edges[(int64_t)head[u] * kEdges + oc]    = (IdT)cur[u];
                    if (MSA)
                    {
                        g.out_cov_cnt[(int64_t)head[u] * kEdges + oc]                       = 1;
                        g.out_cov[((int64_t)head[u] * kEdges + oc) * max_sequences_per_poa] = s;
                    }
                    g.outgoing_edge_count[head[u]] = (uint16_t)(oc + 1);
                    if (oc + 1 >= kEdges || ic[u] + 1 >= kEdges) rp_edgeerr = min(rp_edgeerr, rp);
                }
                else if (MSA)
                {
                    const int32_t oc = oc_head[u];
                    for (int32_t e = 0; e < oc; e++)
                    {
                        if (g.outgoing_edges[(int64_t)head[u] * kEdges + e] == cur[u])
                        {
                            const uint16_t cc = g.out_cov_cnt[(int64_t)head[u] * kEdges + e];
                            g.out_cov[((int64_t)head[u] * kEdges + e) * max_sequences_per_poa + cc] = s;
                            g.out_cov_cnt[(int64_t)head[u] * kEdges + e]                            = cc + 1;
                            break;
                        }
                    }
                }
            }
            else if (MSA)
                *sequence_begin_nodes_ids = (IdT)cur[u];
            g.coverage[cur[u]] = (uint16_t)(cov[u] + 1);
        }
    }
    for (int off = 32; off > 0; off >>= 1) rp_edgeerr = min(rp_edgeerr, __shfl_xor(rp_edgeerr, off));
    wave_sync();
    prof_mark(4);
    if (rp_edgeerr != INT32_MAX) return (int32_t)kEdgeCountExceeded;
    new_node_count = node_count + running;
    return 0;
}

// Kahn order: sources by ascending node id, children in outgoing-slot order, queue == output array.
template <typename IdT>
__device__ void topsort_kahn(IdT* sorted_poa, IdT* node_map, int32_t node_count, const uint16_t* incoming_edge_count,
                             const IdT* outgoing_edges, const uint16_t* outgoing_edge_count, uint16_t* local_cnt)
{
    int32_t position = 0;
    for (int32_t n = 0; n < node_count; n++)
    {
        uint16_t c   = incoming_edge_count[n];
        local_cnt[n] = c;
        if (c == 0)
        {
            node_map[n]            = (IdT)position;
            sorted_poa[position++] = (IdT)n;
        }
    }
    for (int32_t n = 0; n < position; n++)
    {
        int32_t node = sorted_poa[n];
        uint16_t oc  = outgoing_edge_count[node];
        for (int32_t e = 0; e < oc; e++)
        {
            int32_t out_node = outgoing_edges[(int64_t)node * kEdges + e];
            uint16_t c       = local_cnt[out_node];
            if (--c == 0)
            {
                node_map[out_node]     = (IdT)position;
                sorted_poa[position++] = (IdT)out_node;
            }
            local_cnt[out_node] = c;
        }
    }
}

// Kahn order for graphs whose tables live in HBM (long reads: up to ~100 k nodes, 32-bit ids), same output as
// topsort_kahn. The order-defining loop is a pointer chase -- pop a node, read its out-edges, decrement the children's
// counters -- and with the graph in HBM every dependent step was a cold round trip (about 2 000 cycles per node on
// one lane). The chase itself cannot be parallelised (FIFO Kahn order is defined by it), but what it will touch can be
// predicted: one read changes the graph in a few places, so the new order is almost the previous order, and the nodes
// that follow node u in the PREVIOUS order are the ones the loop is about to pop and to decrement. They are fetched 64
// at a time by all lanes (one round trip) into a direct-mapped LDS cache of node records, so the serial loop -- run
// wave-uniformly on the scalar unit -- pays LDS latency per step. The cache is write-through for the in-edge counters
// (local_cnt in HBM always holds the current value), so an eviction loses nothing and a refill reads the truth.
//   LDS (the forward pass's ring, idle during the sort): records[1024] {node, out-degree | unvisited in-edges << 16,
//   out-edge 0, out-edge 1} | previous position[1024] | the queue's most recent 2048 entries.
//   scratch (HBM, 2 x n_old int32: the score matrix, dead until the next forward pass): previous order and positions.
template <typename IdT>
__device__ __forceinline__ void topsort_kahn_cached(const GraphView<IdT>& g, int32_t n_old, int32_t node_count, uint8_t* lds,
                                                    int32_t* scratch, int lane)
{
    constexpr int32_t kSlots = 1024, kQueue = 2048, kAhead = 16;
    uint4* rec        = reinterpret_cast<uint4*>(lds);
    int32_t* rec_pos  = reinterpret_cast<int32_t*>(lds + kSlots * sizeof(uint4));
    int32_t* qring    = rec_pos + kSlots;
    int32_t* old_order = scratch;
    int32_t* old_pos   = scratch + ((n_old + 63) & ~63);
    for (int32_t i = lane; i < kSlots; i += kWave) rec[i] = make_uint4(0xffffffffu, 0u, 0u, 0u);
    // pre-pass (all lanes): counters, previous order / positions into scratch, sources in ascending node id
    int32_t tail = 0;
    for (int32_t base = 0; base < node_count; base += kWave)
    {
        const int32_t n = base + lane;
        bool is_src     = false;
        if (n < node_count)
        {
            const uint16_t c = g.incoming_edge_count[n];
            g.local_cnt[n]   = c;
            is_src           = c == 0;
            if (n < n_old)
            {
                old_order[n] = (int32_t)g.sorted_poa[n];
                old_pos[n]   = (int32_t)g.node_id_to_pos[n];
            }
        }
        const unsigned long long m = __ballot(is_src);
        wave_sync(); // old_order / old_pos of this chunk are read before the sources below overwrite sorted_poa / node_id_to_pos
        if (is_src)
        {
            const int32_t at     = tail + __popcll(m & ((1ull << lane) - 1));
            g.sorted_poa[at]     = (IdT)n;
            g.node_id_to_pos[n]  = (IdT)at;
            qring[at & (kQueue - 1)] = n;
        }
        tail += __popcll(m);
    }
    wave_sync();
    // fills the record of `node` (all lanes pass the same node): returns the record
    auto fill_one = [&](int32_t node) -> uint4 {
        const uint32_t oc = g.outgoing_edge_count[node];
        const uint32_t ic = g.local_cnt[node];
        const uint32_t e0 = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node * kEdges];
        const uint32_t e1 = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node * kEdges + 1];
        const int32_t op  = node < n_old ? old_pos[node] : -1;
        const uint4 r     = make_uint4((uint32_t)node, (oc & 0xffffu) | (ic << 16), e0, e1);
        if (lane == 0)
        {
            rec[node & (kSlots - 1)]     = r;
            rec_pos[node & (kSlots - 1)] = op;
        }
        asm volatile("" ::: "memory"); // LDS executes one wavefront's operations in order: no wait needed
        return r;
    };
    auto lookup = [&](int32_t node, int32_t& oldp) -> uint4 {
        const uint4 r  = rec[node & (kSlots - 1)];
        const int32_t q = rec_pos[node & (kSlots - 1)];
        uint4 u;
        u.x = (uint32_t)wave_first((int32_t)r.x); u.y = (uint32_t)wave_first((int32_t)r.y);
        u.z = (uint32_t)wave_first((int32_t)r.z); u.w = (uint32_t)wave_first((int32_t)r.w);
        oldp = wave_first(q);
        if ((int32_t)u.x != node)
        {
            u    = fill_one(node);
            u.x = (uint32_t)wave_first((int32_t)u.x); u.y = (uint32_t)wave_first((int32_t)u.y);
            u.z = (uint32_t)wave_first((int32_t)u.z); u.w = (uint32_t)wave_first((int32_t)u.w);
            oldp = node < n_old ? wave_first(old_pos[node]) : -1;
        }
        return u;
    };
    // the nodes at previous positions [from, from + 64): one lane each, records that are already cached stay as they are
    auto prefetch = [&](int32_t from) {
        const int32_t p = from + lane;
        if (p < n_old)
        {
            const int32_t node = old_order[p];
            const uint32_t oc  = g.outgoing_edge_count[node];
            const uint32_t ic  = g.local_cnt[node];
            const uint32_t e0  = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node * kEdges];
            const uint32_t e1  = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node * kEdges + 1];
            const uint32_t tag = rec[node & (kSlots - 1)].x;
            if (tag != (uint32_t)node)
            {
                rec[node & (kSlots - 1)] = make_uint4((uint32_t)node, (oc & 0xffffu) | (ic << 16), e0, e1);
                asm volatile("" ::: "memory");
                // two lanes of one block may map to the same slot: whichever record landed owns the position word too
                if (rec[node & (kSlots - 1)].x == (uint32_t)node) rec_pos[node & (kSlots - 1)] = p;
            }
        }
        asm volatile("" ::: "memory");
    };
    int32_t pref_end = 0;
    prefetch(0);
    pref_end = kWave;
    // the FIFO loop, wave-uniform
    for (int32_t n = 0; n < tail; n++)
    {
        int32_t node;
        if (tail - n <= kQueue) node = wave_first(qring[n & (kQueue - 1)]);
        else node = wave_first((int32_t)g.sorted_poa[n]); // a queue longer than the ring: its old entries from HBM
        int32_t oldp;
        const uint4 r = lookup(node, oldp);
        if (oldp >= 0 && oldp + kAhead >= pref_end && pref_end < n_old)
        {
            const int32_t from = max(pref_end, oldp + 1);
            prefetch(from);
            pref_end = from + kWave;
        }
        const int32_t oc = (int32_t)(r.y & 0xffffu);
        for (int32_t e = 0; e < oc; e++)
        {
            const int32_t child = e == 0 ? (int32_t)r.z : (e == 1 ? (int32_t)r.w
                                 : wave_first((int32_t)g.outgoing_edges[(int64_t)node * kEdges + e]));
            int32_t cp;
            const uint4 cr      = lookup(child, cp);
            const uint32_t left = (cr.y >> 16) - 1u;
            if (lane == 0)
            {
                rec[child & (kSlots - 1)].y = (cr.y & 0xffffu) | (left << 16);
                g.local_cnt[child]          = (uint16_t)left; // write-through
            }
            if ((left & 0xffffu) == 0)
            {
                if (lane == 0)
                {
                    g.sorted_poa[tail]          = (IdT)child;
                    g.node_id_to_pos[child]     = (IdT)tail;
                    qring[tail & (kQueue - 1)]  = child;
                }
                tail++;
            }
            asm volatile("" ::: "memory"); // (no wait: the counter's write-through store must not stall the chase)
        }
    }
    wave_sync();
}

// Kahn with its working set in LDS (node counts <= 3072, 16-bit ids): the order-defining serial loop then pays
// LDS latency per dependent step instead of an HBM round trip. Same output as topsort_kahn.
//   ent[n]  : one 64-bit word per node (first two out-edges, out-degree, unvisited in-edges)   24 KB (row-table region)
//   queue[] : the FIFO == the sorted order                                                       6 KB (score-ring region)
template <typename IdT>
__device__ __forceinline__ void topsort_kahn_lds(const GraphView<IdT>& g, int32_t node_count, uint8_t* lds,
                                                 uint8_t* lds_queue, int lane, int32_t dbg = 0, uint64_t* prof_acc = nullptr)
{
    // per node one 64-bit LDS word: [0:16) out-edge 0  [16:32) out-edge 1  [32:40) out-degree  [40:48) unvisited in-edges
    uint64_t* ent   = reinterpret_cast<uint64_t*>(lds);
    uint16_t* queue = reinterpret_cast<uint16_t*>(lds_queue);
    // phase 1 (all lanes): stage the node words; sources in ascending node id (ordered compaction per 64-chunk)
    int32_t tail = 0;
    for (int32_t base = 0; base < node_count; base += kWave)
    {
        const int32_t n = base + lane;
        bool is_src     = false;
        if (n < node_count)
        {
            // four independent loads, one HBM round trip (slots past the out-degree hold stale ids and are masked)
            const uint32_t ic = g.incoming_edge_count[n];
            const uint32_t oc = g.outgoing_edge_count[n];
            const uint32_t e0 = (uint16_t)g.outgoing_edges[(int64_t)n * kEdges];
            const uint32_t e1 = (uint16_t)g.outgoing_edges[(int64_t)n * kEdges + 1];
            const uint32_t e  = (oc > 0 ? e0 : 0u) | (oc > 1 ? e1 << 16 : 0u);
            ent[n] = (uint64_t)e | ((uint64_t)((oc & 0xff) | ((ic & 0xff) << 8)) << 32);
            is_src = (ic == 0);
        }
        const unsigned long long m = __ballot(is_src);
        if (is_src) queue[tail + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)n;
        tail += __popcll(m);
    }
    wave_sync();
    // phase 2: the FIFO loop, executed wave-uniformly (every lane runs the same scalar program, so node words land
    // in SGPRs). On a lone wavefront a taken branch costs ~25 cycles and an LDS round trip ~55 (tools/microbench.hip),
    // so the loop body is written branch-light: the first child is handled with selects and unconditional stores
    // (a dummy node word absorbs the stores of childless nodes), and a child whose last in-edge was just consumed
    // keeps its word in registers, so along a chain (the common case) an iteration is one LDS round trip.
    {
        constexpr int32_t kDummy = 3073; // spare word of the 3074-entry region
        int32_t head      = 0;
        int32_t pend_node = -1;
        uint32_t pend_lo = 0, pend_hi = 0;
        while (head < tail)
        {
            int32_t node   = pend_node;
            uint32_t lo = pend_lo, hi = pend_hi;
            if (!(head == tail - 1 && pend_node >= 0))
            {
                node             = wave_first((int32_t)queue[head]);
                const uint64_t w = wave_first64(ent[node]);
                lo = (uint32_t)w; hi = (uint32_t)(w >> 32);
            }
            head++;
            const int32_t oc  = (int32_t)(hi & 0xff);
            const int32_t c0  = oc > 0 ? (int32_t)(lo & 0xffff) : kDummy;
            const uint64_t w0 = wave_first64(ent[c0]);
            const uint32_t chi0  = (uint32_t)(w0 >> 32);
            const uint32_t left0 = ((chi0 >> 8) - 1) & 0xff;
            const bool push0     = oc > 0 && left0 == 0;
            lane0_store_u8(reinterpret_cast<uint8_t*>(ent + c0) + 5, left0); // a count that reached 0 is never read again
            lane0_store_u16(queue + tail, (uint32_t)c0);                    // only becomes part of the queue if tail advances
            tail += push0 ? 1 : 0;
            pend_node = push0 ? c0 : pend_node;
            pend_lo   = push0 ? (uint32_t)w0 : pend_lo;
            pend_hi   = push0 ? (chi0 & 0xff) : pend_hi;
            for (int32_t k = 1; k < oc; k++) // further children (bubble openings)
            {
                const int32_t child = k == 1 ? (int32_t)(lo >> 16)
                                             : wave_first((int32_t)g.outgoing_edges[(int64_t)node * kEdges + k]);
                const uint64_t cw   = wave_first64(ent[child]);
                const uint32_t chi  = (uint32_t)(cw >> 32);
                const uint32_t left = ((chi >> 8) - 1) & 0xff;
                lane0_store_u8(reinterpret_cast<uint8_t*>(ent + child) + 5, left);
                if (left == 0)
                {
                    lane0_store_u16(queue + tail, (uint32_t)child);
                    tail++;
                    pend_node = child;
                    pend_lo   = (uint32_t)cw;
                    pend_hi   = chi & 0xff;
                }
            }
        }
    }
    wave_sync();
    // phase 3 (all lanes): publish order and inverse map
    for (int32_t i = lane; i < node_count; i += kWave)
    {
        const int32_t node    = queue[i];
        g.sorted_poa[i]       = (IdT)node;
        g.node_id_to_pos[node] = (IdT)i;
    }
}

// ------------------------------------------------------------------------------------------------
// Incremental Kahn order. The reference re-sorts the whole graph after every read (cudapoa_kernels.cuh:516-531),
// but one read changes the graph in a few places only, and Kahn's FIFO order is a deterministic replay: if, when
// position p of the PREVIOUS order sigma is about to be popped, (1) the old nodes output so far are exactly
// sigma[0..p) and (2) the queue holds exactly what the previous run's queue held at that moment, then popping a
// node without a new out-edge whose children have no new in-edge does what the previous run did. Such steps are
// taken up to 64 at a time (one lane per step; in-edge counters decremented with LDS atomics; order copied from
// sigma), every other step is an ordinary Kahn step, and the state is in sync again as soon as (1) and (2) hold.
// Output identical to topsort_kahn; the algorithm is modelled and checked against the plain restatement on CPU
// (oracle/topsort_incr_model.inc, tests/test_oracle_poa.py).
//   per node, carried from read to read in GraphView::local_cnt (uint16): queue length when the node was popped
//   (4 bits, 15 = "15 or more": never a sync point) | out-degree << 4 | in-degree << 10;
//   node word lo: out-edge 0 [0:12)  out-edge 1 [12:24)  min(out-degree, 4) [24:27)  previous queue length [27:31)
//             hi: unvisited in-edges [0:6)  new in-edge / new node [6]  new out-edge / new node [7]
//                 previous position [8:20)  out-edge 2 [20:32)
//   (three out-edges in the word; out-edges 3..5 of the few nodes that have more sit in a 256-slot direct-mapped LDS
//   table keyed by node id: {node [0:12), valid [12], out-degree [13:19), edge 3 [20:32), edge 4 [32:44), edge 5 [44:56)};
//   a node that lost its slot or has more than six out-edges reads the rest of its list from HBM in an ordinary step)
//   LDS: ent (row-table region), queue + wide-node table (score-ring region), previous order (read + trace-code-tile
//   regions, 6 KB).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ti_inleft(uint32_t h) { return h & 0x3fu; }
__device__ __forceinline__ uint32_t ti_din(uint32_t h) { return (h >> 6) & 1u; }
__device__ __forceinline__ uint32_t ti_dout(uint32_t h) { return (h >> 7) & 1u; }
__device__ __forceinline__ int32_t ti_pos(uint32_t h) { return (int32_t)((h >> 8) & 0xfffu); }
__device__ __forceinline__ int32_t ti_e2(uint32_t h) { return (int32_t)(h >> 20); }
__device__ __forceinline__ int32_t ti_ocq(uint32_t l) { return (int32_t)((l >> 24) & 7u); }
__device__ __forceinline__ int32_t ti_qlen(uint32_t l) { return (int32_t)((l >> 27) & 15u); }
__device__ __forceinline__ uint32_t lds_dec_u32(uint32_t* p) // returns the old value
{
    return __hip_atomic_fetch_sub(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <typename IdT>
__device__ __forceinline__ void topsort_kahn_incr_lds(const GraphView<IdT>& g, int32_t n_old, int32_t node_count,
                                                      uint8_t* lds, uint8_t* lds_queue, uint8_t* lds_old, int lane,
                                                      int32_t dbg = 0, uint64_t* prof_acc = nullptr)
{
    // profiling (GWHIP_DEBUG bits 25-27): cycles of 1 phase 1, 2 replayed blocks, 3 ordinary steps, 4 phase 3;
    // 5 number of blocks x 1000, 6 number of ordinary steps x 1000
    const int32_t tsel = prof_acc ? (dbg >> 25) & 7 : 0;
    uint64_t tacc      = 0;
    const uint64_t t_p1 = tsel == 1 ? clock64() : 0;
    constexpr int32_t kDummy = 3073; // spare word of the 3074-entry region
    constexpr int32_t kQClip = 15;
    uint64_t* ent   = reinterpret_cast<uint64_t*>(lds);
    uint32_t* ent32 = reinterpret_cast<uint32_t*>(lds);
    uint16_t* queue = reinterpret_cast<uint16_t*>(lds_queue);
    uint16_t* sold  = reinterpret_cast<uint16_t*>(lds_old);
    uint64_t* wide  = reinterpret_cast<uint64_t*>(lds_queue + 6144 + 16); // 256 slots behind the 3072-entry queue (+ its spill slot)
    auto hi_of = [&](int32_t n) -> uint32_t { return ent32[2 * n + 1]; };
    auto wide_hit = [&](uint64_t we, int32_t n) -> bool {
        return (int32_t)(we & 0xfffu) == n && ((we >> 12) & 1u) != 0 && ((we >> 13) & 63u) <= 6u;
    };
#pragma unroll
    for (int q = 0; q < 4; q++) wide[q * kWave + lane] = 0; // entries of the previous read are stale
    wave_sync();
    // phase 1 (all lanes): node words with change flags, previous order into LDS, sources in ascending node id.
    // Four 64-node chunks per iteration share one HBM round trip (eight independent loads per node; edge slots
    // past the out-degree hold stale ids: masked).
    int32_t tail = 0;
    constexpr int kU = 4;
    for (int32_t base = 0; base < node_count; base += kU * kWave)
    {
        uint32_t ic[kU], oc[kU], e0[kU], e1[kU], e2[kU], e3[kU], e4[kU], e5[kU], m[kU], po[kU], so[kU];
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            const int32_t n  = min(base + u * kWave + lane, node_count - 1);
            const int32_t nn = n >= n_old ? 0 : n;
            ic[u] = g.incoming_edge_count[n];
            oc[u] = g.outgoing_edge_count[n];
            e0[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges] & 0xfffu;
            e1[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges + 1] & 0xfffu;
            e2[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges + 2] & 0xfffu;
            e3[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges + 3] & 0xfffu;
            e4[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges + 4] & 0xfffu;
            e5[u] = (uint32_t)g.outgoing_edges[(int64_t)n * kEdges + 5] & 0xfffu;
            m[u]  = g.local_cnt[nn];
            po[u] = (uint32_t)g.node_id_to_pos[nn] & 0xfffu;
            so[u] = (uint32_t)g.sorted_poa[nn] & 0xfffu;
        }
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            const int32_t n = base + u * kWave + lane;
            bool is_src     = false;
            if (n < node_count)
            {
                const bool is_new = n >= n_old;
                if (!is_new) sold[n] = (uint16_t)so[u];
                const uint32_t din  = (is_new || ((m[u] >> 10) & 63u) != ic[u]) ? 1u : 0u;
                const uint32_t dout = (is_new || ((m[u] >> 4) & 63u) != oc[u]) ? 1u : 0u;
                const uint32_t lo   = (oc[u] > 0 ? e0[u] : 0u) | (oc[u] > 1 ? e1[u] << 12 : 0u) | (min(oc[u], 4u) << 24) |
                                    (is_new ? 0u : (m[u] & 15u) << 27);
                const uint32_t hi = (ic[u] & 0x3fu) | (din << 6) | (dout << 7) | (is_new ? 0u : po[u] << 8) | (oc[u] > 2 ? e2[u] << 20 : 0u);
                ent[n] = (uint64_t)lo | ((uint64_t)hi << 32);
                if (oc[u] > 3) // last writer of a slot wins; the others fall back to the HBM list
                    wide[n & 255] = (uint64_t)((uint32_t)n | (1u << 12) | (oc[u] << 13) | (e3[u] << 20)) | ((uint64_t)e4[u] << 32) |
                                    ((uint64_t)e5[u] << 44);
                is_src = (ic[u] == 0);
            }
            const unsigned long long ms = __ballot(is_src);
            if (is_src) queue[tail + __popcll(ms & ((1ull << lane) - 1))] = (uint16_t)n;
            tail += __popcll(ms);
        }
    }
    wave_sync();
    if (tsel == 1) tacc += clock64() - t_p1;
    // phase 2: wave-uniform control; k = new nodes output so far, M = highest previous position output so far
    int32_t head = 0, k = 0, M = -1;
    while (head < tail)
    {
        const uint64_t t_it = (tsel == 2 || tsel == 3) ? clock64() : 0;
        const int32_t u   = wave_first((int32_t)queue[head]) & 0xfff;
        const uint64_t w  = wave_first64(ent[u]);
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        const int32_t ocq = ti_ocq(lo);
        const int32_t c0  = ocq > 0 ? (int32_t)(lo & 0xfffu) : kDummy;
        const int32_t c1  = ocq > 1 ? (int32_t)((lo >> 12) & 0xfffu) : kDummy;
        const int32_t c2  = ocq > 2 ? ti_e2(hi) : kDummy;
        const uint32_t h0 = (uint32_t)wave_first((int32_t)hi_of(c0));
        const uint32_t h1 = (uint32_t)wave_first((int32_t)hi_of(c1));
        const uint32_t h2 = (uint32_t)wave_first((int32_t)hi_of(c2));
        const int32_t len = tail - head;
        const int32_t p = ti_pos(hi), qo = ti_qlen(lo);
        const bool is_new = u >= n_old;
        // an ordinary step is needed when the node has a new out-edge or a child has a new in-edge
        bool need_real = is_new | (ti_dout(hi) != 0) | ((ocq > 0) & (ti_din(h0) != 0)) | ((ocq > 1) & (ti_din(h1) != 0)) |
                         ((ocq > 2) & (ti_din(h2) != 0));
        // more than three out-edges (about 1.6 % of the nodes): edges 3..5 from the wide-node table
        uint64_t uw = 0;
        bool uw_ok  = false;
        int32_t uoc = ocq;
        if (ocq > 3)
        {
            uw    = wave_first64(wide[u & 255]);
            uw_ok = wide_hit(uw, u);
            uoc   = uw_ok ? (int32_t)((uw >> 13) & 63u) : 4;
            if (uw_ok)
            {
                const int32_t x3 = (int32_t)((uw >> 20) & 0xfffu), x4 = uoc > 4 ? (int32_t)((uw >> 32) & 0xfffu) : x3,
                              x5 = uoc > 5 ? (int32_t)((uw >> 44) & 0xfffu) : x3;
                const uint32_t d = ti_din((uint32_t)wave_first((int32_t)hi_of(x3))) | ti_din((uint32_t)wave_first((int32_t)hi_of(x4))) |
                                   ti_din((uint32_t)wave_first((int32_t)hi_of(x5)));
                need_real |= d != 0;
            }
            else
                need_real = true;
        }
        bool block = !need_real && (head - k == p) && (M == p - 1) && (len == qo) && (qo < kQClip);
        if (block && len > 1) // the queue must be the previous run's queue at p, element by element
        {
            const int32_t qi  = lane < len ? (int32_t)(queue[head + lane] & 0xfff) : u;
            const uint32_t qh = hi_of(qi);
            const bool ok     = lane >= len || (qi < n_old && ti_pos(qh) == p + lane);
            block             = __ballot(!ok) == 0;
        }
        if (block)
        {
            // lane l replays the pop of position p + l of the previous order
            const int32_t posl = p + lane;
            const bool valid   = posl < n_old;
            const int32_t node = valid ? (int32_t)sold[posl] : u;
            const uint64_t wn  = ent[node];
            const uint32_t nlo = (uint32_t)wn, nhi = (uint32_t)(wn >> 32);
            const int32_t noc  = ti_ocq(nlo);
            const int32_t ch0  = noc > 0 ? (int32_t)(nlo & 0xfffu) : node;
            const int32_t ch1  = noc > 1 ? (int32_t)((nlo >> 12) & 0xfffu) : node;
            const int32_t ch2  = noc > 2 ? ti_e2(nhi) : node;
            const uint32_t g0 = hi_of(ch0), g1 = hi_of(ch1), g2 = hi_of(ch2);
            bool bad = !valid | (ti_dout(nhi) != 0) | ((lane > 0) & (ti_din(nhi) != 0)) | ((noc > 0) & (ti_din(g0) != 0)) |
                       ((noc > 1) & (ti_din(g1) != 0)) | ((noc > 2) & (ti_din(g2) != 0));
            // more than three out-edges: edges 3..5 from the wide-node table; a lane whose node is not there ends
            // the block (its own step is an ordinary one)
            const bool is_wide   = valid && noc > 3;
            const bool any_wide  = __ballot(is_wide) != 0;
            int32_t woc = 0, ch3 = node, ch4 = node, ch5 = node;
            if (any_wide)
            {
                const uint64_t we = wide[node & 255];
                if (is_wide)
                {
                    if (wide_hit(we, node))
                    {
                        woc = (int32_t)((we >> 13) & 63u);
                        ch3 = (int32_t)((we >> 20) & 0xfffu);
                        ch4 = woc > 4 ? (int32_t)((we >> 32) & 0xfffu) : node;
                        ch5 = woc > 5 ? (int32_t)((we >> 44) & 0xfffu) : node;
                    }
                    else
                        bad = true;
                }
                const uint32_t g3 = hi_of(ch3), g4 = hi_of(ch4), g5 = hi_of(ch5);
                bad |= ((woc > 3) & (ti_din(g3) != 0)) | ((woc > 4) & (ti_din(g4) != 0)) | ((woc > 5) & (ti_din(g5) != 0));
            }
            const unsigned long long mb = __ballot(bad);
            const int32_t b = mb ? __ffsll(mb) - 1 : kWave; // >= 1: lane 0 passed the test above
            // all decrements in flight together; a child whose counter reaches 0 was pushed by the previous run here
            const bool do0 = lane < b && noc > 0, do1 = lane < b && noc > 1, do2 = lane < b && noc > 2;
            uint32_t r0 = 0, r1 = 0, r2 = 0;
            if (do0) r0 = lds_dec_u32(ent32 + 2 * ch0 + 1);
            if (do1) r1 = lds_dec_u32(ent32 + 2 * ch1 + 1);
            if (do2) r2 = lds_dec_u32(ent32 + 2 * ch2 + 1);
            int32_t npush = __popcll(__ballot(do0 && ti_inleft(r0) == 1u)) + __popcll(__ballot(do1 && ti_inleft(r1) == 1u)) +
                            __popcll(__ballot(do2 && ti_inleft(r2) == 1u));
            if (any_wide)
            {
                const bool do3 = lane < b && woc > 3, do4 = lane < b && woc > 4, do5 = lane < b && woc > 5;
                uint32_t r3 = 0, r4 = 0, r5 = 0;
                if (do3) r3 = lds_dec_u32(ent32 + 2 * ch3 + 1);
                if (do4) r4 = lds_dec_u32(ent32 + 2 * ch4 + 1);
                if (do5) r5 = lds_dec_u32(ent32 + 2 * ch5 + 1);
                npush += __popcll(__ballot(do3 && ti_inleft(r3) == 1u)) + __popcll(__ballot(do4 && ti_inleft(r4) == 1u)) +
                         __popcll(__ballot(do5 && ti_inleft(r5) == 1u));
            }
            // pushed entries continue the previous order; slots popped inside this same block are written by the
            // popping lane (with their queue length), the others here: disjoint slots
            for (int32_t j = lane; j < npush; j += kWave)
                if (tail + j >= head + b) queue[tail + j] = sold[p + len + j];
            if (lane < b) queue[head + lane] = (uint16_t)((uint32_t)node | ((uint32_t)ti_qlen(nlo) << 12));
            head += b;
            tail += npush;
            M = p + b - 1;
            asm volatile("" ::: "memory"); // LDS executes one wavefront's operations in order: no wait, no barrier
            if (tsel == 2) tacc += clock64() - t_it;
            if (tsel == 5) tacc += 1000;
            continue;
        }
        // ordinary Kahn step
        lane0_store_u16(queue + head, (uint32_t)u | ((uint32_t)min(len, kQClip) << 12));
        head++;
        k += is_new ? 1 : 0;
        M = is_new ? M : max(M, p);
        {
            const uint32_t left0 = (ti_inleft(h0) - 1u) & 0x3fu;
            lane0_store_u8(reinterpret_cast<uint8_t*>(ent + c0) + 4, (h0 & 0xc0u) | left0); // the dummy word absorbs it when ocq == 0
            lane0_store_u16(queue + tail, (uint32_t)c0);                                   // only part of the queue if tail advances
            tail += (ocq > 0 && left0 == 0) ? 1 : 0;
        }
        if (ocq > 1)
        {
            const uint32_t left1 = (ti_inleft(h1) - 1u) & 0x3fu;
            lane0_store_u8(reinterpret_cast<uint8_t*>(ent + c1) + 4, (h1 & 0xc0u) | left1);
            lane0_store_u16(queue + tail, (uint32_t)c1);
            tail += left1 == 0 ? 1 : 0;
            if (ocq > 2)
            {
                const uint32_t left2 = (ti_inleft(h2) - 1u) & 0x3fu;
                lane0_store_u8(reinterpret_cast<uint8_t*>(ent + c2) + 4, (h2 & 0xc0u) | left2);
                lane0_store_u16(queue + tail, (uint32_t)c2);
                tail += left2 == 0 ? 1 : 0;
                if (ocq > 3) // the rest of the list: the wide-node table, or (rare) HBM
                {
                    const int32_t oc = uw_ok ? uoc : wave_first((int32_t)g.outgoing_edge_count[u]);
                    for (int32_t e = 3; e < oc; e++)
                    {
                        const int32_t child = uw_ok ? (int32_t)((uw >> (20 + 12 * (e - 3))) & 0xfffu)
                                                    : wave_first((int32_t)g.outgoing_edges[(int64_t)u * kEdges + e]) & 0xfff;
                        const uint32_t hc   = (uint32_t)wave_first((int32_t)hi_of(child));
                        const uint32_t left = (ti_inleft(hc) - 1u) & 0x3fu;
                        lane0_store_u8(reinterpret_cast<uint8_t*>(ent + child) + 4, (hc & 0xc0u) | left);
                        lane0_store_u16(queue + tail, (uint32_t)child);
                        tail += left == 0 ? 1 : 0;
                    }
                }
            }
        }
        if (tsel == 3) tacc += clock64() - t_it;
        if (tsel == 6) tacc += 1000;
    }
    wave_sync();
    const uint64_t t_p3 = tsel == 4 ? clock64() : 0;
    // phase 3 (all lanes): publish order, inverse map and the per-node record for the next read
    for (int32_t i0 = lane; i0 < node_count; i0 += 4 * kWave) // four chunks per HBM round trip
    {
        uint32_t e[4], oc[4], ic[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            e[u] = queue[min(i0 + u * kWave, node_count - 1)];
            const int32_t node = (int32_t)(e[u] & 0xfffu);
            oc[u] = g.outgoing_edge_count[node];
            ic[u] = g.incoming_edge_count[node];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int32_t i = i0 + u * kWave;
            if (i < node_count)
            {
                const int32_t node     = (int32_t)(e[u] & 0xfffu);
                g.sorted_poa[i]        = (IdT)node;
                g.node_id_to_pos[node] = (IdT)i;
                g.local_cnt[node]      = (uint16_t)((e[u] >> 12) | (oc[u] << 4) | (ic[u] << 10));
            }
        }
    }
    if (tsel == 4)
    {
        wave_sync();
        tacc += clock64() - t_p3;
    }
    if (tsel && lane == 0) *prof_acc += tacc;
}

// ------------------------------------------------------------------------------------------------
// The incremental Kahn order (topsort_kahn_incr_lds above: replay of the previous run, 64 positions at a time, wherever
// the state is in sync with it) for graphs beyond the LDS tables: same algorithm, its working set in HBM scratch (the
// score matrix, dead between the merge and the next forward pass) with ids of up to 17 bits:
//   ent[n] (uint4): x, y, z = out-edges 0..2;  w = unvisited in-edges [0:6) | new in-edge or new node [6] | new out-edge or
//                   new node [7] | min(out-degree, 4) [8:11) | previous queue length [11:15) | previous position [15:32)
//   queue[] = the new order: a pushed entry is the node id, a popped one node | queue length at the pop << 28
//   sold[]  = the previous order
// The w words are the in-edge counters: they are only touched with device-scope atomics after the build pass (block replay
// decrements them from many lanes; a plain load could be served from a stale L1 line), everything else with plain loads and
// stores of this one wavefront. Nodes with more than three out-edges always take an ordinary step (1.6 % of the nodes).
// At long-read divergence the scalar model replays 86-88 % of a re-sort in blocks of 16-18 (tools/topsort_replay_stats.py).
// Per node carried from read to read in GraphView::local_cnt as in the LDS version (queue length | out-degree << 4 |
// in-degree << 10), so a window must use this routine for every read or for none.
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ void topsort_kahn_incr_hbm(const GraphView<IdT>& g, int32_t n_old, int32_t node_count, int32_t* scratch, int lane)
{
    constexpr int32_t kQClip = 15;
    constexpr uint32_t kIdMask = 0x0fffffffu;
    const int32_t n_pad = (node_count + 63) & ~63;
    uint4* ent     = reinterpret_cast<uint4*>(scratch);
    uint32_t* entw = reinterpret_cast<uint32_t*>(scratch);
    int32_t* queue = scratch + 4 * n_pad;
    int32_t* sold  = queue + n_pad + 64;
    auto load_w = [&](int32_t n) -> uint32_t { return __hip_atomic_load(entw + 4 * (size_t)n + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto dec_w  = [&](int32_t n) -> uint32_t { return __hip_atomic_fetch_sub(entw + 4 * (size_t)n + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto w_inleft = [](uint32_t w) -> uint32_t { return w & 0x3fu; };
    auto w_din    = [](uint32_t w) -> uint32_t { return (w >> 6) & 1u; };
    auto w_dout   = [](uint32_t w) -> uint32_t { return (w >> 7) & 1u; };
    auto w_ocq    = [](uint32_t w) -> int32_t { return (int32_t)((w >> 8) & 7u); };
    auto w_qlen   = [](uint32_t w) -> int32_t { return (int32_t)((w >> 11) & 15u); };
    auto w_pos    = [](uint32_t w) -> int32_t { return (int32_t)(w >> 15); };

    // phase 1 (all lanes): node words with change flags, previous order, sources in ascending node id
    int32_t tail = 0;
    for (int32_t base = 0; base < node_count; base += kWave)
    {
        const int32_t n  = base + lane;
        const int32_t nc = min(n, node_count - 1);
        const int32_t nn = nc >= n_old ? 0 : nc;
        const uint32_t ic = g.incoming_edge_count[nc], oc = g.outgoing_edge_count[nc];
        const uint32_t e0 = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)nc * kEdges];
        const uint32_t e1 = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)nc * kEdges + 1];
        const uint32_t e2 = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)nc * kEdges + 2];
        const uint32_t m  = g.local_cnt[nn];
        const uint32_t po = (uint32_t)(int32_t)g.node_id_to_pos[nn];
        const int32_t so  = (int32_t)g.sorted_poa[nn];
        bool is_src = false;
        if (n < node_count)
        {
            const bool is_new = n >= n_old;
            if (!is_new) sold[n] = so;
            const uint32_t din  = (is_new || ((m >> 10) & 63u) != ic) ? 1u : 0u;
            const uint32_t dout = (is_new || ((m >> 4) & 63u) != oc) ? 1u : 0u;
            const uint32_t w    = (ic & 0x3fu) | (din << 6) | (dout << 7) | (min(oc, 4u) << 8) | (is_new ? 0u : (m & 15u) << 11) | (is_new ? 0u : po << 15);
            ent[n] = make_uint4(oc > 0 ? e0 : 0u, oc > 1 ? e1 : 0u, oc > 2 ? e2 : 0u, w);
            is_src = ic == 0;
        }
        const unsigned long long ms = __ballot(is_src);
        if (is_src) queue[tail + __popcll(ms & ((1ull << lane) - 1))] = n;
        tail += __popcll(ms);
    }
    wave_sync();
    auto lane0_dec = [&](int32_t n) -> uint32_t { // one decrement, its old value to every lane
        uint32_t old = 0;
        if (lane == 0) old = dec_w(n);
        return (uint32_t)wave_first((int32_t)old);
    };
    // phase 2: wave-uniform control; k = new nodes output so far, M = highest previous position output so far
    int32_t head = 0, k = 0, M = -1;
    while (head < tail)
    {
        const int32_t u   = (int32_t)((uint32_t)wave_first(queue[head]) & kIdMask);
        const uint4 eu    = ent[u];
        const uint32_t hi = (uint32_t)wave_first((int32_t)load_w(u));
        const int32_t ocq = w_ocq(hi);
        const int32_t c0 = wave_first((int32_t)eu.x), c1 = wave_first((int32_t)eu.y), c2 = wave_first((int32_t)eu.z);
        const uint32_t h0 = ocq > 0 ? (uint32_t)wave_first((int32_t)load_w(c0)) : 0u;
        const uint32_t h1 = ocq > 1 ? (uint32_t)wave_first((int32_t)load_w(c1)) : 0u;
        const uint32_t h2 = ocq > 2 ? (uint32_t)wave_first((int32_t)load_w(c2)) : 0u;
        const int32_t len = tail - head;
        const int32_t p = w_pos(hi), qo = w_qlen(hi);
        const bool is_new = u >= n_old;
        // an ordinary step is needed when the node has a new out-edge, a child has a new in-edge, or it has more than three children
        const bool need_real = is_new | (w_dout(hi) != 0) | (ocq > 3) | ((ocq > 0) & (w_din(h0) != 0)) | ((ocq > 1) & (w_din(h1) != 0)) |
                               ((ocq > 2) & (w_din(h2) != 0));
        bool block = !need_real && (head - k == p) && (M == p - 1) && (len == qo) && (qo < kQClip);
        if (block && len > 1) // the queue must be the previous run's queue at p, element by element
        {
            const int32_t qi  = lane < len ? (int32_t)((uint32_t)queue[head + lane] & kIdMask) : u;
            const uint32_t qh = load_w(qi);
            const bool ok     = lane >= len || (qi < n_old && w_pos(qh) == p + lane);
            block             = __ballot(!ok) == 0;
        }
        if (block)
        {
            // lane l replays the pop of position p + l of the previous order
            const int32_t posl = p + lane;
            const bool valid   = posl < n_old;
            const int32_t node = valid ? sold[posl] : u;
            const uint4 en     = ent[node];
            const uint32_t nhi = load_w(node);
            const int32_t noc  = w_ocq(nhi);
            const int32_t ch0  = noc > 0 ? (int32_t)en.x : node;
            const int32_t ch1  = noc > 1 ? (int32_t)en.y : node;
            const int32_t ch2  = noc > 2 ? (int32_t)en.z : node;
            const uint32_t g0 = load_w(ch0), g1 = load_w(ch1), g2 = load_w(ch2);
            const bool bad = !valid | (w_dout(nhi) != 0) | (noc > 3) | ((lane > 0) & (w_din(nhi) != 0)) | ((noc > 0) & (w_din(g0) != 0)) |
                             ((noc > 1) & (w_din(g1) != 0)) | ((noc > 2) & (w_din(g2) != 0));
            const unsigned long long mb = __ballot(bad);
            const int32_t b = mb ? __ffsll((unsigned long long)mb) - 1 : kWave; // >= 1: lane 0 passed the test above
            // all decrements in flight together; a child whose counter reaches 0 was pushed by the previous run here
            const bool do0 = lane < b && noc > 0, do1 = lane < b && noc > 1, do2 = lane < b && noc > 2;
            uint32_t r0 = 0, r1 = 0, r2 = 0;
            if (do0) r0 = dec_w(ch0);
            if (do1) r1 = dec_w(ch1);
            if (do2) r2 = dec_w(ch2);
            const int32_t npush = __popcll(__ballot(do0 && w_inleft(r0) == 1u)) + __popcll(__ballot(do1 && w_inleft(r1) == 1u)) +
                                  __popcll(__ballot(do2 && w_inleft(r2) == 1u));
            // pushed entries continue the previous order; slots popped inside this same block are written by the popping lane
            // (with their queue length), the others here: disjoint slots
            for (int32_t j = lane; j < npush; j += kWave)
                if (tail + j >= head + b) queue[tail + j] = sold[p + len + j];
            if (lane < b) queue[head + lane] = (int32_t)((uint32_t)node | ((uint32_t)w_qlen(nhi) << 28));
            head += b;
            tail += npush;
            M = p + b - 1;
            continue;
        }
        // ordinary Kahn step
        if (lane == 0) queue[head] = (int32_t)((uint32_t)u | ((uint32_t)min(len, kQClip) << 28));
        head++;
        k += is_new ? 1 : 0;
        M = is_new ? M : max(M, p);
        const int32_t oc = ocq > 3 ? wave_first((int32_t)g.outgoing_edge_count[u]) : ocq;
        for (int32_t e = 0; e < oc; e++)
        {
            const int32_t child = e == 0 ? c0 : (e == 1 ? c1 : (e == 2 ? c2 : wave_first((int32_t)g.outgoing_edges[(int64_t)u * kEdges + e])));
            const uint32_t old  = lane0_dec(child);
            if (w_inleft(old) == 1u)
            {
                if (lane == 0) queue[tail] = child;
                tail++;
            }
        }
    }
    wave_sync();
    // phase 3 (all lanes): publish order, inverse map and the per-node record for the next read
    for (int32_t i = lane; i < node_count; i += kWave)
    {
        const uint32_t e   = (uint32_t)queue[i];
        const int32_t node = (int32_t)(e & kIdMask);
        const uint32_t oc = g.outgoing_edge_count[node], ic = g.incoming_edge_count[node];
        g.sorted_poa[i]        = (IdT)node;
        g.node_id_to_pos[node] = (IdT)i;
        g.local_cnt[node]      = (uint16_t)((e >> 28) | (oc << 4) | (ic << 10));
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// The incremental Kahn order for long-read graphs with its hot state in LDS (round 3). topsort_kahn_incr_hbm pays half a
// dozen dependent L2 / HBM round trips per replayed block and three or four per ordinary step: its node words (in-edge
// counters, change flags, previous position) and the previous order live in HBM. A multi-wave block owns its CU's LDS and
// the score ring is idle during the sort, so here
//   cnt8[n]   one byte per node: unvisited in-edges [0:6) | new in-edge or new node [6] | new out-edge or new node [7];
//             decremented with ds_sub_rtn_u32 on the dword around the byte (a counter is only decremented while >= 1, so
//             no borrow crosses into the neighbour bytes)
//   win[]     a sliding window over 1024 positions of the PREVIOUS order, slot = position & 1023: {out-edges 0..2,
//             node [0:20) | min(out-degree, 4) [20:23) | previous queue length [28:32)}, refilled 256 positions at a time by
//             all lanes (four 64-position chunks per round trip) while the sort is still more than 512 positions behind
//             its end -- positions only move forward (p = old nodes output so far)
//   qring[]   the live part of the queue (1024 entries; a longer queue gives up: the caller re-runs the HBM routine,
//             nothing has been published yet)
// "In sync at p" needs no previous position of the head node: the old nodes output so far are sigma[0..p) iff their count
// is p and the highest position output is p - 1, and the head is then compared with sigma[p] from the window like the rest
// of the queue. A replayed block and an ordinary step on the expected node (most of them: an old node popped where the
// previous run popped it, but with a changed neighbourhood) touch LDS only; HBM round trips remain for new nodes, old nodes
// popped out of turn, nodes with more than three out-edges (lane e fetches edge e: one trip) and the window refills.
// The popped entries go to the HBM queue (node | queue length << 28) as before and phase 3 publishes from there; the record
// carried from read to read (GraphView::local_cnt) is the same, so a window may switch between this routine and
// topsort_kahn_incr_hbm from read to read.
// ------------------------------------------------------------------------------------------------
constexpr int32_t kTwWin   = 1024;
constexpr int32_t kTwRing  = 1024;
constexpr int32_t kTwFixed = kTwWin * 16 + kTwRing * 4;
__device__ __forceinline__ int32_t topsort_incr_cnt8_lds_bytes(int32_t node_count) { return kTwFixed + ((node_count + 3) & ~3) + 64; }

template <typename IdT>
__device__ __forceinline__ bool topsort_kahn_incr_cnt8(const GraphView<IdT>& g, int32_t n_old, int32_t node_count, uint8_t* lds,
                                                       int32_t* queue, int lane)
{
    constexpr int32_t kQClip = 15;
    constexpr uint32_t kId   = 0xfffffu;
    uint4* win      = reinterpret_cast<uint4*>(lds);
    uint32_t* qring = reinterpret_cast<uint32_t*>(lds + kTwWin * 16);
    uint8_t* cnt8   = lds + kTwFixed;
    uint32_t* cntw  = reinterpret_cast<uint32_t*>(cnt8);
    const unsigned long long lanes_below = (1ull << lane) - 1;
    auto dec8 = [&](int32_t n) -> uint32_t { // old value of node n's byte
        const uint32_t sh  = ((uint32_t)n & 3u) * 8u;
        const uint32_t old = __hip_atomic_fetch_sub(cntw + (n >> 2), 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (old >> sh) & 0xffu;
    };
    // phase 1 (all lanes, four chunks per round trip): counters and change flags, sources in ascending node id
    int32_t tail = 0;
    for (int32_t base = 0; base < node_count; base += 4 * kWave)
    {
        uint32_t ic[4], oc[4], m[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int32_t nc = min(base + q * kWave + lane, node_count - 1);
            ic[q]            = g.incoming_edge_count[nc];
            oc[q]            = g.outgoing_edge_count[nc];
            m[q]             = g.local_cnt[nc >= n_old ? 0 : nc];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int32_t n     = base + q * kWave + lane;
            const bool in       = n < node_count;
            const bool is_new   = n >= n_old;
            const uint32_t din  = (is_new || ((m[q] >> 10) & 63u) != ic[q]) ? 1u : 0u;
            const uint32_t dout = (is_new || ((m[q] >> 4) & 63u) != oc[q]) ? 1u : 0u;
            if (in) cnt8[n] = (uint8_t)((ic[q] & 0x3fu) | (din << 6) | (dout << 7));
            const bool is_src           = in && ic[q] == 0;
            const unsigned long long ms = __ballot(is_src);
            const int32_t slot          = tail + __popcll(ms & lanes_below);
            if (is_src && slot < kTwRing) qring[slot] = (uint32_t)n;
            tail += __popcll(ms);
        }
    }
    if (tail > kTwRing) return false;
    auto fill = [&](int32_t pos0) { // window slots of positions [pos0, pos0 + 256)
        int32_t node[4];
        uint32_t oc[4], e0[4], e1[4], e2[4], m[4];
#pragma unroll
        for (int q = 0; q < 4; q++) node[q] = (int32_t)g.sorted_poa[min(pos0 + q * kWave + lane, n_old - 1)];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            oc[q] = g.outgoing_edge_count[node[q]];
            e0[q] = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node[q] * kEdges];
            e1[q] = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node[q] * kEdges + 1];
            e2[q] = (uint32_t)(int32_t)g.outgoing_edges[(int64_t)node[q] * kEdges + 2];
            m[q]  = g.local_cnt[node[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int32_t pos = pos0 + q * kWave + lane;
            if (pos < n_old)
                win[pos & (kTwWin - 1)] = make_uint4(e0[q], e1[q], e2[q], (uint32_t)node[q] | (min(oc[q], 4u) << 20) | ((m[q] & 15u) << 28));
        }
    };
    for (int32_t pos0 = 0; pos0 < min(n_old, kTwWin); pos0 += 4 * kWave) fill(pos0);
    wave_sync();
    // phase 2: wave-uniform control; k = new nodes output so far, M = highest previous position output so far
    int32_t head = 0, k = 0, M = -1, wbase = 0;
    while (head < tail)
    {
        const int32_t p = head - k; // old nodes output so far = the previous position the state could be in sync with
        while (p >= wbase + kTwWin / 2 && wbase + kTwWin < n_old)
        {
            fill(wbase + kTwWin);
            wbase += 4 * kWave;
        }
        const int32_t len = tail - head;
        // lane l looks at queue entry head + l and at position p + l of the previous order
        const uint32_t ql  = qring[(head + lane) & (kTwRing - 1)];
        const int32_t posl = p + lane;
        const bool valid   = posl < n_old;
        const uint4 e      = win[posl & (kTwWin - 1)];
        const int32_t node = valid ? (int32_t)(e.w & kId) : 0;
        const int32_t noc  = valid ? (int32_t)((e.w >> 20) & 7u) : 0;
        const int32_t ch0  = noc > 0 ? (int32_t)e.x : node;
        const int32_t ch1  = noc > 1 ? (int32_t)e.y : node;
        const int32_t ch2  = noc > 2 ? (int32_t)e.z : node;
        const uint32_t fn = cnt8[node], f0 = cnt8[ch0], f1 = cnt8[ch1], f2 = cnt8[ch2];
        const int32_t u   = wave_first((int32_t)ql);
        const bool is_new = u >= n_old;
        // a position cannot be replayed when its node has a new out-edge or more than three children, or a child has a new
        // in-edge; behind the head also when the node itself has a new in-edge (the head has really been pushed)
        const bool bad = !valid | ((fn & 0x80u) != 0) | (noc > 3) | ((lane > 0) & ((fn & 0x40u) != 0)) | ((noc > 0) & ((f0 & 0x40u) != 0)) |
                         ((noc > 1) & ((f1 & 0x40u) != 0)) | ((noc > 2) & ((f2 & 0x40u) != 0));
        const unsigned long long mb = __ballot(bad);
        const bool q_differs        = lane < len && !(valid && (int32_t)ql == node);
        const uint32_t w0           = (uint32_t)wave_first((int32_t)e.w);
        const bool expected         = !is_new && p < n_old && (int32_t)(w0 & kId) == u; // u is sigma[p]
        const int32_t qo            = (int32_t)(w0 >> 28);
        const bool block            = expected && !(mb & 1ull) && M == p - 1 && len == qo && qo < kQClip && __ballot(q_differs) == 0;
        if (block)
        {
            // lane l replays the pop of position p + l; all decrements in flight together; a child whose counter reaches 0 was
            // pushed here by the previous run, and the pushed entries continue the previous order
            const int32_t b = mb ? __ffsll((unsigned long long)mb) - 1 : kWave;
            const bool do0 = lane < b && noc > 0, do1 = lane < b && noc > 1, do2 = lane < b && noc > 2;
            uint32_t r0 = 0, r1 = 0, r2 = 0;
            if (do0) r0 = dec8(ch0);
            if (do1) r1 = dec8(ch1);
            if (do2) r2 = dec8(ch2);
            const int32_t npush = __popcll(__ballot(do0 && (r0 & 0x3fu) == 1u)) + __popcll(__ballot(do1 && (r1 & 0x3fu) == 1u)) +
                                  __popcll(__ballot(do2 && (r2 & 0x3fu) == 1u));
            for (int32_t j = lane; j < npush; j += kWave) qring[(tail + j) & (kTwRing - 1)] = win[(p + len + j) & (kTwWin - 1)].w & kId;
            if (lane < b) queue[head + lane] = (int32_t)(e.w & (kId | 0xf0000000u));
            head += b;
            tail += npush;
            M = p + b - 1;
            if (tail - head > kTwRing) return false;
            continue;
        }
        // ordinary Kahn step: lane e owns out-edge e
        int32_t oc, child, pos = p;
        if (expected && ((w0 >> 20) & 7u) <= 3u)
        {
            oc               = (int32_t)((w0 >> 20) & 7u);
            const int32_t c0 = wave_first((int32_t)e.x), c1 = wave_first((int32_t)e.y), c2 = wave_first((int32_t)e.z);
            child            = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
        }
        else
        {
            const int32_t pu = (int32_t)g.node_id_to_pos[is_new ? 0 : u];
            child            = (int32_t)g.outgoing_edges[(int64_t)u * kEdges + min(lane, kEdges - 1)];
            oc               = wave_first((int32_t)g.outgoing_edge_count[u]);
            if (!expected) pos = wave_first(pu);
        }
        const bool act     = lane < oc;
        uint32_t old       = 0;
        if (act) old = dec8(child);
        const bool hit              = act && (old & 0x3fu) == 1u;
        const unsigned long long mh = __ballot(hit);
        if (hit) qring[(tail + __popcll(mh & lanes_below)) & (kTwRing - 1)] = (uint32_t)child;
        if (lane == 0) queue[head] = (int32_t)((uint32_t)u | ((uint32_t)min(len, kQClip) << 28));
        head++;
        tail += __popcll(mh);
        k += is_new ? 1 : 0;
        M = is_new ? M : max(M, pos);
        if (tail - head > kTwRing) return false;
    }
    wave_sync();
    // phase 3 (all lanes, four chunks per round trip): publish order, inverse map and the per-node record for the next read
    for (int32_t base = 0; base < node_count; base += 4 * kWave)
    {
        uint32_t e[4], oc[4], ic[4];
#pragma unroll
        for (int q = 0; q < 4; q++) e[q] = (uint32_t)queue[min(base + q * kWave + lane, node_count - 1)];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int32_t node = (int32_t)(e[q] & kId);
            oc[q] = g.outgoing_edge_count[node];
            ic[q] = g.incoming_edge_count[node];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const int32_t i = base + q * kWave + lane;
            if (i < node_count)
            {
                const int32_t node     = (int32_t)(e[q] & kId);
                g.sorted_poa[i]        = (IdT)node;
                g.node_id_to_pos[node] = (IdT)i;
                g.local_cnt[node]      = (uint16_t)((e[q] >> 28) | (oc[q] << 4) | (ic[q] << 10));
            }
        }
    }
    wave_sync();
    return true;
}

// racon/spoa DFS order (aligned nodes adjacent)
template <typename IdT>
__device__ void topsort_racon(const GraphView<IdT>& g, int32_t node_count, int32_t max_nodes_per_graph)
{
    int32_t node_idx   = -1;
    int32_t sorted_idx = 0;
    for (int32_t i = 0; i < max_nodes_per_graph; i++)
    {
        g.marks[i] = 0;
        g.check[i] = 1;
    }
    for (int32_t i = 0; i < node_count; i++)
    {
        if (g.marks[i] != 0) continue;
        node_idx++;
        g.to_visit[node_idx] = (IdT)i;
        while (node_idx != -1)
        {
            int32_t node_id = g.to_visit[node_idx];
            bool valid      = true;
            if (g.marks[node_id] != 2)
            {
                for (int32_t e = 0; e < g.incoming_edge_count[node_id]; e++)
                {
                    int32_t begin = g.incoming_edges[(int64_t)node_id * kEdges + e];
                    if (g.marks[begin] != 2)
                    {
                        node_idx++;
                        g.to_visit[node_idx] = (IdT)begin;
                        valid                = false;
                    }
                }
                if (g.check[node_id])
                {
                    for (int32_t a = 0; a < g.node_alignment_count[node_id]; a++)
                    {
                        int32_t aid = g.node_alignments[(int64_t)node_id * kAligns + a];
                        if (g.marks[aid] != 2)
                        {
                            node_idx++;
                            g.to_visit[node_idx] = (IdT)aid;
                            g.check[aid]         = 0;
                            valid                = false;
                        }
                    }
                }
                if (valid)
                {
                    g.marks[node_id] = 2;
                    if (g.check[node_id])
                    {
                        g.sorted_poa[sorted_idx]    = (IdT)node_id;
                        g.node_id_to_pos[node_id]   = (IdT)sorted_idx;
                        sorted_idx++;
                        for (int32_t a = 0; a < g.node_alignment_count[node_id]; a++)
                        {
                            int32_t aid              = g.node_alignments[(int64_t)node_id * kAligns + a];
                            g.sorted_poa[sorted_idx] = (IdT)aid;
                            g.node_id_to_pos[aid]    = (IdT)sorted_idx;
                            sorted_idx++;
                        }
                    }
                }
                else
                    g.marks[node_id] = 1;
            }
            if (valid) node_idx--;
        }
    }
}

// The same order, for the MSA kernel, by the whole wavefront: the depth-first walk stays serial (the order is defined by
// it) but a visit is one memory round trip instead of eight dependent ones -- all lanes load the node's in-edges and aligned
// nodes at once (lane e: edge / alignment e; the two counts ride along), the per-node marks and "check" flags live in LDS
// (one byte per node: marks [0:2), check [2]), the nodes to push are found with a ballot and written in slot order by prefix
// popcount, and the stack is in LDS. Returns false when the stack outgrows its LDS (the caller then runs topsort_racon).
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ bool topsort_racon_wave(const GraphView<IdT>& g, int32_t node_count, uint8_t* state, uint32_t* stack,
                                                   int32_t stack_cap, int lane)
{
    static_assert(kEdges <= kWave && kAligns <= kWave, "one lane per edge / alignment slot");
    for (int32_t i = lane; i < node_count; i += kWave) state[i] = 4; // marks 0, check 1
    wave_sync();
    int32_t top = -1, sorted_idx = 0;
    for (int32_t i = 0; i < node_count; i++)
    {
        if ((wave_first((int32_t)state[i]) & 3) != 0) continue;
        top = 0;
        if (lane == 0) stack[0] = (uint32_t)i;
        while (top >= 0)
        {
            // a stack entry: node [0:24) | its alignment count [24:30) | "expanded" [31]: this very entry has pushed its
            // unfinished predecessors and aligned nodes once -- they sat directly above it and are all finished by the
            // time it is on top again (LIFO), so the second visit needs no look at its edges (a second entry of the same
            // node, pushed by somebody else in between, carries no such bit and takes the full look)
            const uint32_t entry = (uint32_t)wave_first((int32_t)stack[top]);
            const int32_t node   = (int32_t)(entry & 0xffffffu);
            const uint32_t st    = (uint32_t)wave_first((int32_t)state[node]);
            bool valid           = true;
            if ((st & 3) != 2)
            {
                const bool check = (st & 4) != 0;
                int32_t ac = 0, al = 0;
                if (entry >> 31)
                {
                    ac = (int32_t)((entry >> 24) & 63u);
                    if (check && ac > 0) al = lane < kAligns ? (int32_t)g.node_alignments[(int64_t)node * kAligns + lane] : 0;
                }
                else
                {
                    // one round trip: both counts, every in-edge slot, every alignment slot
                    const int32_t ic  = wave_first((int32_t)g.incoming_edge_count[node]);
                    ac                = wave_first((int32_t)g.node_alignment_count[node]);
                    const int32_t ed  = lane < kEdges ? (int32_t)g.incoming_edges[(int64_t)node * kEdges + lane] : 0;
                    al                = lane < kAligns ? (int32_t)g.node_alignments[(int64_t)node * kAligns + lane] : 0;
                    const bool push_e = lane < ic && (state[max(ed, 0)] & 3) != 2;
                    const bool push_a = check && lane < ac && (state[max(al, 0)] & 3) != 2;
                    const unsigned long long me = __ballot(push_e), ma = __ballot(push_a);
                    const int32_t ne = __popcll(me), na = __popcll(ma);
                    if (top + ne + na >= stack_cap) return false;
                    const unsigned long long below = (1ull << lane) - 1;
                    if (push_e) stack[top + 1 + __popcll(me & below)] = (uint32_t)ed;
                    if (push_a)
                    {
                        stack[top + 1 + ne + __popcll(ma & below)] = (uint32_t)al;
                        state[al] &= (uint8_t)~4u; // check[aid] = 0 (two slots never name the same node)
                    }
                    valid = ne + na == 0;
                    if (!valid && lane == 0)
                    {
                        stack[top]  = (uint32_t)node | ((uint32_t)min(ac, 63) << 24) | 0x80000000u;
                        state[node] = (uint8_t)((st & ~3u) | 1u);
                    }
                    top += ne + na;
                }
                if (valid)
                {
                    if (lane == 0) state[node] = (uint8_t)((st & ~3u) | 2u);
                    if (check)
                    {
                        if (lane == 0)
                        {
                            g.sorted_poa[sorted_idx]  = (IdT)node;
                            g.node_id_to_pos[node]    = (IdT)sorted_idx;
                        }
                        if (lane < ac)
                        {
                            g.sorted_poa[sorted_idx + 1 + lane] = (IdT)al;
                            g.node_id_to_pos[al]                = (IdT)(sorted_idx + 1 + lane);
                        }
                        sorted_idx += 1 + ac;
                    }
                }
                // (LDS only: one wavefront's LDS operations execute in order; the order's global stores are not read before
                // the end and must not be waited for in every visit)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (valid) top--;
        }
    }
    wave_sync();
    return true;
}

// MSA column of every node from the racon order and the check flags topsort_racon_wave left in `state`: a node with its
// flag set opens a column, the aligned nodes behind it share it (node_id_to_msa_pos below, all lanes). Returns the MSA length.
template <typename IdT>
__device__ __forceinline__ int32_t node_id_to_msa_pos_wave(const GraphView<IdT>& g, int32_t node_count, const uint8_t* state, int lane)
{
    int32_t columns = 0;
    for (int32_t base = 0; base < node_count; base += kWave)
    {
        const int32_t rank  = base + lane;
        const int32_t node  = rank < node_count ? (int32_t)g.sorted_poa[rank] : 0;
        const bool opens    = rank < node_count && (state[node] & 4) != 0;
        const unsigned long long m = __ballot(opens);
        if (rank < node_count) g.msa_pos[node] = (IdT)(columns + __popcll(m & ((2ull << lane) - 1)) - 1);
        columns += __popcll(m);
    }
    wave_sync();
    return columns;
}

// heaviest bundle + branch completion. `scores` has a guard element at index -1.
template <typename IdT>
__device__ int32_t branch_completion(int32_t max_score_id_pos, const GraphView<IdT>& g, int32_t node_count,
                                     int32_t* scores, IdT* predecessors)
{
    int32_t node_id    = g.sorted_poa[max_score_id_pos];
    uint16_t out_edges = g.outgoing_edge_count[node_id];
    for (int32_t oe = 0; oe < out_edges; oe++)
    {
        int32_t out_node_id = g.outgoing_edges[(int64_t)node_id * kEdges + oe];
        uint16_t in_edges   = g.incoming_edge_count[out_node_id];
        for (int32_t ie = 0; ie < in_edges; ie++)
        {
            int32_t id = g.incoming_edges[(int64_t)out_node_id * kEdges + ie];
            if (id != node_id) scores[id] = -1;
        }
    }
    int32_t max_score = 0, max_score_id = 0;
    for (int32_t graph_pos = max_score_id_pos + 1; graph_pos < node_count; graph_pos++)
    {
        node_id               = g.sorted_poa[graph_pos];
        int32_t pred          = -1;
        int32_t score_node_id = -1;
        uint16_t in_edges     = g.incoming_edge_count[node_id];
        for (int32_t e = 0; e < in_edges; e++)
        {
            int32_t begin = g.incoming_edges[(int64_t)node_id * kEdges + e];
            if (scores[begin] == -1) continue;
            int32_t edge_w = (int32_t)g.incoming_edge_w[(int64_t)node_id * kEdges + e];
            if (score_node_id < edge_w || (score_node_id == edge_w && scores[pred] <= scores[begin]))
            {
                score_node_id = edge_w;
                pred          = begin;
            }
        }
        predecessors[node_id] = (IdT)pred;
        if (pred != -1) score_node_id += scores[pred];
        if (max_score <= score_node_id)
        {
            max_score    = score_node_id;
            max_score_id = node_id;
        }
        scores[node_id] = score_node_id;
    }
    return max_score_id;
}

template <typename IdT>
__device__ void generate_consensus(const GraphView<IdT>& g, int32_t node_count, IdT* predecessors,
                                   int32_t* scores_base, uint8_t* consensus, uint16_t* coverage,
                                   int32_t max_limit_consensus_size)
{
    int32_t* scores = scores_base + 1;
    scores[-1]      = -1;
    int32_t max_score_id = 0, max_score = -1;
    for (int32_t graph_pos = 0; graph_pos < node_count; graph_pos++)
    {
        int32_t node_id       = g.sorted_poa[graph_pos];
        uint16_t in_edges     = g.incoming_edge_count[node_id];
        int32_t score_node_id = -1;
        int32_t pred          = -1;
        for (int32_t e = 0; e < in_edges; e++)
        {
            int32_t edge_w = (int32_t)g.incoming_edge_w[(int64_t)node_id * kEdges + e];
            int32_t begin  = g.incoming_edges[(int64_t)node_id * kEdges + e];
            if (score_node_id < edge_w || (score_node_id == edge_w && scores[pred] <= scores[begin]))
            {
                score_node_id = edge_w;
                pred          = begin;
            }
        }
        predecessors[node_id] = (IdT)pred;
        if (pred != -1) score_node_id += scores[pred];
        if (max_score <= score_node_id)
        {
            max_score_id = node_id;
            max_score    = score_node_id;
        }
        scores[node_id] = score_node_id;
    }
    int32_t loop_count = 0;
    if (g.outgoing_edge_count[max_score_id] != 0)
    {
        while (g.outgoing_edge_count[max_score_id] != 0 && loop_count < node_count)
        {
            max_score_id = branch_completion<IdT>(g.node_id_to_pos[max_score_id], g, node_count, scores, predecessors);
            loop_count++;
        }
    }
    if (loop_count >= node_count)
    {
        consensus[0] = kKernelError;
        consensus[1] = kLoopCountExceeded;
        return;
    }
    int32_t consensus_pos = 0, consensus_count = 0;
    auto cov_of = [&](int32_t id) -> uint16_t {
        uint16_t cov = g.coverage[id];
        for (int32_t a = 0; a < g.node_alignment_count[id]; a++)
            cov = (uint16_t)(cov + g.coverage[g.node_alignments[(int64_t)id * kAligns + a]]);
        return cov;
    };
    while (predecessors[max_score_id] != -1)
    {
        consensus[consensus_pos] = g.nodes[max_score_id];
        coverage[consensus_pos]  = cov_of(max_score_id);
        max_score_id             = predecessors[max_score_id];
        consensus_pos            = min(consensus_pos + 1, max_limit_consensus_size - 1);
        consensus_count++;
    }
    consensus[consensus_pos] = g.nodes[max_score_id];
    coverage[consensus_pos]  = cov_of(max_score_id);
    if (consensus_count >= (max_limit_consensus_size - 1))
    {
        consensus[0] = kKernelError;
        consensus[1] = kExceededMaximumSequenceSize;
        return;
    }
    consensus_pos++;
    consensus[consensus_pos] = '\0';
}

// ------------------------------------------------------------------------------------------------
// Heaviest bundle + branch completion + consensus read-out with the working set in LDS (16-bit ids, <= 3072 nodes).
// Same decisions as generate_consensus / branch_completion above (cudapoa_generate_consensus.cuh:35-283); the
// serial passes run wave-uniformly on the scalar unit against LDS copies staged by all lanes, so a node costs LDS
// round trips instead of a chain of HBM round trips, and the consensus / coverage read-out is a parallel map.
//   rec[n]    3 x u32 : in-edge 0..2 (12 bit each) + in-degree | in-edge 2, weight 0 | weight 1, weight 2
//   scores[n] i32 with a guard element at index -1; pred[n] i16; the path list aliases rec after the passes
// ------------------------------------------------------------------------------------------------
// The tables are laid out for `cap` nodes, chosen at launch: 18 B per node, so the full 3072-node graph needs 55 KB
// (two blocks per CU), while 2176 nodes fit the 39 KB that let four blocks share a CU -- which is what a batch of
// more than 512 windows needs to run in one round. A window whose graph is larger than `cap` takes the HBM routine.
constexpr int kConsLdsNodes      = 3072;
constexpr int kConsLdsNodesSmall = 2176;
__host__ __device__ constexpr int cons_lds_bytes(int cap) { return cap * 12 + (cap + 4) * 4 + cap * 2; }
constexpr int kConsLdsBytes = cons_lds_bytes(kConsLdsNodes);

template <typename IdT>
__device__ __forceinline__ void generate_consensus_lds(const GraphView<IdT>& g, int32_t node_count, uint8_t* lds, int32_t cap,
                                                       uint8_t* consensus, uint16_t* coverage,
                                                       int32_t max_limit_consensus_size, int lane, bool node_by_node = false)
{
    uint32_t* rec    = reinterpret_cast<uint32_t*>(lds);
    int32_t* scores  = reinterpret_cast<int32_t*>(lds + cap * 12) + 1; // index -1 is the guard
    int16_t* pred    = reinterpret_cast<int16_t*>(lds + cap * 12 + (cap + 4) * 4);
    uint16_t* path   = reinterpret_cast<uint16_t*>(lds); // aliases rec once the passes are done

    // four 64-node chunks share one HBM round trip (seven independent loads per node)
    constexpr int kU = 4;
    for (int32_t base = 0; base < node_count; base += kU * kWave)
    {
        uint32_t cnt[kU], e0[kU], e1[kU], e2[kU], w0[kU], w1[kU], w2[kU];
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            const int32_t n = min(base + u * kWave + lane, node_count - 1);
            cnt[u] = g.incoming_edge_count[n];
            e0[u]  = (uint16_t)g.incoming_edges[(int64_t)n * kEdges + 0];
            e1[u]  = (uint16_t)g.incoming_edges[(int64_t)n * kEdges + 1];
            e2[u]  = (uint16_t)g.incoming_edges[(int64_t)n * kEdges + 2];
            w0[u]  = g.incoming_edge_w[(int64_t)n * kEdges + 0];
            w1[u]  = g.incoming_edge_w[(int64_t)n * kEdges + 1];
            w2[u]  = g.incoming_edge_w[(int64_t)n * kEdges + 2];
        }
#pragma unroll
        for (int u = 0; u < kU; u++)
        {
            const int32_t n = base + u * kWave + lane;
            if (n >= node_count) continue;
            // slots past the in-degree hold stale values: point them at node 0 so their score reads stay in range
            rec[3 * n + 0] = (cnt[u] > 0 ? e0[u] & 0xfff : 0u) | ((cnt[u] > 1 ? e1[u] & 0xfff : 0u) << 12) | (min(cnt[u], 255u) << 24);
            rec[3 * n + 1] = (cnt[u] > 2 ? e2[u] & 0xfff : 0u) | (w0[u] << 16);
            rec[3 * n + 2] = w1[u] | (w2[u] << 16);
        }
    }
    if (lane == 0) scores[-1] = -1;
    wave_sync();

    // one node of the heaviest-bundle recurrence, wave-uniform against LDS (cudapoa_generate_consensus.cuh:120-170);
    // skip_cut: edges from nodes whose score was cut to -1 are ignored (branch completion)
    auto node_step = [&](int32_t node, bool skip_cut, int32_t& best_out, int32_t& score_out) {
        const uint32_t r0 = (uint32_t)wave_first((int32_t)rec[3 * node + 0]);
        const uint32_t r1 = (uint32_t)wave_first((int32_t)rec[3 * node + 1]);
        const uint32_t r2 = (uint32_t)wave_first((int32_t)rec[3 * node + 2]);
        const int32_t cnt = (int32_t)(r0 >> 24);
        const int32_t b0 = (int32_t)(r0 & 0xfff), b1 = (int32_t)((r0 >> 12) & 0xfff), b2 = (int32_t)(r1 & 0xfff);
        const int32_t w0 = (int32_t)(r1 >> 16), w1 = (int32_t)(r2 & 0xffff), w2 = (int32_t)(r2 >> 16);
        const int32_t s0 = wave_first(scores[b0]), s1 = wave_first(scores[b1]), s2 = wave_first(scores[b2]);
        int32_t best_w = -1, best = -1, best_score = -1; // scores[-1] == -1
        auto consider = [&](bool present, int32_t begin, int32_t w, int32_t sc) {
            const bool take = present && !(skip_cut && sc == -1) && (best_w < w || (best_w == w && best_score <= sc));
            best_w     = take ? w : best_w;
            best       = take ? begin : best;
            best_score = take ? sc : best_score;
        };
        consider(cnt > 0, b0, w0, s0);
        consider(cnt > 1, b1, w1, s1);
        consider(cnt > 2, b2, w2, s2);
        for (int32_t e = 3; e < cnt; e++) // rare: more than three in-edges, from the HBM lists
        {
            const int32_t begin = wave_first((int32_t)g.incoming_edges[(int64_t)node * kEdges + e]);
            const int32_t w     = wave_first((int32_t)g.incoming_edge_w[(int64_t)node * kEdges + e]);
            consider(true, begin, w, wave_first(scores[begin]));
        }
        best_out  = best;
        score_out = best != -1 ? best_w + best_score : best_w;
    };
    // one pass over sorted positions [first_pos, node_count), node by node
    auto bundle_pass = [&](int32_t first_pos, bool skip_cut, int32_t max_score) -> int32_t {
        int32_t max_score_id = 0;
        int32_t chunk_base   = first_pos;
        int32_t chunk        = (chunk_base + lane < node_count) ? (int32_t)g.sorted_poa[chunk_base + lane] : 0;
        for (int32_t pos = first_pos; pos < node_count; pos++)
        {
            if (pos - chunk_base >= kWave)
            {
                chunk_base = pos;
                chunk      = (chunk_base + lane < node_count) ? (int32_t)g.sorted_poa[chunk_base + lane] : 0;
            }
            const int32_t node = __builtin_amdgcn_readlane(chunk, pos - chunk_base);
            int32_t best, score;
            node_step(node, skip_cut, best, score);
            lane0_store_u16(pred + node, (uint32_t)best);
            lane0_store_u32(scores + node, (uint32_t)score);
            const bool better = max_score <= score;
            max_score         = better ? score : max_score;
            max_score_id      = better ? node : max_score_id;
        }
        return max_score_id;
    };
    // The first pass (all positions, nothing cut), 64 positions at a time (round 4). Which in-edge a node takes is decided by the
    // edge weights; the predecessors' scores only break ties. So every lane decides for its own node, and what is left of the
    // serial recurrence is score = weight + score of the chosen predecessor for the nodes whose chosen predecessor sits in the
    // same 64 positions: sums along chains, taken by pointer jumping (at most six rounds of two ds_bpermute). A predecessor
    // inside the chunk is recognised by a marker (-2 - lane) that its lane leaves in the score table before the others look
    // their predecessors up. A node with a tie that involves such a predecessor, or with more than three in-edges, takes
    // node_step once the scores of the lanes before it are in LDS. Same decisions, same order of the running maximum (the
    // later position wins a tie).
    auto bundle_pass_chunked = [&]() -> int32_t {
        int32_t lane_max = -2, lane_arg = 0; // every score is >= -1
        int32_t next = lane < node_count ? (int32_t)g.sorted_poa[lane] : 0;
        for (int32_t base = 0; base < node_count; base += kWave)
        {
            const int32_t node = next;
            const bool valid   = base + lane < node_count;
            if (base + kWave < node_count) next = (base + kWave + lane < node_count) ? (int32_t)g.sorted_poa[base + kWave + lane] : 0;
            if (valid) scores[node] = -2 - lane;
            asm volatile("" ::: "memory"); // one wavefront's LDS operations execute in order
            const uint32_t r0 = rec[3 * node + 0], r1 = rec[3 * node + 1], r2 = rec[3 * node + 2];
            const int32_t cnt = (int32_t)(r0 >> 24);
            const int32_t b0 = (int32_t)(r0 & 0xfff), b1 = (int32_t)((r0 >> 12) & 0xfff), b2 = (int32_t)(r1 & 0xfff);
            const int32_t w0 = (int32_t)(r1 >> 16), w1 = (int32_t)(r2 & 0xffff), w2 = (int32_t)(r2 >> 16);
            const int32_t s0 = scores[b0], s1 = scores[b1], s2 = scores[b2];
            int32_t best_w = -1, best = -1, best_score = -1;
            bool slow = cnt > 3;
            auto consider = [&](bool present, int32_t begin, int32_t w, int32_t sc) {
                slow |= present && best_w == w && (sc <= -2 || best_score <= -2);
                const bool take = present && (best_w < w || (best_w == w && best_score <= sc));
                best_w     = take ? w : best_w;
                best       = take ? begin : best;
                best_score = take ? sc : best_score;
            };
            consider(cnt > 0, b0, w0, s0);
            consider(cnt > 1, b1, w1, s1);
            consider(cnt > 2, b2, w2, s2);
            const bool inside = best_score <= -2; // the chosen predecessor is lane (-2 - best_score) of this chunk
            slow &= valid;
            // score = weight + score of the chosen predecessor along the chains inside the chunk: pointer jumping. A lane that
            // waits holds the sum of the weights from its node down to (not including) the node `ref` points at; a lane that
            // needs node_step stops the chains that run through it (it adds nothing and points at itself) until it is resolved.
            int32_t val  = slow ? 0 : (inside ? best_w : (best != -1 ? best_w + best_score : best_w));
            int32_t ref  = (inside && !slow) ? -2 - best_score : lane;
            bool waiting = valid && (inside || slow);
            unsigned long long stoppers = __ballot(slow);
            for (;;)
            {
                for (;;)
                {
                    const bool hop = waiting && !((stoppers >> lane) & 1) && !((stoppers >> ref) & 1);
                    if (__ballot(hop) == 0) break;
                    const int32_t vr = __shfl(val, ref, kWave);
                    const int32_t pr = __shfl(ref | (waiting ? 64 : 0), ref, kWave);
                    if (hop)
                    {
                        val += vr;
                        waiting = (pr & 64) != 0;
                        ref     = waiting ? (pr & 63) : ref;
                    }
                }
                if (stoppers == 0) break;
                // the first node that needs node_step: every lane before it is final
                const int32_t j = __ffsll(stoppers) - 1;
                stoppers &= stoppers - 1;
                if (valid && lane < j) scores[node] = val; // what the node's in-edges may look up
                asm volatile("" ::: "memory");
                int32_t bj, sj;
                node_step(__builtin_amdgcn_readlane(node, j), false, bj, sj);
                best = lane == j ? bj : best;
                if (lane == j)
                {
                    val     = sj;
                    waiting = false;
                }
                else if (waiting && ref == j && !((stoppers >> lane) & 1))
                {
                    val += sj;
                    waiting = false;
                }
            }
            if (valid)
            {
                scores[node] = val;
                pred[node]   = (int16_t)best;
            }
            asm volatile("" ::: "memory");
            // running maximum per lane (a later position replaces an equal score); reduced over the lanes behind the loop
            if (valid && lane_max <= val)
            {
                lane_max = val;
                lane_arg = ((base + lane) << 12) | node;
            }
        }
        // the largest score; among equal ones the last position (node ids and positions are 12-bit)
        int32_t m = lane_max;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off, kWave));
        int32_t arg = lane_max == m ? lane_arg : -1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) arg = max(arg, __shfl_xor(arg, off, kWave));
        const int32_t max_score_id = wave_first(arg) & 0xfff;
        return max_score_id;
    };

    int32_t max_score_id = node_by_node ? bundle_pass(0, false, -1) : bundle_pass_chunked();
    int32_t loop_count   = 0;
    while (wave_first((int32_t)g.outgoing_edge_count[max_score_id]) != 0 && loop_count < node_count)
    {
        // branch completion (:35-97): cut every other way into the successors of the bundle's end, redo the tail
        const int32_t node_id = max_score_id;
        const int32_t oc      = wave_first((int32_t)g.outgoing_edge_count[node_id]);
        for (int32_t oe = 0; oe < oc; oe++)
        {
            const int32_t out_node = wave_first((int32_t)g.outgoing_edges[(int64_t)node_id * kEdges + oe]);
            const int32_t ic       = wave_first((int32_t)g.incoming_edge_count[out_node]);
            for (int32_t ie = 0; ie < ic; ie++)
            {
                const int32_t id = wave_first((int32_t)g.incoming_edges[(int64_t)out_node * kEdges + ie]);
                if (id != node_id) lane0_store_u32(scores + id, (uint32_t)-1);
            }
        }
        max_score_id = bundle_pass(wave_first((int32_t)g.node_id_to_pos[node_id]) + 1, true, 0);
        loop_count++;
    }
    wave_sync();
    if (loop_count >= node_count)
    {
        if (lane == 0) { consensus[0] = kKernelError; consensus[1] = kLoopCountExceeded; }
        return;
    }
    // walk the bundle back to its start (serial, LDS), then fill bases and coverage in parallel
    int32_t count = 0;
    {
        int32_t id = max_score_id;
        for (;;)
        {
            lane0_store_u16(path + min(count, 2 * cap), (uint32_t)id);
            const int32_t p = wave_first((int32_t)pred[id]);
            if (p == -1) break;
            id = p;
            count++;
        }
    }
    wave_sync();
    if (count >= max_limit_consensus_size - 1)
    {
        if (lane == 0) { consensus[0] = kKernelError; consensus[1] = kExceededMaximumSequenceSize; }
        return;
    }
    for (int32_t k = lane; k <= count; k += kWave)
    {
        const int32_t id = path[k];
        uint16_t cov     = g.coverage[id];
        const int32_t na = g.node_alignment_count[id];
        for (int32_t a = 0; a < na; a++) cov = (uint16_t)(cov + g.coverage[g.node_alignments[(int64_t)id * kAligns + a]]);
        consensus[k] = g.nodes[id];
        coverage[k]  = cov;
    }
    if (lane == 0) consensus[count + 1] = '\0';
}

// MSA: column index per node (aligned nodes share a column), then one lane per sequence.
template <typename IdT>
__device__ int32_t node_id_to_msa_pos(const GraphView<IdT>& g, int32_t node_count)
{
    int32_t msa_pos = 0;
    for (int32_t rank = 0; rank < node_count; rank++)
    {
        int32_t node_id    = g.sorted_poa[rank];
        g.msa_pos[node_id] = (IdT)msa_pos;
        uint16_t ac        = g.node_alignment_count[node_id];
        for (int32_t n = 0; n < ac; n++) g.msa_pos[g.sorted_poa[++rank]] = (IdT)msa_pos;
        msa_pos++;
    }
    return msa_pos;
}

// One lane per sequence: the walk along the sequence's path is a chain of dependent loads, so what counts is round trips per
// node. The node's column, base, out-degree and its first two out-edges with their coverage counts are one round trip; an
// edge's coverage list (the sequences that use the edge, up to max_sequences_per_poa) is searched eight entries per round
// trip -- entry by entry the last sequences of a 32-read window paid up to 31 dependent loads per backbone node.
template <typename IdT>
__device__ void generate_msa_row(const GraphView<IdT>& g, uint16_t s, uint8_t* msa, int32_t msa_length,
                                 uint32_t max_sequences_per_poa, uint32_t max_limit_consensus_size)
{
    int32_t node_id      = g.seq_begin[s];
    int32_t filled_until = 0;
    uint8_t* row         = msa + (size_t)s * max_limit_consensus_size;
    for (;;)
    {
        const int64_t eb      = (int64_t)node_id * kEdges;
        const int32_t msa_pos = g.msa_pos[node_id];
        const uint8_t base    = g.nodes[node_id];
        const int32_t oc      = g.outgoing_edge_count[node_id];
        const int32_t e0 = (int32_t)g.outgoing_edges[eb], e1 = (int32_t)g.outgoing_edges[eb + 1];
        const int32_t c0 = (int32_t)g.out_cov_cnt[eb], c1 = (int32_t)g.out_cov_cnt[eb + 1];
        row[msa_pos] = base;
        for (int32_t i = filled_until; i < msa_pos; i++) row[i] = '-';
        filled_until  = msa_pos + 1;
        bool end_node = true;
        for (int32_t n = 0; n < oc && end_node; n++)
        {
            const int32_t to_node = n == 0 ? e0 : (n == 1 ? e1 : (int32_t)g.outgoing_edges[eb + n]);
            const int32_t cc      = n == 0 ? c0 : (n == 1 ? c1 : (int32_t)g.out_cov_cnt[eb + n]);
            const uint16_t* list  = g.out_cov + (eb + n) * max_sequences_per_poa;
            for (int32_t m0 = 0; m0 < cc && end_node; m0 += 8)
            {
                uint16_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = list[min(m0 + u, cc - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (m0 + u < cc && v[u] == s) end_node = false;
            }
            if (!end_node) node_id = to_node;
        }
        if (end_node)
        {
            for (int32_t i = filled_until; i < msa_length; i++) row[i] = '-';
            break;
        }
    }
    row[msa_length] = '\0';
}

// ------------------------------------------------------------------------------------------------
// All MSA rows of a window by the whole wavefront (round 3). generate_msa_row above walks one sequence per lane from its
// first node along the out-edges whose coverage list names it: a chain of two or three dependent HBM round trips per
// node, 30 000 nodes long on long reads, on at most 32 lanes. The same rows follow from the edges alone: sequence s
// visits exactly the end points of the edges whose list contains s (a read leaves every node it visits through one
// edge, and every such edge lies on its path), plus its first node when it has no edge at all. So: fill the rows with
// '-', then one lane per NODE scatters the node's base into the rows of the sequences on its out-edges -- and the
// edge's head into the same rows, which covers each sequence's last node. Loads of different nodes are independent,
// equal bytes may be written twice. Result identical to the walk (GWHIP_MSA_SERIAL=1 selects the walk: A/B in
// tests/test_gpu_poa.py).
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ void generate_msa_rows_wave(const GraphView<IdT>& g, int32_t node_count, int32_t num_seqs, uint8_t* msa,
                                                       int32_t msa_length, uint32_t max_sequences_per_poa,
                                                       uint32_t max_limit_consensus_size, int lane)
{
    for (int32_t s = 0; s < num_seqs; ++s)
    {
        uint8_t* row = msa + (size_t)s * max_limit_consensus_size;
        for (int32_t i = lane; i < msa_length; i += kWave) row[i] = '-';
    }
    wave_sync();
    for (int32_t n = lane; n < node_count; n += kWave)
    {
        const int64_t eb   = (int64_t)n * kEdges;
        const uint8_t base = g.nodes[n];
        const int32_t pos  = (int32_t)g.msa_pos[n];
        const int32_t oc   = g.outgoing_edge_count[n];
        for (int32_t e = 0; e < oc; ++e)
        {
            const int32_t to     = (int32_t)g.outgoing_edges[eb + e];
            const int32_t cc     = (int32_t)g.out_cov_cnt[eb + e];
            const uint8_t tbase  = g.nodes[to];
            const int32_t tpos   = (int32_t)g.msa_pos[to];
            const uint16_t* list = g.out_cov + (eb + e) * max_sequences_per_poa;
            for (int32_t m = 0; m < cc; ++m)
            {
                uint8_t* row = msa + (size_t)list[m] * max_limit_consensus_size;
                row[pos]     = base;
                row[tpos]    = tbase;
            }
        }
    }
    for (int32_t s = lane; s < num_seqs; s += kWave)
    {
        uint8_t* row     = msa + (size_t)s * max_limit_consensus_size;
        const int32_t n0 = (int32_t)g.seq_begin[s];
        row[(int32_t)g.msa_pos[n0]] = g.nodes[n0];
        row[msa_length]             = '\0';
    }
}

} // namespace gwhip
