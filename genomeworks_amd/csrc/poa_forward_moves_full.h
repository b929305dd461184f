// poa_forward_moves_full.h -- FULL-band NW with int16 scores as a packed pass (round 5). What it computes is
// cudapoa_nw.cuh:149-454 (restated in oracle/poa_oracle.c, nw_full of poa_full_device.h is the generic routine): the mode
// both reference benchmarks run (BatchConfig(1024, 200), cudapoa/benchmarks/single_batch.hpp:52, multi_batch.hpp:49).
//
// A full-band row is the banded row of poa_forward_moves_wide.h with everything band-related taken out: the band never
// moves (no row kind 1, per-lane addresses and read characters are constants of the read), every predecessor row covers
// every column (no sentinel cells, no "outside the band" tests, no second phase), column 0 is a real cell -- the vertical
// boundary H[r][0] = gap + max over the predecessors' H[p][0] (:186-216), a source's predecessor being row 0 -- and it IS
// the carry-in of the horizontal recurrence. What is left is NP = ceil(read length / 256) register passes of 256 columns
// per row through ONE instruction stream (lane l of pass p owns columns 256 p + 4 l + 1..4):
//   * the NP cross-lane scans of a row are independent until their carries are chained (pass p + 1 continues behind the
//     maximum of pass p), so a lone wavefront overlaps their DPP latencies instead of waiting for each;
//   * row kinds, register descriptors and the 4-row LDS ring of 1024 absolute column slots are those of the two-pass kernel
//     (classify_kinds<4>): 0 previous row from registers, 2 / 3 one / two to six predecessors at most 4 rows up from the
//     ring, 4 everything else -- here ALSO packed: the predecessor rows come from the HBM matrix in the ring's own shape
//     (one dword + one quad per lane and pass), any number of them;
//   * column 0 of the four ring rows lives in lanes 0..3 of one register (v_readlane by ring slot, a select to update): the
//     slot of column 0 is the slot of column 1024 (2046 mod 2048), which a fourth pass overwrites with a cell nobody reads;
//   * move bytes as in poa_forward_moves.h (rows up << 1 | columns left), EVERY cell of rows of kinds 0 / 2 / 3 decided
//     including column 1 (its horizontal and diagonal operands are real cells here), general rows leave 0; the walk is
//     traceback_moves (poa_traceback_moves.h) over a band that is the whole row.
// Preconditions (nw_full_packed): 1 <= read length <= 1023, gap < 0, and bounds on the scores that keep every packed
// operation inside int16 -- else the generic nw_full runs.
#pragma once

namespace gwhip
{

template <typename IdT, int NP>
__device__ __forceinline__ void full_forward_moves(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count, const uint8_t* read,
                                                   int32_t read_length, int16_t* scores, int32_t stride, uint8_t* moves, uint8_t* ring,
                                                   const uint64_t* xpred, int32_t gap_score, int32_t mismatch_score, int32_t match_score,
                                                   int32_t dbg)
{
    static_assert(NP >= 1 && NP <= 4, "256-column passes of a row of at most 1023 columns");
    static_assert(kWdSlotBytes == 2048 && kWdSlots == 4, "ring geometry of the two-pass kernel");
    constexpr uint32_t kMask = kWdSlotBytes - 1;
    const int lane      = threadIdx.x & (kWave - 1);
    const int32_t lane4 = lane * 4, lane8 = lane * 8;

    classify_kinds<kWdMaxDist>(rowinfo, graph_count, lane, xpred, dbg); // band starts are 0 in every row: kinds 0, 2, 3, 4
    wave_sync();
    // a score row goes to HBM only when somebody will read it back (mark_score_rows, poa_forward_moves.h): 98 % of the rows of
    // the metric windows keep theirs in move bytes, ring and registers. Column 0 is stored for every row (the walk along column
    // 0). GWHIP_DEBUG bit 25 (debug instantiation): every row stores its scores (A/B).
    mark_score_rows<IdT>(g, rowinfo, graph_count, lane, (dbg & (1 << 25)) != 0);

    const uint32_t GAP2   = pin_vgpr(pk_dup(gap_score));
    const uint32_t MAT2   = pin_vgpr(pk_dup(match_score));
    const uint32_t DIF2   = pin_vgpr(pk_dup(mismatch_score - match_score));
    const uint32_t ONE2   = pin_vgpr(0x00010001u);
    const uint32_t THREE2 = pin_vgpr(0x00030003u);
    const uint32_t NEG1   = pin_vgpr(0xffffffffu);
    const uint32_t ring_base = lds_addr(ring);
    const bool is_lane0      = lane == 0;
    // the last pass straddles the end of the read: lanes whose quad starts past it store nothing to HBM (the row ends there)
    const bool act            = (256 * (NP - 1) + lane4) < read_length;
    const uint32_t ld_last    = act ? (uint32_t)(512 * (NP - 1) + lane8) : 0u; // byte offset of the lane's last-pass cells in an HBM row

    // per pass: t * gap of the lane's cells (t = column - 1), read characters of its columns, ring byte offsets of its quad
    // (a1) and of the dword whose high half is the cell left of it (a0), the previous row's cells
    uint32_t K01[NP], K23[NP], rd[NP], a1[NP], a0[NP], P01[NP], P23[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        const int32_t t = 256 * p + lane4;
        K01[p] = pk_make((t + 0) * gap_score, (t + 1) * gap_score);
        K23[p] = pk_make((t + 2) * gap_score, (t + 3) * gap_score);
        rd[p]  = *reinterpret_cast<const uint32_t*>(read + t); // (the input buffer keeps zero slack behind every read)
        a1[p]  = (uint32_t)(512 * p + lane8);
        a0[p]  = (a1[p] - 4u) & kMask;
        P01[p] = pk_make((t + 1) * gap_score, (t + 2) * gap_score); // row 0: H[0][x] = x * gap
        P23[p] = pk_make((t + 3) * gap_score, (t + 4) * gap_score);
    }
    int32_t prev_c0 = 0;   // H[r - 1][0]
    int32_t c0ring  = 0;   // lanes 0..3: H[row][0] of the row in ring slot 0..3 (row 0 in slot 0: 0)
    const int32_t stride2 = stride * 2;
    uint8_t* score_ptr = reinterpret_cast<uint8_t*>(scores) + lane8 + 2 * (1 + kRelShift);
    uint8_t* move_ptr  = moves + lane4 + (1 + kRelShift);

    // row 0 into ring slot 0
#pragma unroll
    for (int p = 0; p < NP; p++) lds_store_u64(ring_base + a1[p], P01[p], P23[p]);

    // horizontal max-plus scan of all passes' candidates; cu = carry-in as element t = -1 of u; leaves the row in P*
    auto scan_row = [&](const uint32_t (&s01)[NP], const uint32_t (&s23)[NP], int32_t cu) {
        uint32_t pm01[NP], pm23[NP];
        int32_t in[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const uint32_t u01 = pk_sub(s01[p], K01[p]), u23 = pk_sub(s23[p], K23[p]);
            pm01[p] = pk_max(u01, (u01 << 16) | 0x8000u);
            pm23[p] = pk_max(u23, (u23 << 16) | 0x8000u);
            in[p]   = wave_inclusive_max((int32_t)pk_max(pm01[p], pm23[p]) >> 16); // inclusive maximum over the lanes' quads
        }
        int32_t c = cu;
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const int32_t ex   = max(wave_shr1(in[p], c), c); // lane 0: the carry-in alone
            const uint32_t e2  = __builtin_amdgcn_perm((uint32_t)ex, (uint32_t)ex, 0x01000100u);
            const uint32_t m1b = __builtin_amdgcn_perm(pm01[p], pm01[p], 0x03020302u); // max(u0, u1) in both halves
            P01[p] = pk_add(pk_max(pm01[p], e2), K01[p]);
            P23[p] = pk_add(pk_max(pk_max(pm23[p], m1b), e2), K23[p]);
            // the next pass continues behind this one's last cell
            if (p + 1 < NP) c = max(__builtin_amdgcn_readlane(in[p], kWave - 1), c);
        }
    };
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    auto costs = [&](uint32_t rd4, uint32_t base4, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd4 ^ base4;
        const uint32_t x01 = __builtin_amdgcn_perm(0u, x, 0x0c010c00u);
        const uint32_t x23 = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
        c01 = pk_mad_u16(pk_min_u16(x01, ONE2), DIF2, MAT2);
        c23 = pk_mad_u16(pk_min_u16(x23, ONE2), DIF2, MAT2);
    };
    // diagonal / vertical candidates of a quad from one predecessor row: q01 / q23 = its cells of the quad's columns,
    // s0x = its cell of the column left of them in the HIGH half
    auto from_pred = [&](uint32_t s0x, uint32_t q01, uint32_t q23, uint32_t c01, uint32_t c23, uint32_t& D01, uint32_t& D23,
                         uint32_t& V01, uint32_t& V23) {
        D01 = pk_add(__builtin_amdgcn_alignbit(q01, s0x, 16), c01);
        D23 = pk_add(__builtin_amdgcn_alignbit(q23, q01, 16), c23);
        V01 = pk_add(q01, GAP2);
        V23 = pk_add(q23, GAP2);
    };
    auto pack_moves = [&](uint32_t m01, uint32_t m23) -> uint32_t { return __builtin_amdgcn_perm(m23, m01, 0x06040200u); };
    // timing ablations (GWHIP_DEBUG, debug instantiation only; results are garbage): bit 26 no score-row stores, bit 27 no
    // move-row stores
    const bool abl       = (dbg & (1 << 14)) != 0; // (the store ablations only with bit 14: bits 26 / 27 are topsort selectors too)
    const bool st_scores = !(abl && (dbg & (1 << 26))), st_moves = !(abl && (dbg & (1 << 27)));
    // the finished row (P*) of row r with H[r][0] = c0: HBM score row, ring slot r & 3, move bytes. Both HBM stores are
    // streaming stores (round 5, same-box A/B on the 1024 windows: 59.0 -> 55.2 ms; the matrices of a full-band batch are 10 GB,
    // far beyond every cache level, and the walk reads a sliver of the move rows)
    auto store_row = [&](int32_t r, int32_t c0, const uint32_t (&mv)[NP], bool need_scores) {
        score_ptr += stride2;
        move_ptr += stride;
        const uint32_t sbase = ring_base + (((uint32_t)r & (kWdSlots - 1)) * kWdSlotBytes);
        if (need_scores && st_scores) // (wave-uniform: one row in fifty)
        {
#pragma unroll
            for (int p = 0; p < NP - 1; p++) gstore_nt_u64(score_ptr + 512 * p, P01[p], P23[p]);
            if (act) gstore_nt_u64(score_ptr + 512 * (NP - 1), P01[NP - 1], P23[NP - 1]);
        }
#pragma unroll
        for (int p = 0; p < NP - 1; p++)
        {
            lds_store_u64(sbase + a1[p], P01[p], P23[p]);
            if (st_moves) __builtin_nontemporal_store(mv[p], reinterpret_cast<uint32_t*>(move_ptr + 256 * p));
        }
        lds_store_u64(sbase + a1[NP - 1], P01[NP - 1], P23[NP - 1]);
        if (act && st_moves) __builtin_nontemporal_store(mv[NP - 1], reinterpret_cast<uint32_t*>(move_ptr + 256 * (NP - 1)));
        gstore_u16_lane0_below(score_ptr, (uint32_t)c0); // column 0
        c0ring  = lane == (int)((uint32_t)r & (kWdSlots - 1)) ? c0 : c0ring;
        prev_c0 = c0;
    };

    // ---------------- kind 0: one predecessor, the previous row, in registers ----------------
    auto reg_row = [&](int32_t r, uint32_t base4, bool need_scores) {
        uint32_t s01[NP], s23[NP], D01[NP], D23[NP], V01[NP], V23[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            // the cell left of the lane's quad: the previous lane's last cell; lane 0: column 0 / the previous pass's last cell
            const int32_t first = p == 0 ? (int32_t)((uint32_t)prev_c0 << 16) : __builtin_amdgcn_readlane((int32_t)P23[p > 0 ? p - 1 : 0], kWave - 1);
            const uint32_t s0x  = (uint32_t)wave_shr1((int32_t)P23[p], first);
            uint32_t c01, c23;
            costs(rd[p], base4, c01, c23);
            from_pred(s0x, P01[p], P23[p], c01, c23, D01[p], D23[p], V01[p], V23[p]);
            s01[p] = pk_max(D01[p], V01[p]);
            s23[p] = pk_max(D23[p], V23[p]);
        }
        const int32_t c0 = prev_c0 + gap_score;
        scan_row(s01, s23, c0 + gap_score);
        // move = H == D ? 3 : H == V ? 2 : 1   ==  3 + [H != D] * (-1 - [H != V])
        uint32_t mv[NP];
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const uint32_t m01 = pk_mad_u16(nz(P01[p], D01[p]), pk_mad_u16(nz(P01[p], V01[p]), NEG1, NEG1), THREE2);
            const uint32_t m23 = pk_mad_u16(nz(P23[p], D23[p]), pk_mad_u16(nz(P23[p], V23[p]), NEG1, NEG1), THREE2);
            mv[p]              = pack_moves(m01, m23);
        }
        store_row(r, c0, mv, need_scores);
    };

    // ---------------- kind 4: any predecessors, from the HBM matrix, same packed arithmetic, moves undecided ----------------
    auto general_row = [&](int32_t r, uint32_t base4) {
        const RowInfo<true> ri   = uniform_row(rowinfo[r]);
        const int32_t pred_count = ri.cnt();
        const int32_t node_id    = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t k) -> int32_t {
            if (pred_count == 0) return 0; // a source: the virtual row 0 (:228)
            if (k < 3) return ri.pred(k);
            return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + k]] + 1);
        };
        wave_sync(); // rows this wavefront stored are read back by other lanes
        uint32_t c01[NP], c23[NP], s01[NP], s23[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) costs(rd[p], base4, c01[p], c23[p]);
        int32_t c0m      = -32768;
        const int32_t np = max(pred_count, 1);
        for (int32_t k = 0; k < np; k++)
        {
            const uint8_t* rowp = reinterpret_cast<const uint8_t*>(scores) + (int64_t)pred_row(k) * stride2;
            uint32_t x[NP];
            uint2 q[NP];
#pragma unroll
            for (int p = 0; p < NP; p++)
            {
                const uint32_t off = p == NP - 1 ? ld_last : (uint32_t)(512 * p + lane8);
                x[p] = *reinterpret_cast<const uint32_t*>(rowp + off + 4); // elements (column - 1, column) of the column left of the quad
                q[p] = *reinterpret_cast<const uint2*>(rowp + off + 8);
            }
            c0m = max(c0m, wave_first((int32_t)x[0]) >> 16); // lane 0, pass 0: column 0 of that row
#pragma unroll
            for (int p = 0; p < NP; p++)
            {
                uint32_t D01, D23, V01, V23;
                from_pred(x[p], q[p].x, q[p].y, c01[p], c23[p], D01, D23, V01, V23);
                const uint32_t t01 = pk_max(D01, V01), t23 = pk_max(D23, V23);
                s01[p] = k == 0 ? t01 : pk_max(s01[p], t01);
                s23[p] = k == 0 ? t23 : pk_max(s23[p], t23);
            }
        }
        const int32_t c0 = c0m + gap_score;
        scan_row(s01, s23, c0 + gap_score);
        uint32_t mv[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) mv[p] = 0u;
        store_row(r, c0, mv, true);
    };

    // ---------------- the rows ----------------
    int32_t r = 1;
    while (r <= graph_count)
    {
        // descriptors of rows r .. r + 63, one per lane
        const int32_t r0 = r;
        uint32_t D0v, D1v;
        {
            const int32_t rr    = min(r0 + lane, graph_count);
            const uint64_t w    = rowinfo[rr].w;
            const uint32_t kind = (uint32_t)(w >> kKindShift) & 7u;
            const uint32_t cnt  = (uint32_t)(w >> 8) & 0x3fu;
            const uint32_t p0 = (uint32_t)(w >> 24) & 0xfffu, p1 = (uint32_t)(w >> 36) & 0xfffu, p2 = (uint32_t)(w >> 48) & 0xfffu;
            const uint32_t slots = (p0 & 3u) | ((p1 & 3u) << 3) | ((p2 & 3u) << 6);
            const uint32_t dists = (((uint32_t)rr - p0) & 7u) | ((((uint32_t)rr - p1) & 7u) << 3) | ((((uint32_t)rr - p2) & 7u) << 6);
            D0v = kind | ((uint32_t)(w >> 63) << 3) | (slots << 12) | (dists << 21) | ((cnt <= 3 ? cnt : 0u) << 30); // bit 3: scores to HBM
            D1v = ((uint32_t)w & 0xffu) * 0x01010101u;
        }
        D0v = (r0 + lane <= graph_count) ? D0v : 7u; // rows past the end read as kind 7 = "end of block"
        int32_t k      = 0;
        uint32_t d0    = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, 0);
        uint32_t base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, 0);
        uint32_t kind  = d0 & 7u;
        auto advance = [&]() {
            r++;
            k++;
            const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int32_t)D0v, k & (kWave - 1));
            base4 = (uint32_t)__builtin_amdgcn_readlane((int32_t)D1v, k & (kWave - 1));
            d0    = k == kWave ? 7u : nd;
            kind  = d0 & 7u;
        };
        for (;;)
        {
            while (kind == 0)
            {
                reg_row(r, base4, (d0 & 8u) != 0);
                advance();
            }
            if (kind == 7u) break;
            if (kind == 2 || kind == 3)
            {
                // ===== predecessors from the LDS ring =====
                const uint32_t sl0 = (d0 >> 12) & 7u, dd0 = (d0 >> 21) & 7u;
                struct PredCells { uint32_t x[NP]; uint2 q[NP]; int32_t c0; };
                auto load_pred = [&](uint32_t slot) -> PredCells {
                    PredCells pc;
                    const uint32_t b = ring_base + slot * kWdSlotBytes;
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        pc.x[p] = lds_load_u32(b + a0[p]);
                        pc.q[p] = lds_load_u64(b + a1[p]);
                    }
                    pc.c0   = __builtin_amdgcn_readlane(c0ring, (int)slot);
                    pc.x[0] = is_lane0 ? ((uint32_t)pc.c0 << 16) : pc.x[0]; // column 0 is not in the slot
                    return pc;
                };
                if (kind == 2)
                {
                    const PredCells pc = load_pred(sl0);
                    uint32_t s01[NP], s23[NP], D01[NP], D23[NP], V01[NP], V23[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        uint32_t c01, c23;
                        costs(rd[p], base4, c01, c23);
                        from_pred(pc.x[p], pc.q[p].x, pc.q[p].y, c01, c23, D01[p], D23[p], V01[p], V23[p]);
                        s01[p] = pk_max(D01[p], V01[p]);
                        s23[p] = pk_max(D23[p], V23[p]);
                    }
                    const int32_t c0 = pc.c0 + gap_score;
                    scan_row(s01, s23, c0 + gap_score);
                    // move = H == D ? 2 d + 1 : H == V ? 2 d : 1  ==  (2 d + 1) + [H != D] * (-1 + [H != V] * (1 - 2 d))
                    const uint32_t cD = pk_dup((int32_t)(2u * dd0 + 1u)), cV = pk_dup(1 - (int32_t)(2u * dd0));
                    uint32_t mv[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        const uint32_t m01 = pk_mad_u16_vvs(nz(P01[p], D01[p]), pk_mad_u16_vsv(nz(P01[p], V01[p]), cV, NEG1), cD);
                        const uint32_t m23 = pk_mad_u16_vvs(nz(P23[p], D23[p]), pk_mad_u16_vsv(nz(P23[p], V23[p]), cV, NEG1), cD);
                        mv[p]              = pack_moves(m01, m23);
                    }
                    store_row(r, c0, mv, (d0 & 8u) != 0);
                }
                else
                {
                    const uint32_t cnt3 = d0 >> 30;          // 2, 3, or 0 = more than three
                    const int32_t cnt   = cnt3 == 2 ? 2 : 3; // predecessors in the descriptor
                    const uint32_t sl1 = (d0 >> 15) & 7u, sl2 = cnt > 2 ? (d0 >> 18) & 7u : sl0;
                    const uint32_t dd1 = (d0 >> 24) & 7u, dd2 = (d0 >> 27) & 7u;
                    const PredCells p0c = load_pred(sl0);
                    const PredCells p1c = load_pred(sl1);
                    PredCells p2c       = p0c;
                    if (cnt > 2) p2c = load_pred(sl2);
                    int32_t c0m = max(p0c.c0, p1c.c0);
                    if (cnt > 2) c0m = max(c0m, p2c.c0);
                    // best diagonal / vertical candidate over the predecessors and, per cell, the move of the first slot that
                    // attains it: A (diagonal) = mD0 + n0 * (E1 + n1 * E2), E1 = 2 (d1 - d0), E2 = 2 (d2 - d1), n_k = [slot k
                    // misses the maximum]; B (vertical) likewise from mV0
                    const uint32_t mD0 = pk_dup((int32_t)(2u * dd0 + 1u)), mV0 = pk_dup((int32_t)(2u * dd0));
                    const uint32_t E1v = pin_vgpr(pk_dup(2 * ((int32_t)dd1 - (int32_t)dd0)));
                    const uint32_t E2v = pin_vgpr(pk_dup(2 * ((int32_t)dd2 - (int32_t)dd1)));
                    uint32_t bD01[NP], bD23[NP], bV01[NP], bV23[NP], A01[NP], A23[NP], B01[NP], B23[NP], c01[NP], c23[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        costs(rd[p], base4, c01[p], c23[p]);
                        uint32_t D0a, D0b, V0a, V0b, D1a, D1b, V1a, V1b;
                        from_pred(p0c.x[p], p0c.q[p].x, p0c.q[p].y, c01[p], c23[p], D0a, D0b, V0a, V0b);
                        from_pred(p1c.x[p], p1c.q[p].x, p1c.q[p].y, c01[p], c23[p], D1a, D1b, V1a, V1b);
                        bD01[p] = pk_max(D0a, D1a); bD23[p] = pk_max(D0b, D1b); bV01[p] = pk_max(V0a, V1a); bV23[p] = pk_max(V0b, V1b);
                        if (cnt > 2)
                        {
                            uint32_t D2a, D2b, V2a, V2b;
                            from_pred(p2c.x[p], p2c.q[p].x, p2c.q[p].y, c01[p], c23[p], D2a, D2b, V2a, V2b);
                            bD01[p] = pk_max(bD01[p], D2a); bD23[p] = pk_max(bD23[p], D2b); bV01[p] = pk_max(bV01[p], V2a); bV23[p] = pk_max(bV23[p], V2b);
                            A01[p] = pk_mad_u16_vvs(nz(bD01[p], D0a), pk_mad_u16(nz(bD01[p], D1a), E2v, E1v), mD0);
                            A23[p] = pk_mad_u16_vvs(nz(bD23[p], D0b), pk_mad_u16(nz(bD23[p], D1b), E2v, E1v), mD0);
                            B01[p] = pk_mad_u16_vvs(nz(bV01[p], V0a), pk_mad_u16(nz(bV01[p], V1a), E2v, E1v), mV0);
                            B23[p] = pk_mad_u16_vvs(nz(bV23[p], V0b), pk_mad_u16(nz(bV23[p], V1b), E2v, E1v), mV0);
                        }
                        else
                        {
                            A01[p] = pk_mad_u16_vvs(nz(bD01[p], D0a), E1v, mD0); A23[p] = pk_mad_u16_vvs(nz(bD23[p], D0b), E1v, mD0);
                            B01[p] = pk_mad_u16_vvs(nz(bV01[p], V0a), E1v, mV0); B23[p] = pk_mad_u16_vvs(nz(bV23[p], V0b), E1v, mV0);
                        }
                    }
                    if (cnt3 == 0)
                    {
                        // predecessors 3..5 (rows from the side table, cells from the ring): they raise the maxima; where only
                        // they attain a maximum the first attaining slot is >= 3, whose distance the pass does not track -> move 0
                        const uint64_t xe     = wave_first64(xpred[r & 255]);
                        const int32_t cnt_all = (int32_t)((xe >> 13) & 63u);
                        for (int32_t kk = 3; kk < cnt_all; kk++)
                        {
                            const PredCells pkc = load_pred((uint32_t)xpred_row(xe, kk) & (kWdSlots - 1));
                            c0m                 = max(c0m, pkc.c0);
#pragma unroll
                            for (int p = 0; p < NP; p++)
                            {
                                uint32_t Da, Db, Va, Vb;
                                from_pred(pkc.x[p], pkc.q[p].x, pkc.q[p].y, c01[p], c23[p], Da, Db, Va, Vb);
                                const uint32_t fD01 = pk_max(bD01[p], Da), fD23 = pk_max(bD23[p], Db), fV01 = pk_max(bV01[p], Va), fV23 = pk_max(bV23[p], Vb);
                                // A *= [the maximum so far == the new maximum]
                                A01[p] = pk_mad_u16(nz(bD01[p], fD01), pk_sub(0u, A01[p]), A01[p]); A23[p] = pk_mad_u16(nz(bD23[p], fD23), pk_sub(0u, A23[p]), A23[p]);
                                B01[p] = pk_mad_u16(nz(bV01[p], fV01), pk_sub(0u, B01[p]), B01[p]); B23[p] = pk_mad_u16(nz(bV23[p], fV23), pk_sub(0u, B23[p]), B23[p]);
                                bD01[p] = fD01; bD23[p] = fD23; bV01[p] = fV01; bV23[p] = fV23;
                            }
                        }
                    }
                    uint32_t s01[NP], s23[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                    {
                        s01[p] = pk_max(bD01[p], bV01[p]);
                        s23[p] = pk_max(bD23[p], bV23[p]);
                    }
                    const int32_t c0 = c0m + gap_score;
                    scan_row(s01, s23, c0 + gap_score);
                    // move = H == bestD ? A : H == bestV ? B : 1
                    auto move_of = [&](uint32_t H, uint32_t bD, uint32_t bV, uint32_t A, uint32_t B) -> uint32_t {
                        const uint32_t t1 = pk_mad_u16(nz(H, bV), pk_sub(ONE2, B), B);
                        return pk_mad_u16(nz(H, bD), pk_sub(t1, A), A);
                    };
                    uint32_t mv[NP];
#pragma unroll
                    for (int p = 0; p < NP; p++)
                        mv[p] = pack_moves(move_of(P01[p], bD01[p], bV01[p], A01[p], B01[p]), move_of(P23[p], bD23[p], bV23[p], A23[p], B23[p]));
                    store_row(r, c0, mv, (d0 & 8u) != 0);
                }
            }
            else
                general_row(r, base4);
            advance();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Full-band NW through the packed pass: forward pass, sink selection (:320-337), traceback over the move bytes
// (poa_traceback_moves.h; undecided cells by recomputation from the score matrix, :340-445). `handled` stays false --
// nothing touched -- when a precondition fails; the caller then runs nw_full.
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ int32_t nw_full_packed(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count, const uint8_t* read,
                                                  int32_t read_length, int16_t* scores, int32_t scores_width, uint8_t* moves, uint8_t* ring,
                                                  int32_t ring_bytes, const uint64_t* xpred, int32_t* alignment_graph, int32_t* alignment_read,
                                                  int32_t gap_score, int32_t mismatch_score, int32_t match_score, uint64_t& cells, int32_t dbg,
                                                  bool& handled, PhaseClock& pc)
{
    const int lane = threadIdx.x & (kWave - 1);
    handled        = false;
    const int32_t u_span = 1024 * abs(gap_score); // u-space offset of the last column
    const bool ok = read_length >= 1 && read_length <= 1023 && scores_width >= read_length + 1 + kCellsPerLane && moves != nullptr &&
                    xpred != nullptr && ring_bytes >= kWdSlots * kWdSlotBytes && ring_bytes >= kMtBytes && !(dbg & 256) && gap_score < 0 &&
                    abs(gap_score) <= 30 && abs(match_score) <= 100 && abs(mismatch_score) <= 100 &&
                    // no packed operation can leave int16: the largest score (all matches) plus the u-space offset of the last
                    // column, and the smallest (every step at the worst penalty)
                    max(match_score, 0) * min(read_length, graph_count) + u_span + abs(match_score) <= 32767 &&
                    (graph_count + read_length) * min(min(gap_score, mismatch_score), 0) >= -32768 + 256;
    if (!ok) return 0;
    handled = true;
    cells += (uint64_t)graph_count * (uint64_t)read_length;
    for (int32_t j = lane; j <= read_length; j += kWave) scores[j + kRelShift] = (int16_t)(j * gap_score); // row 0 (:176-179)
    const int32_t np = (read_length + 255) >> 8;
    if (np == 4)
        full_forward_moves<IdT, 4>(g, rowinfo, graph_count, read, read_length, scores, scores_width, moves, ring, xpred, gap_score, mismatch_score, match_score, dbg);
    else if (np == 3)
        full_forward_moves<IdT, 3>(g, rowinfo, graph_count, read, read_length, scores, scores_width, moves, ring, xpred, gap_score, mismatch_score, match_score, dbg);
    else if (np == 2)
        full_forward_moves<IdT, 2>(g, rowinfo, graph_count, read, read_length, scores, scores_width, moves, ring, xpred, gap_score, mismatch_score, match_score, dbg);
    else
        full_forward_moves<IdT, 1>(g, rowinfo, graph_count, read, read_length, scores, scores_width, moves, ring, xpred, gap_score, mismatch_score, match_score, dbg);
    wave_sync(); // matrices complete and visible to every lane
    pc.tick(kPhForward);

    // sink selection (:320-337): first row with the strictly greatest H(row, L) among sink rows
    int32_t best = Limits<int16_t>::min, best_i = 0;
    for (int32_t idx = 1 + lane; idx <= graph_count; idx += kWave)
    {
        if (rowinfo[idx].sink())
        {
            const int32_t s = scores[(int64_t)idx * scores_width + read_length + kRelShift];
            if (best < s) { best = s; best_i = idx; }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        const int32_t ob = __shfl_xor(best, off), oi = __shfl_xor(best_i, off);
        if (ob > best || (ob == best && oi != 0 && (best_i == 0 || oi < best_i))) { best = ob; best_i = oi; }
    }
    // the walk sees a band that is the whole row: band start 0 in every row, every column committed
    BandedCtx<int16_t> b;
    b.scores     = scores;
    b.ring       = nullptr;
    b.ring_rows  = 0;
    b.stride     = scores_width;
    b.band_width = scores_width; // > max_column: band_start_for_row() is 0 for every row
    b.band_shift = 0;
    b.max_column = read_length + 1;
    b.gradient   = __fdiv_rn((float)(read_length + 1), (float)(graph_count + 1));
    b.min_score  = Limits<int16_t>::min;
    return traceback_moves<int16_t, IdT, RowInfo<true>, false>(b, g, rowinfo, graph_count, read, read_length, wave_first(best_i), alignment_graph,
                                                               alignment_read, gap_score, mismatch_score, match_score, 0, ring, moves);
}

} // namespace gwhip
