// poa_forward_moves_wide.h -- banded NW forward pass for the 384- and 512-column bands with int16 scores (round 4):
// the row of poa_forward_moves.h (row kinds, register descriptors, move bytes, two phases) with TWO register passes of
// 256 columns per row. What it computes is cudapoa_nw_banded.cuh:269-408 (restated in oracle/poa_nw.inc), which is
// width-agnostic; the reference accepts every multiple of 128 (cudapoa/src/batch.cu:41).
//
//   * Every lane owns two quads of a row: pass A = columns bs + 4 l + 1..4, pass B = the same 256 columns further right
//     (band 384: pass B lives in lanes 0..31, the other lanes compute cells right of the band that nobody reads -- a
//     prefix scan runs left to right -- and their stores are masked). Both passes go through one instruction stream:
//     one descriptor fetch, one branch and one guard store per row, two cross-lane scans chained by pass A's maximum.
//   * The LDS ring keeps 4 rows of 1024 absolute column slots (2048 B per row: band + sentinel cells behind the band end;
//     4 x 2048 = the 8 KB of the 8 x 1024 ring of the 256-column pass, so the block still fits four to a CU). Slot r & 3
//     still holds row r - 4 until row r is stored (a wavefront's LDS operations execute in order), so predecessors up to
//     4 rows up are served: 98.4 % of the rows of the metric windows (tools/row_distance_stats.py,
//     profiles/r03_row_distance_model.json); the rest take the general routine against the HBM matrix.
//   * A band move of one quad shifts both passes by one lane; pass A's last lane takes pass B's first quad (v_readlane).
//
// Move bytes, the traceback (poa_traceback_moves.h is width-agnostic) and all preconditions are those of
// poa_forward_moves.h with the band width in place of 256 (checked by nw_banded).
#pragma once

namespace gwhip
{

constexpr int kWdSlots     = 4;    // ring rows
constexpr int kWdSlotBytes = 2048; // 1024 column slots x int16
constexpr int kWdMaxDist   = 4;    // see above: a row is read before its slot is overwritten

template <typename IdT, int BW>
__device__ __forceinline__ void banded_forward_moves_wide(const GraphView<IdT>& g, RowInfo<true>* rowinfo, int32_t graph_count,
                                                          const uint8_t* lds_read, int16_t* scores, uint8_t* moves, uint8_t* ring,
                                                          const uint64_t* xpred, int32_t max_column, int32_t gap_score,
                                                          int32_t mismatch_score, int32_t match_score, int32_t dbg)
{
    static_assert(BW == 384 || BW == 512, "band widths of the two-pass packed pass");
    constexpr int32_t band_width = BW;
    constexpr int32_t stride     = band_width + kRightPad;
    constexpr int kBLanes        = (BW - 256) / kCellsPerLane; // lanes that own band cells of pass B
    constexpr uint32_t kMask     = kWdSlotBytes - 1;
    const int lane               = threadIdx.x & (kWave - 1);
    const bool bandB_lane        = lane < kBLanes;
    const int32_t lane4 = lane * 4, lane8 = lane * 8;
    const int32_t min_score = Limits<int16_t>::min / 2;

    const int32_t first_moved = classify_kinds<kWdMaxDist>(rowinfo, graph_count, lane, xpred, dbg);
    wave_sync();

    const uint32_t MIN2   = pin_vgpr(pk_dup(min_score));
    const uint32_t SENT2  = pin_vgpr(pk_dup(kPkSentinel));
    const uint32_t GAP2   = pin_vgpr(pk_dup(gap_score));
    const uint32_t MAT2   = pin_vgpr(pk_dup(match_score));
    const uint32_t DIF2   = pin_vgpr(pk_dup(mismatch_score - match_score));
    const uint32_t ONE2   = pin_vgpr(0x00010001u);
    const uint32_t THREE2 = pin_vgpr(0x00030003u);
    const uint32_t NEG1   = pin_vgpr(0xffffffffu);
    // t * gap for the lane's cells: pass A t = 4 lane + k, pass B t = 256 + 4 lane + k
    const uint32_t KA01 = pk_make((lane4 + 0) * gap_score, (lane4 + 1) * gap_score);
    const uint32_t KA23 = pk_make((lane4 + 2) * gap_score, (lane4 + 3) * gap_score);
    const uint32_t KB01 = pk_make((256 + lane4 + 0) * gap_score, (256 + lane4 + 1) * gap_score);
    const uint32_t KB23 = pk_make((256 + lane4 + 2) * gap_score, (256 + lane4 + 3) * gap_score);
    const uint32_t ring_base = lds_addr(ring);
    const uint32_t read_base = lds_addr(lds_read);
    // guard store: lanes 0..15 write sentinel cells for columns band_end + 1 .. + 64, lane 16 the quad that ends in the
    // left-boundary slot (column band_start); byte offsets relative to the lane's own pass-A offset
    // (lanes 17..63 take part in the same store -- masking them off costs two writes of EXEC per row, ~45 cycles for a lone
    // wavefront -- and write quads of columns band_start - 7 and below, lane l at byte 2 band_start - 16 - 8 (l - 17): no reader
    // of this row looks left of column band_start - 3, and the lowest of them, 191 columns down, still lies clear of the
    // sentinel cells when the slot wraps)
    const uint32_t guard_off  = lane < 16 ? (uint32_t)(2 * BW) : (lane == 16 ? (uint32_t)-136 : (uint32_t)(120 - 16 * lane));
    const bool is_lane16      = lane == 16;
    const bool is_lastB       = lane == kBLanes - 1; // the band's last lane (pass B)
    const uint32_t move_keepA = lane == 0 ? 0xffffff00u : 0xffffffffu; // the band's first cell stays undecided
    const uint32_t GUARD_HI_MIN = pin_vgpr(is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)min_score << 16)) : pk_dup(kPkSentinel));
    const uint32_t S0X_MIN      = (uint32_t)min_score << 16;

    // state carried from row to row: the previous row's cells of both passes
    uint32_t PA01 = pk_make((lane4 + 1) * gap_score, (lane4 + 2) * gap_score); // row 0: H[0][x] = x * gap
    uint32_t PA23 = pk_make((lane4 + 3) * gap_score, (lane4 + 4) * gap_score);
    uint32_t PB01 = pk_make((256 + lane4 + 1) * gap_score, (256 + lane4 + 2) * gap_score);
    uint32_t PB23 = pk_make((256 + lane4 + 3) * gap_score, (256 + lane4 + 4) * gap_score);
    int32_t prev_rel0 = 0;
    // per-lane values that only change when the band moves
    uint32_t rdA  = lds_load_u32(read_base + lane4);
    uint32_t rdAn = lds_load_u32(read_base + lane4 + 4);
    uint32_t rdB  = lds_load_u32(read_base + 256 + lane4);
    uint32_t rdBn = lds_load_u32(read_base + 256 + lane4 + 4);
    uint32_t a1A  = (uint32_t)lane8;
    uint32_t a1B  = (uint32_t)lane8 + 512u;
    uint32_t ga   = (a1A + guard_off) & kMask;
    uint8_t* score_ptr = reinterpret_cast<uint8_t*>(scores) + lane8 + 2 * (1 + kRelShift);
    uint8_t* move_ptr  = moves + lane4 + (1 + kRelShift);

    // row 0 into ring slot 0
    lds_store_u64(ring_base + a1A, PA01, PA23);
    if (BW == 512 || bandB_lane) lds_store_u64(ring_base + a1B, PB01, PB23);
    lds_store_guard(ring_base + ga, SENT2, is_lane16 ? (((uint32_t)kPkSentinel & 0xffffu) | (0u << 16)) : SENT2);

    // horizontal max-plus scan of both passes' candidates; cu = carry-in as element t = -1 of u; leaves the row in P*
    auto scan_row = [&](uint32_t sA01, uint32_t sA23, uint32_t sB01, uint32_t sB23, int32_t cu) {
        const uint32_t uA01 = pk_sub(sA01, KA01), uA23 = pk_sub(sA23, KA23);
        const uint32_t uB01 = pk_sub(sB01, KB01), uB23 = pk_sub(sB23, KB23);
        const uint32_t pA01 = pk_max(uA01, (uA01 << 16) | 0x8000u), pA23 = pk_max(uA23, (uA23 << 16) | 0x8000u);
        const uint32_t pB01 = pk_max(uB01, (uB01 << 16) | 0x8000u), pB23 = pk_max(uB23, (uB23 << 16) | 0x8000u);
        const int32_t mA    = (int32_t)pk_max(pA01, pA23) >> 16; // max(u0..u3) of the lane's pass-A quad
        const int32_t mB    = (int32_t)pk_max(pB01, pB23) >> 16;
        const int32_t inA   = wave_inclusive_max(mA);
        const int32_t inB   = wave_inclusive_max(mB);
        const int32_t exA   = max(wave_shr1(inA, cu), cu); // lane 0: the carry-in alone
        // pass B continues behind pass A's last cell: its carry-in is the maximum over the carry-in and all of pass A
        const int32_t cuB   = max(__builtin_amdgcn_readlane(inA, kWave - 1), cu);
        const int32_t exB   = max(wave_shr1(inB, cuB), cuB);
        const uint32_t eA2  = __builtin_amdgcn_perm((uint32_t)exA, (uint32_t)exA, 0x01000100u);
        const uint32_t eB2  = __builtin_amdgcn_perm((uint32_t)exB, (uint32_t)exB, 0x01000100u);
        const uint32_t mA1  = __builtin_amdgcn_perm(pA01, pA01, 0x03020302u); // max(u0,u1) in both halves
        const uint32_t mB1  = __builtin_amdgcn_perm(pB01, pB01, 0x03020302u);
        PA01 = pk_add(pk_max(pA01, eA2), KA01);
        PA23 = pk_add(pk_max(pk_max(pA23, mA1), eA2), KA23);
        PB01 = pk_add(pk_max(pB01, eB2), KB01);
        PB23 = pk_add(pk_max(pk_max(pB23, mB1), eB2), KB23);
    };
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    auto costs = [&](uint32_t rd4, uint32_t base4, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd4 ^ base4;
        const uint32_t x01 = __builtin_amdgcn_perm(0u, x, 0x0c010c00u);
        const uint32_t x23 = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
        c01 = pk_mad_u16(pk_min_u16(x01, ONE2), DIF2, MAT2);
        c23 = pk_mad_u16(pk_min_u16(x23, ONE2), DIF2, MAT2);
    };
    auto from_pred = [&](uint32_t s0x, uint32_t q01, uint32_t q23, uint32_t c01, uint32_t c23, uint32_t& D01, uint32_t& D23,
                         uint32_t& V01, uint32_t& V23) {
        D01 = pk_add(__builtin_amdgcn_alignbit(q01, s0x, 16), c01);
        D23 = pk_add(__builtin_amdgcn_alignbit(q23, q01, 16), c23);
        V01 = pk_add(q01, GAP2);
        V23 = pk_add(q23, GAP2);
    };
    auto pack_moves = [&](uint32_t m01, uint32_t m23) -> uint32_t { return __builtin_amdgcn_perm(m23, m01, 0x06040200u); };
    // the finished row (P*) of row r: HBM score row, ring slot r & 3 with its guard quad, and its move bytes
    auto store_row = [&](auto bs0_tag, int32_t r, int32_t rel0_val, uint32_t mvA, uint32_t mvB) {
        constexpr bool BS0 = decltype(bs0_tag)::value;
        score_ptr += stride * 2;
        move_ptr += stride;
        const uint32_t sbase = ring_base + (((uint32_t)r & (kWdSlots - 1)) * kWdSlotBytes);
        gstore_nt_u64(score_ptr, PA01, PA23);
        lds_store_u64(sbase + a1A, PA01, PA23);
        if (BW == 512 || bandB_lane)
        {
            gstore_nt_u64(score_ptr + 512, PB01, PB23);
            lds_store_u64(sbase + a1B, PB01, PB23);
        }
        if constexpr (BS0)
        {
            const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
            lds_store_guard(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
            gstore_u16_lane0_below(score_ptr, (uint32_t)rel0_val);
            prev_rel0 = rel0_val;
        }
        else
            lds_store_guard(sbase + ga, SENT2, GUARD_HI_MIN);
        *reinterpret_cast<uint32_t*>(move_ptr) = mvA & move_keepA;
        if (BW == 512 || bandB_lane) *reinterpret_cast<uint32_t*>(move_ptr + 256) = mvB;
    };

    // ---------------- general row (kind 4): 32-bit arithmetic, previous row from registers, others from HBM ----------------
    auto general_row = [&](int32_t r, int32_t& prev_rel0_io) {
        const RowInfo<true> ri = uniform_row(rowinfo[r]);
        const int32_t bs       = ri.bs();
        const uint32_t base    = (uint32_t)ri.base();
        const int32_t prev_bs  = r > 1 ? uniform_row(rowinfo[r - 1]).bs() : 0;
        a1A  = (uint32_t)(2 * bs + lane8) & kMask;
        a1B  = (a1A + 512u) & kMask;
        ga   = (a1A + guard_off) & kMask;
        rdA  = lds_load_u32(read_base + bs + lane4);
        rdAn = lds_load_u32(read_base + bs + lane4 + 4);
        rdB  = lds_load_u32(read_base + bs + 256 + lane4);
        rdBn = lds_load_u32(read_base + bs + 256 + lane4 + 4);
        const int32_t pred_count = ri.cnt();
        const int32_t RA0 = pk_lo(PA01), RA1 = pk_hi(PA01), RA2 = pk_lo(PA23), RA3 = pk_hi(PA23);
        const int32_t RB0 = pk_lo(PB01), RB1 = pk_hi(PB01), RB2 = pk_lo(PB23), RB3 = pk_hi(PB23);
        bool synced = false;
        const int32_t node_id = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
        auto pred_row = [&](int32_t p) -> int32_t {
            if (pred_count == 0) return 0;
            if (p < 3) return ri.pred(p);
            return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1);
        };
        auto rel0_of = [&](int32_t prow) -> int32_t {
            if (prow == r - 1) return prev_rel0_io;
            const int32_t pbs = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
            if (pbs > 0) return min_score;
            if (!synced) { wave_sync(); synced = true; }
            return wave_first((int32_t)scores[(int64_t)prow * stride + kRelShift]);
        };
        int32_t fe = 0, rel0_val = min_score;
        if (pred_count == 0)
        {
            if (bs == 0) rel0_val = gap_score; // carry-in stays 0 (reference quirk)
        }
        else
        {
            if (bs > kCellsPerLane && pred_count == 1)
                fe = min_score + gap_score;
            else
            {
                int32_t penalty = min_score;
                for (int32_t p = 0; p < pred_count; p++) penalty = max(penalty, rel0_of(pred_row(p)));
                fe = penalty + gap_score;
            }
            if (bs == 0) rel0_val = fe;
        }
        // candidates of one pass (col_off = 0 / 256) from every predecessor
        auto pass_candidates = [&](int32_t col_off, uint32_t rd4, int32_t& s0, int32_t& s1, int32_t& s2, int32_t& s3) {
            const int32_t c   = bs + col_off + lane4;
            const int32_t cp0 = ((rd4 & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp1 = (((rd4 >> 8) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp2 = (((rd4 >> 16) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp3 = ((rd4 >> 24) == base) ? match_score : mismatch_score;
            s0 = s1 = s2 = s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = pred_row(p);
                int32_t S0, S1, S2, S3, S4;
                bool valid;
                if (prow == r - 1)
                {
                    // the previous row is in registers: virtual lane v of its 128 quads (pass A 0..63, pass B 64..127)
                    const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
                    valid              = c <= pend;
                    const int32_t v    = ((c - prev_bs) >> 2);
                    // the shuffles must run with every lane active
                    const int32_t a3m = __shfl(RA3, (v - 1) & 63), b3m = __shfl(RB3, (v - 1) & 63);
                    const int32_t left = ((v - 1) & 64) ? b3m : a3m;
                    S0 = (v == 0) ? prev_rel0_io : left;
                    const int32_t a0 = __shfl(RA0, v & 63), a1 = __shfl(RA1, v & 63), a2 = __shfl(RA2, v & 63), a3 = __shfl(RA3, v & 63);
                    const int32_t b0 = __shfl(RB0, v & 63), b1 = __shfl(RB1, v & 63), b2 = __shfl(RB2, v & 63), b3 = __shfl(RB3, v & 63);
                    const bool hiq = (v & 64) != 0;
                    S1 = hiq ? b0 : a0; S2 = hiq ? b1 : a1; S3 = hiq ? b2 : a2; S4 = hiq ? b3 : a3;
                }
                else
                {
                    const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                    const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                    valid              = !(c > pend || c < pbs);
                    if (!synced) { wave_sync(); synced = true; }
                    // no divergent branch: lanes outside the predecessor's band load the row's first quad and discard it
                    const int16_t* rowp = scores + (int64_t)prow * stride + (valid ? (c - pbs) : 0) + kRelShift;
                    S0 = rowp[0];
                    const Quad<int16_t> qd = *reinterpret_cast<const Quad<int16_t>*>(rowp + 1);
                    S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                    if (pbs > 0 && c == pbs) S0 = min_score; // relative-0 slot of a row whose band starts past column 0
                }
                const int32_t t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
                const int32_t t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
                const int32_t t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
                const int32_t t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
                if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
                else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
            }
        };
        int32_t a0, a1, a2, a3, b0, b1, b2, b3;
        pass_candidates(0, rdA, a0, a1, a2, a3);
        pass_candidates(256, rdB, b0, b1, b2, b3);
        scan_row(pk_make(a0, a1), pk_make(a2, a3), pk_make(b0, b1), pk_make(b2, b3), fe + gap_score);
        // stores (either flavour of left boundary)
        score_ptr += stride * 2;
        move_ptr += stride;
        const uint32_t sbase  = ring_base + (((uint32_t)r & (kWdSlots - 1)) * kWdSlotBytes);
        const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0_val << 16);
        gstore_nt_u64(score_ptr, PA01, PA23);
        lds_store_u64(sbase + a1A, PA01, PA23);
        *reinterpret_cast<uint32_t*>(move_ptr) = 0u;
        if (BW == 512 || bandB_lane)
        {
            gstore_nt_u64(score_ptr + 512, PB01, PB23);
            lds_store_u64(sbase + a1B, PB01, PB23);
            *reinterpret_cast<uint32_t*>(move_ptr + 256) = 0u;
        }
        lds_store_guard(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
        if (bs == 0) gstore_u16_lane0_below(score_ptr, (uint32_t)rel0_val);
        prev_rel0_io = rel0_val;
    };

    // ---------------- kinds 0 / 1: one predecessor, the previous row, in registers ----------------
    auto reg_row = [&](auto bs0_tag, auto moved_tag, int32_t r, uint32_t d0, uint32_t base4) {
        constexpr bool BS0   = decltype(bs0_tag)::value;
        constexpr bool MOVED = decltype(moved_tag)::value;
        uint32_t sA, qA01, qA23, sB, qB01, qB23;
        if constexpr (!MOVED)
        {
            sA   = (uint32_t)wave_shr1((int32_t)PA23, BS0 ? (int32_t)((uint32_t)prev_rel0 << 16) : (int32_t)S0X_MIN);
            sB   = (uint32_t)wave_shr1((int32_t)PB23, __builtin_amdgcn_readlane((int32_t)PA23, kWave - 1));
            qA01 = PA01; qA23 = PA23; qB01 = PB01; qB23 = PB23;
        }
        else
        {
            // band moved by one quad: the lane's columns are those of the next lane of the previous row; pass A's last lane
            // takes pass B's first quad
            sA   = PA23;
            sB   = PB23;
            qA01 = (uint32_t)wave_shl1((int32_t)PA01, __builtin_amdgcn_readlane((int32_t)PB01, 0));
            qA23 = (uint32_t)wave_shl1((int32_t)PA23, __builtin_amdgcn_readlane((int32_t)PB23, 0));
            qB01 = (uint32_t)wave_shl1((int32_t)PB01, (int32_t)pk_dup(kPkSentinel));
            qB23 = (uint32_t)wave_shl1((int32_t)PB23, (int32_t)pk_dup(kPkSentinel));
            a1A  = (a1A + 8) & kMask;
            a1B  = (a1B + 8) & kMask;
            ga   = (ga + 8) & kMask;
            rdA  = rdAn;
            rdB  = rdBn;
            const uint32_t nb = read_base + (((d0 >> 3) & 0x1ffu) << 2) + lane4 + 4;
            rdAn = lds_load_u32(nb);
            rdBn = lds_load_u32(nb + 256);
        }
        int32_t cu = min_score + 2 * gap_score, rel0_val = min_score;
        if constexpr (BS0)
        {
            rel0_val = max(min_score, prev_rel0) + gap_score;
            cu       = rel0_val + gap_score;
        }
        uint32_t cA01, cA23, cB01, cB23, DA01, DA23, VA01, VA23, DB01, DB23, VB01, VB23;
        costs(rdA, base4, cA01, cA23);
        costs(rdB, base4, cB01, cB23);
        from_pred(sA, qA01, qA23, cA01, cA23, DA01, DA23, VA01, VA23);
        from_pred(sB, qB01, qB23, cB01, cB23, DB01, DB23, VB01, VB23);
        uint32_t sA01 = pk_max(DA01, VA01), sA23 = pk_max(DA23, VA23);
        uint32_t sB01 = pk_max(DB01, VB01), sB23 = pk_max(DB23, VB23);
        if constexpr (MOVED) // the band's last lane: the chunk beyond the previous row's band
        {
            sB01 = is_lastB ? MIN2 : sB01;
            sB23 = is_lastB ? MIN2 : sB23;
        }
        scan_row(sA01, sA23, sB01, sB23, cu);
        // move = H == D ? 3 : H == V ? 2 : 1   ==  3 + [H != D] * (-1 - [H != V])
        const uint32_t mA01 = pk_mad_u16(nz(PA01, DA01), pk_mad_u16(nz(PA01, VA01), NEG1, NEG1), THREE2);
        const uint32_t mA23 = pk_mad_u16(nz(PA23, DA23), pk_mad_u16(nz(PA23, VA23), NEG1, NEG1), THREE2);
        const uint32_t mB01 = pk_mad_u16(nz(PB01, DB01), pk_mad_u16(nz(PB01, VB01), NEG1, NEG1), THREE2);
        const uint32_t mB23 = pk_mad_u16(nz(PB23, DB23), pk_mad_u16(nz(PB23, VB23), NEG1, NEG1), THREE2);
        const uint32_t mvA  = pack_moves(mA01, mA23);
        uint32_t mvB        = pack_moves(mB01, mB23);
        if constexpr (MOVED) mvB = is_lastB ? 0u : mvB;
        store_row(bs0_tag, r, rel0_val, mvA, mvB);
    };

    // ---------------- the rows of one phase ----------------
    auto run_rows = [&](auto bs0_tag, int32_t r_from, int32_t r_to) {
        constexpr bool BS0 = decltype(bs0_tag)::value;
        int32_t r = r_from;
        while (r <= r_to)
        {
            // descriptors of rows r .. r + 63, one per lane
            const int32_t r0 = r;
            uint32_t D0v, D1v;
            {
                const int32_t rr    = min(r0 + lane, graph_count);
                const uint64_t w    = rowinfo[rr].w;
                const uint32_t kind = (uint32_t)(w >> kKindShift) & 7u;
                const uint32_t cnt  = (uint32_t)(w >> 8) & 0x3fu;
                const uint32_t bs4  = (uint32_t)(w >> 15) & 0x1ffu;
                const uint32_t p0 = (uint32_t)(w >> 24) & 0xfffu, p1 = (uint32_t)(w >> 36) & 0xfffu, p2 = (uint32_t)(w >> 48) & 0xfffu;
                const uint32_t slots = (p0 & 3u) | ((p1 & 3u) << 3) | ((p2 & 3u) << 6);
                const uint32_t dists = (((uint32_t)rr - p0) & 7u) | ((((uint32_t)rr - p1) & 7u) << 3) | ((((uint32_t)rr - p2) & 7u) << 6);
                D0v = kind | (bs4 << 3) | (slots << 12) | (dists << 21) | ((cnt <= 3 ? cnt : 0u) << 30);
                D1v = ((uint32_t)w & 0xffu) * 0x01010101u;
            }
            D0v = (r0 + lane <= r_to) ? D0v : 7u; // rows past the end of the phase read as kind 7 = "end of block"
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
                    reg_row(bs0_tag, std::false_type{}, r, d0, base4);
                    advance();
                }
                if (kind == 7u) break;
                if (kind == 1)
                    reg_row(bs0_tag, std::true_type{}, r, d0, base4);
                else if (kind <= 3)
                {
                    // ===== predecessors from the LDS ring =====
                    const uint32_t bs = ((d0 >> 3) & 0x1ffu) << 2;
                    a1A = (2u * bs + (uint32_t)lane8) & kMask;
                    a1B = (a1A + 512u) & kMask;
                    ga  = (a1A + guard_off) & kMask;
                    const uint32_t a0A = (a1A - 4) & kMask; // dword whose high half is the cell of column c
                    const uint32_t a0B = (a1B - 4) & kMask;
                    const uint32_t b0  = ring_base + (((d0 >> 12) & 7u) * kWdSlotBytes);
                    const uint32_t sent16 = (uint32_t)kPkSentinel & 0xffffu;
                    const uint32_t dd0 = (d0 >> 21) & 7u;
                    // one predecessor's loads (one LDS round trip for all of a row's)
                    struct PredCells { uint32_t xA; uint2 qA; uint32_t xB; uint2 qB; };
                    auto load_pred = [&](uint32_t b) -> PredCells {
                        PredCells pc;
                        pc.xA = lds_load_u32(b + a0A);
                        pc.qA = lds_load_u64(b + a1A);
                        pc.xB = lds_load_u32(b + a0B);
                        pc.qB = lds_load_u64(b + a1B);
                        return pc;
                    };
                    auto rel0_in_slot = [&](uint32_t b) -> int32_t {
                        return (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b + kWdSlotBytes - 4) >> 16));
                    };
                    if (kind == 2)
                    {
                        const PredCells p0c = load_pred(b0);
                        rdA  = lds_load_u32(read_base + bs + lane4);
                        rdAn = lds_load_u32(read_base + bs + lane4 + 4);
                        rdB  = lds_load_u32(read_base + bs + 256 + lane4);
                        rdBn = lds_load_u32(read_base + bs + 256 + lane4 + 4);
                        int32_t cu = min_score + 2 * gap_score, rel0_val = min_score;
                        if constexpr (BS0)
                        {
                            rel0_val = max(min_score, rel0_in_slot(b0)) + gap_score;
                            cu       = rel0_val + gap_score;
                        }
                        uint32_t cA01, cA23, cB01, cB23, DA01, DA23, VA01, VA23, DB01, DB23, VB01, VB23;
                        costs(rdA, base4, cA01, cA23);
                        costs(rdB, base4, cB01, cB23);
                        from_pred(p0c.xA, p0c.qA.x, p0c.qA.y, cA01, cA23, DA01, DA23, VA01, VA23);
                        from_pred(p0c.xB, p0c.qB.x, p0c.qB.y, cB01, cB23, DB01, DB23, VB01, VB23);
                        // a chunk beyond the predecessor's band: pass A's never is (the band starts differ by <= 60 columns)
                        const bool outB = (p0c.qB.x & 0xffffu) == sent16;
                        const uint32_t sB01 = pk_max(DB01, VB01), sB23 = pk_max(DB23, VB23);
                        scan_row(pk_max(DA01, VA01), pk_max(DA23, VA23), outB ? MIN2 : sB01, outB ? MIN2 : sB23, cu);
                        // move = H == D ? 2 d + 1 : H == V ? 2 d : 1  ==  (2 d + 1) + [H != D] * (-1 + [H != V] * (1 - 2 d))
                        const uint32_t cD = pk_dup((int32_t)(2u * dd0 + 1u)), cV = pk_dup(1 - (int32_t)(2u * dd0));
                        const uint32_t mA01 = pk_mad_u16_vvs(nz(PA01, DA01), pk_mad_u16_vsv(nz(PA01, VA01), cV, NEG1), cD);
                        const uint32_t mA23 = pk_mad_u16_vvs(nz(PA23, DA23), pk_mad_u16_vsv(nz(PA23, VA23), cV, NEG1), cD);
                        const uint32_t mB01 = pk_mad_u16_vvs(nz(PB01, DB01), pk_mad_u16_vsv(nz(PB01, VB01), cV, NEG1), cD);
                        const uint32_t mB23 = pk_mad_u16_vvs(nz(PB23, DB23), pk_mad_u16_vsv(nz(PB23, VB23), cV, NEG1), cD);
                        store_row(bs0_tag, r, rel0_val, pack_moves(mA01, mA23), outB ? 0u : pack_moves(mB01, mB23));
                    }
                    else
                    {
                        const uint32_t cnt3 = d0 >> 30;          // 2, 3, or 0 = more than three
                        const int32_t cnt   = cnt3 == 2 ? 2 : 3; // predecessors in the descriptor
                        const uint32_t b1   = ring_base + (((d0 >> 15) & 7u) * kWdSlotBytes);
                        const uint32_t b2   = cnt > 2 ? ring_base + (((d0 >> 18) & 7u) * kWdSlotBytes) : b0;
                        const uint32_t dd1 = (d0 >> 24) & 7u, dd2 = (d0 >> 27) & 7u;
                        const PredCells p0c = load_pred(b0);
                        const PredCells p1c = load_pred(b1);
                        PredCells p2c       = p0c;
                        if (cnt > 2) p2c = load_pred(b2);
                        rdA  = lds_load_u32(read_base + bs + lane4);
                        rdAn = lds_load_u32(read_base + bs + lane4 + 4);
                        rdB  = lds_load_u32(read_base + bs + 256 + lane4);
                        rdBn = lds_load_u32(read_base + bs + 256 + lane4 + 4);
                        int32_t fe = min_score + gap_score;
                        if constexpr (BS0) // left boundary in band: carry-in from the predecessors' column-0 values (:293-326)
                        {
                            int32_t pen = max(min_score, rel0_in_slot(b0));
                            pen         = max(pen, rel0_in_slot(b1));
                            if (cnt > 2) pen = max(pen, rel0_in_slot(b2));
                            fe = pen + gap_score;
                        }
                        uint32_t cA01, cA23, cB01, cB23;
                        costs(rdA, base4, cA01, cA23);
                        costs(rdB, base4, cB01, cB23);
                        // one pass of the row: best diagonal / vertical candidate over the predecessors and, per cell, the move
                        // of the first slot that attains it (A = diagonal move, B = vertical move); returns "undecided here"
                        struct PassOut { uint32_t bD01, bD23, bV01, bV23, A01, A23, B01, B23; bool undecided; };
                        const uint32_t mD0 = pk_dup((int32_t)(2u * dd0 + 1u)), mV0 = pk_dup((int32_t)(2u * dd0));
                        const uint32_t E1v = pin_vgpr(pk_dup(2 * ((int32_t)dd1 - (int32_t)dd0)));
                        const uint32_t E2v = pin_vgpr(pk_dup(2 * ((int32_t)dd2 - (int32_t)dd1)));
                        auto one_pass = [&](auto outside_tag, uint32_t x0, uint2 q0, uint32_t x1, uint2 q1, uint32_t x2, uint2 q2,
                                            uint32_t c01, uint32_t c23) -> PassOut {
                            constexpr bool may_be_outside = decltype(outside_tag)::value;
                            PassOut o;
                            uint32_t D0a, D0b, V0a, V0b, D1a, D1b, V1a, V1b;
                            from_pred(x0, q0.x, q0.y, c01, c23, D0a, D0b, V0a, V0b);
                            from_pred(x1, q1.x, q1.y, c01, c23, D1a, D1b, V1a, V1b);
                            const bool out0 = may_be_outside && (q0.x & 0xffffu) == sent16, out1 = may_be_outside && (q1.x & 0xffffu) == sent16;
                            o.undecided     = out0 | out1;
                            D0a = out0 ? MIN2 : D0a; D0b = out0 ? MIN2 : D0b; V0a = out0 ? MIN2 : V0a; V0b = out0 ? MIN2 : V0b;
                            D1a = out1 ? MIN2 : D1a; D1b = out1 ? MIN2 : D1b; V1a = out1 ? MIN2 : V1a; V1b = out1 ? MIN2 : V1b;
                            o.bD01 = pk_max(D0a, D1a); o.bD23 = pk_max(D0b, D1b); o.bV01 = pk_max(V0a, V1a); o.bV23 = pk_max(V0b, V1b);
                            //   A = mD0 + n0 * (E1 + n1 * E2),  E1 = 2 (d1 - d0), E2 = 2 (d2 - d1),  n_k = [slot k misses the maximum]
                            if (cnt > 2)
                            {
                                uint32_t D2a, D2b, V2a, V2b;
                                from_pred(x2, q2.x, q2.y, c01, c23, D2a, D2b, V2a, V2b);
                                const bool out2 = may_be_outside && (q2.x & 0xffffu) == sent16;
                                o.undecided     = o.undecided | out2;
                                D2a = out2 ? MIN2 : D2a; D2b = out2 ? MIN2 : D2b; V2a = out2 ? MIN2 : V2a; V2b = out2 ? MIN2 : V2b;
                                o.bD01 = pk_max(o.bD01, D2a); o.bD23 = pk_max(o.bD23, D2b); o.bV01 = pk_max(o.bV01, V2a); o.bV23 = pk_max(o.bV23, V2b);
                                o.A01 = pk_mad_u16_vvs(nz(o.bD01, D0a), pk_mad_u16(nz(o.bD01, D1a), E2v, E1v), mD0);
                                o.A23 = pk_mad_u16_vvs(nz(o.bD23, D0b), pk_mad_u16(nz(o.bD23, D1b), E2v, E1v), mD0);
                                o.B01 = pk_mad_u16_vvs(nz(o.bV01, V0a), pk_mad_u16(nz(o.bV01, V1a), E2v, E1v), mV0);
                                o.B23 = pk_mad_u16_vvs(nz(o.bV23, V0b), pk_mad_u16(nz(o.bV23, V1b), E2v, E1v), mV0);
                            }
                            else
                            {
                                o.A01 = pk_mad_u16_vvs(nz(o.bD01, D0a), E1v, mD0); o.A23 = pk_mad_u16_vvs(nz(o.bD23, D0b), E1v, mD0);
                                o.B01 = pk_mad_u16_vvs(nz(o.bV01, V0a), E1v, mV0); o.B23 = pk_mad_u16_vvs(nz(o.bV23, V0b), E1v, mV0);
                            }
                            return o;
                        };
                        PassOut oA = one_pass(std::false_type{}, p0c.xA, p0c.qA, p1c.xA, p1c.qA, p2c.xA, p2c.qA, cA01, cA23);
                        PassOut oB = one_pass(std::true_type{}, p0c.xB, p0c.qB, p1c.xB, p1c.qB, p2c.xB, p2c.qB, cB01, cB23);
                        if (cnt3 == 0)
                        {
                            // predecessors 3..5 (rows from the side table, cells from the ring): they raise the maxima; where only
                            // they attain a maximum the first attaining slot is >= 3, whose distance the pass does not track -> move 0
                            const uint64_t xe     = wave_first64(xpred[r & 255]);
                            const int32_t cnt_all = (int32_t)((xe >> 13) & 63u);
                            uint32_t xDA01 = MIN2, xDA23 = MIN2, xVA01 = MIN2, xVA23 = MIN2;
                            uint32_t xDB01 = MIN2, xDB23 = MIN2, xVB01 = MIN2, xVB23 = MIN2;
                            int32_t pen_x = min_score;
                            for (int32_t kk = 3; kk < cnt_all; kk++)
                            {
                                const uint32_t bk   = ring_base + (((uint32_t)xpred_row(xe, kk) & (kWdSlots - 1)) * kWdSlotBytes);
                                const PredCells pkc = load_pred(bk);
                                uint32_t Da, Db, Va, Vb;
                                from_pred(pkc.xA, pkc.qA.x, pkc.qA.y, cA01, cA23, Da, Db, Va, Vb);
                                xDA01 = pk_max(xDA01, Da); xDA23 = pk_max(xDA23, Db); xVA01 = pk_max(xVA01, Va); xVA23 = pk_max(xVA23, Vb);
                                from_pred(pkc.xB, pkc.qB.x, pkc.qB.y, cB01, cB23, Da, Db, Va, Vb);
                                const bool outk = (pkc.qB.x & 0xffffu) == sent16;
                                oB.undecided    = oB.undecided | outk;
                                xDB01 = pk_max(xDB01, outk ? MIN2 : Da); xDB23 = pk_max(xDB23, outk ? MIN2 : Db);
                                xVB01 = pk_max(xVB01, outk ? MIN2 : Va); xVB23 = pk_max(xVB23, outk ? MIN2 : Vb);
                                if constexpr (BS0) pen_x = max(pen_x, rel0_in_slot(bk));
                            }
                            if constexpr (BS0) fe = max(fe - gap_score, pen_x) + gap_score;
                            auto fold = [&](PassOut& o, uint32_t xD01, uint32_t xD23, uint32_t xV01, uint32_t xV23) {
                                const uint32_t fD01 = pk_max(o.bD01, xD01), fD23 = pk_max(o.bD23, xD23), fV01 = pk_max(o.bV01, xV01), fV23 = pk_max(o.bV23, xV23);
                                // A *= [max of the first three == overall max]
                                o.A01 = pk_mad_u16(nz(o.bD01, fD01), pk_sub(0u, o.A01), o.A01); o.A23 = pk_mad_u16(nz(o.bD23, fD23), pk_sub(0u, o.A23), o.A23);
                                o.B01 = pk_mad_u16(nz(o.bV01, fV01), pk_sub(0u, o.B01), o.B01); o.B23 = pk_mad_u16(nz(o.bV23, fV23), pk_sub(0u, o.B23), o.B23);
                                o.bD01 = fD01; o.bD23 = fD23; o.bV01 = fV01; o.bV23 = fV23;
                            };
                            fold(oA, xDA01, xDA23, xVA01, xVA23);
                            fold(oB, xDB01, xDB23, xVB01, xVB23);
                        }
                        const int32_t rel0_val = BS0 ? fe : min_score;
                        scan_row(pk_max(oA.bD01, oA.bV01), pk_max(oA.bD23, oA.bV23), pk_max(oB.bD01, oB.bV01), pk_max(oB.bD23, oB.bV23), fe + gap_score);
                        // move = H == bestD ? A : H == bestV ? B : 1
                        auto move_of = [&](uint32_t H, uint32_t bD, uint32_t bV, uint32_t A, uint32_t B) -> uint32_t {
                            const uint32_t t1 = pk_mad_u16(nz(H, bV), pk_sub(ONE2, B), B);
                            return pk_mad_u16(nz(H, bD), pk_sub(t1, A), A);
                        };
                        const uint32_t mvA = pack_moves(move_of(PA01, oA.bD01, oA.bV01, oA.A01, oA.B01), move_of(PA23, oA.bD23, oA.bV23, oA.A23, oA.B23));
                        const uint32_t mvB = pack_moves(move_of(PB01, oB.bD01, oB.bV01, oB.A01, oB.B01), move_of(PB23, oB.bD23, oB.bV23, oB.A23, oB.B23));
                        store_row(bs0_tag, r, rel0_val, oA.undecided ? 0u : mvA, oB.undecided ? 0u : mvB);
                    }
                }
                else
                {
                    general_row(r, prev_rel0);
                    if constexpr (!BS0) prev_rel0 = min_score;
                }
                advance();
            }
        }
    };

    const int32_t bs0_end = min(first_moved - 1, graph_count); // last row whose band starts at column 0
    run_rows(std::true_type{}, 1, bs0_end);
    // from here on every row's left boundary is min_score by construction; the first such row sees the previous row's
    // real boundary through the general routine (band-start transition rows are kind 4)
    run_rows(std::false_type{}, bs0_end + 1, graph_count);
}

} // namespace gwhip
