// poa_forward_packed.h -- banded NW forward pass for the 256-column band with int16 scores, written for the
// execution profile of a lone wavefront on gfx950 (tools/microbench.hip: ~4.5 cycles per issued instruction of
// any kind, ~25 per taken branch, ~55-70 per LDS round trip). What the pass computes is cudapoa_nw_banded.cuh:
// 269-408 (restated in oracle/poa_nw.inc); how:
//
//   * two score cells per 32-bit register (v_pk_add_i16 / v_pk_max_i16): a lane's four cells are two registers,
//     which are also exactly the 8 bytes it stores, so nothing is packed or unpacked around memory;
//   * every row is classified once per read, by all lanes in parallel (classify_rows): class 0 rows have the
//     previous row as only predecessor and an unmoved band -- the previous row is in registers and one DPP lane
//     shift aligns the diagonal; class 1 rows have one predecessor up to 7 rows back (or a moved band) and read it
//     from an LDS ring; class 2 rows have 2-6 such predecessors (rows 3..5 of them in the LDS side table that
//     build_rowinfo fills); class 3 rows (no predecessor, > 6 predecessors, far predecessors, band-start
//     transition) take the general 32-bit routine against the HBM matrix. The row loop itself only tests two bits
//     of the row-table word;
//   * the LDS ring holds 8 rows of 512 absolute column slots (cell of column x at slot (x - 1) & 511), so a
//     reader addresses a predecessor row by column alone and never needs that row's band start; each row also
//     stores sentinel cells behind its band end, which is how a reader recognises a 4-cell chunk that lies
//     outside the predecessor's band (the reference's chunk predicate, cudapoa_nw_banded.cuh:139-156) without
//     any band arithmetic, and its left-boundary value at the slot of column band_start;
//   * the horizontal max-plus recurrence is a prefix maximum of u[t] = v[t] - t*gap (see poa_device.h).
//
// Preconditions (checked by the caller, otherwise banded_forward_1pass runs): band_width == 256,
// max_column >= band_width (no chunk reaches past the read), and score parameters small enough that
// v - t*gap stays inside int16 (|gap| <= 30). Under the reference's own precondition that no stored score wraps,
// every value this routine forms fits int16, so packed 16-bit arithmetic is exact.
#pragma once

namespace gwhip
{

typedef short pk_i16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_i16, a) + __builtin_bit_cast(pk_i16, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_i16, a) - __builtin_bit_cast(pk_i16, b));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_i16, a), __builtin_bit_cast(pk_i16, b)));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
}
__device__ __forceinline__ uint32_t pk_mad_u(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(pk_u16, a) * __builtin_bit_cast(pk_u16, b) + __builtin_bit_cast(pk_u16, c));
}
__device__ __forceinline__ uint32_t pk_dup(int32_t v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
__device__ __forceinline__ uint32_t pk_make(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int32_t pk_lo(uint32_t v) { return (int32_t)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int32_t pk_hi(uint32_t v) { return (int32_t)v >> 16; }

constexpr int kPkSlots     = 8;    // ring rows
constexpr int kPkSlotBytes = 1024; // 512 column slots x int16
constexpr int kPkMaxDist   = kPkSlots - 1;
constexpr int kPkGuardCols = 60;   // a reader's band may start at most this far right of a ring predecessor's
constexpr int kPkSentinel  = -32768;
constexpr int kClassShift  = 60;   // row class in bits 60..61 of the packed row-table word

// LDS byte address of a pointer into the dynamic shared segment
__device__ __forceinline__ uint32_t lds_addr(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(addr);
}
__device__ __forceinline__ uint2 lds_load_u64(uint32_t addr)
{
    const u32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(addr);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_store_u64(uint32_t addr, uint32_t lo, uint32_t hi)
{
    u32x2 v;
    v.x = lo; v.y = hi;
    *reinterpret_cast<__attribute__((address_space(3))) u32x2*>(addr) = v;
}
// 8-byte LDS store by lanes 0..16 only (the guard cells + left-boundary quad of a ring row), no branch
__device__ __forceinline__ void lds_store_u64_lanes17(uint32_t addr, uint32_t lo, uint32_t hi)
{
    const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
    asm volatile("s_mov_b64 exec, 0x1ffff\n\tds_write_b64 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(addr), "v"(v) : "memory");
}

// ------------------------------------------------------------------------------------------------
// Row classes for banded_forward_packed, all lanes in parallel. Needs the band starts in the table already.
//   0: one predecessor, the previous row, band not moved                       (previous row from registers)
//   1: one predecessor, 1..7 rows back, band starts compatible with the ring     (predecessor from the LDS ring)
//   2: 2..6 predecessors, each 1..7 rows back, band starts compatible            (predecessors from the LDS ring)
//   3: everything else                                                           (general routine, HBM matrix)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool xpred_hit(uint64_t e, int32_t row, int32_t cnt)
{
    return (int32_t)(e & 0xfffu) == row && ((e >> 12) & 1u) != 0 && (int32_t)((e >> 13) & 63u) == cnt;
}
__device__ __forceinline__ int32_t xpred_row(uint64_t e, int32_t k) { return (int32_t)((e >> (20 + 12 * (k - 3))) & 0xfffu); } // k = 3..5

__device__ __forceinline__ void classify_rows(RowInfo<true>* rowinfo, int32_t graph_count, int lane, const uint64_t* xpred,
                                              int32_t dbg = 0, uint64_t* prof_acc = nullptr)
{
    int32_t prof_count = 0; // profiling (GWHIP_DEBUG bits 2-3 = class to count: 1, 2 or 3)
    for (int32_t r = 1 + lane; r <= graph_count; r += kWave)
    {
        RowInfo<true> ri = rowinfo[r];
        const int32_t cnt = ri.cnt(), bs = ri.bs();
        uint64_t cls = 3;
        // profiling (GWHIP_DEBUG bits 12, 14, 19 = reason selector): rows that are class 3 because of
        // 1 no predecessor, 2 more than three, 3 a predecessor more than 7 rows back, 4 band guard, 5 band-start transition
        const int32_t rsel = ((dbg >> 12) & 1) | (((dbg >> 14) & 1) << 1) | (((dbg >> 19) & 1) << 2);
        // 4..6 predecessors: rows 3..5 from the side table build_rowinfo left (a row that lost its slot stays class 3)
        const uint64_t xe  = (cnt > 3 && cnt <= 6 && xpred != nullptr) ? xpred[r & 255] : 0ull;
        const bool many_ok = cnt > 3 && cnt <= 6 && xpred_hit(xe, r, cnt) && !(dbg & (1 << 30));
        int32_t reason     = cnt == 0 ? 1 : ((cnt > 3 && !many_ok) ? 2 : 0);
        if (cnt >= 1 && (cnt <= 3 || many_ok))
        {
            bool ok = true, first_is_prev_unmoved = false;
            for (int32_t k = 0; k < cnt; k++)
            {
                const int32_t p   = k < 3 ? ri.pred(k) : xpred_row(xe, k);
                const int32_t d   = r - p;
                const int32_t pbs = rowinfo[p].bs(); // row 0 holds band start 0
                ok                = ok && d >= 1 && d <= kPkMaxDist && (bs - pbs) <= kPkGuardCols && (bs == 0 || pbs > 0);
                if (k == 0) first_is_prev_unmoved = (d == 1 && pbs == bs);
                if (reason == 0 && !(d >= 1 && d <= kPkMaxDist)) reason = 3;
                if (reason == 0 && !((bs - pbs) <= kPkGuardCols)) reason = 4;
                if (reason == 0 && !(bs == 0 || pbs > 0)) reason = 5;
            }
            if (ok) cls = cnt > 1 ? 2 : (first_is_prev_unmoved ? 0 : 1);
        }
        if (rsel && reason == rsel) prof_count += 1000;
        // ablations (GWHIP_DEBUG): demote classes to check them against each other
        if ((dbg & 1024) && cls == 0) cls = 1;
        if ((dbg & 2048) && cls == 0) cls = 3;
        if ((dbg & 512) && (cls == 1 || cls == 2)) cls = 3;
        if ((dbg & 32768) && cls == 1) cls = 2;
        ri.w       = (ri.w & ~(3ull << kClassShift)) | (cls << kClassShift);
        rowinfo[r] = ri;
        if ((dbg & 12) && cls == (uint64_t)((dbg >> 2) & 3)) prof_count++;
    }
    if (((dbg & 12) || (dbg & ((1 << 12) | (1 << 14) | (1 << 19)))) && prof_acc)
    {
        for (int off = 32; off > 0; off >>= 1) prof_count += __shfl_xor(prof_count, off);
        *prof_acc += (uint64_t)prof_count;
    }
}

// ------------------------------------------------------------------------------------------------
// Trace codes. Besides the score row, classes 0, 1 and 2 store one byte per cell that names the move the reference's
// traceback (cudapoa_nw_banded.cuh:428-549: diagonal through predecessor 0..n-1, then vertical through
// predecessor 0..n-1, then horizontal, first equality wins) takes from that cell:
//   0 undecided here -> the traceback recomputes the step from the score matrix
//   1 horizontal     2 + k diagonal through predecessor slot k     5 + k vertical through predecessor slot k
// A code is only written where the forward pass saw exactly the operands the traceback's get_score() would see:
// the first cell of the band (its horizontal operand is the carry-in, not a stored cell), chunks that lie outside
// some predecessor's band, cells whose maximum is only attained by predecessor slot 3 or later (a code names slots
// 0..2) and all class 3 rows stay 0. Equality of H with a candidate is tested on the stored 16-bit values, which is
// the comparison the traceback makes.
// ------------------------------------------------------------------------------------------------
constexpr int kCodeHoriz = 1, kCodeDiag = 2, kCodeVert = 5;

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// keep a loop-invariant in a VGPR (the kernel is SGPR-bound: a uniform constant would otherwise live in an SGPR
// and be spilled / reloaded with v_readlane inside the row loop)
__device__ __forceinline__ uint32_t pin_vgpr(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}
// lane-0 2-byte global store without a divergent branch (wave-uniform caller, all lanes active)
__device__ __forceinline__ void global_store_u16_lane0(void* p, uint32_t v)
{
    const uint32_t zero = 0;
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_short %0, %1, %2\n\ts_mov_b64 exec, -1" ::"v"(zero), "v"(v), "s"(p) : "memory");
}

// ------------------------------------------------------------------------------------------------
// The forward pass. `ring` is kPkSlots * kPkSlotBytes of LDS; `scores` the HBM score matrix and `codes` the HBM
// trace-code matrix (row stride 264 elements, our layout of poa_device.h); lds_read the LDS copy of the read.
// ------------------------------------------------------------------------------------------------
template <typename IdT>
__device__ __forceinline__ void banded_forward_packed(const GraphView<IdT>& g, const RowInfo<true>* rowinfo,
                                                      int32_t graph_count, const uint8_t* lds_read, int16_t* scores,
                                                      uint8_t* codes, uint8_t* ring, const uint64_t* xpred, int32_t max_column,
                                                      int32_t gap_score, int32_t mismatch_score, int32_t match_score,
                                                      int32_t dbg, uint64_t* prof_acc)
{
    constexpr int32_t band_width = 256;
    constexpr int32_t stride     = band_width + kRightPad;
    const int lane               = threadIdx.x & (kWave - 1);
    const int32_t lane4 = lane * 4, lane8 = lane * 8;
    const int32_t min_score = Limits<int16_t>::min / 2;
    const uint32_t MIN2   = pin_vgpr(pk_dup(min_score));
    const uint32_t SENT2  = pin_vgpr(pk_dup(kPkSentinel));
    const uint32_t GAP2   = pin_vgpr(pk_dup(gap_score));
    const uint32_t MAT2   = pin_vgpr(pk_dup(match_score));
    const uint32_t DIF2   = pin_vgpr(pk_dup(mismatch_score - match_score));
    const uint32_t ONE2   = pin_vgpr(0x00010001u);
    const uint32_t TWO2   = pin_vgpr(0x00020002u);
    const uint32_t THREE2 = pin_vgpr(0x00030003u);
    const uint32_t FIVE2  = pin_vgpr(0x00050005u);
    const uint32_t NEG4   = pin_vgpr(0xfffcfffcu);
    // t * gap for the lane's cells t = 4*lane + k
    const uint32_t K01 = pk_make((lane4 + 0) * gap_score, (lane4 + 1) * gap_score);
    const uint32_t K23 = pk_make((lane4 + 2) * gap_score, (lane4 + 3) * gap_score);
    const uint32_t ring_base = lds_addr(ring);
    // guard store: lanes 0..15 write sentinel cells for columns band_end + 1 .. + 64, lane 16 the quad that ends in
    // the left-boundary slot (column band_start); byte offsets relative to the lane's own cell offset
    const uint32_t guard_off  = lane < 16 ? 512u : (uint32_t)-136;
    const bool is_lane16      = lane == 16;
    const uint32_t code_keep  = lane == 0 ? 0xffffff00u : 0xffffffffu; // the band's first cell keeps code 0
    const uint32_t out_off    = (uint32_t)lane8 + 2u * (1 + kRelShift); // byte offset of the lane's quad in an HBM score row
    const uint32_t code_off   = (uint32_t)lane4 + (1 + kRelShift);      // ... of its four codes in a code row

    // state carried from row to row
    uint32_t P01 = pk_make((lane4 + 1) * gap_score, (lane4 + 2) * gap_score); // row 0: H[0][x] = x * gap
    uint32_t P23 = pk_make((lane4 + 3) * gap_score, (lane4 + 4) * gap_score);
    int32_t prev_bs = 0, prev_rel0 = 0; // band start and left-boundary value of the row in P
    int32_t slot    = 0;                // ring slot of that row
    // per-lane values that only change when the band moves: read characters of columns c+1..c+4, ring byte offset
    // of the lane's quad and of its guard quad
    uint32_t rd4 = *reinterpret_cast<const uint32_t*>(lds_read + lane4);
    uint32_t a1  = (uint32_t)lane8;
    uint32_t ga  = (a1 + guard_off) & (kPkSlotBytes - 1);
    uint8_t* row_out  = reinterpret_cast<uint8_t*>(scores); // HBM score row of the row in P
    uint8_t* code_out = codes;
    bool hbm_dirty    = true;
    uint64_t prof     = 0;

    auto ring_write = [&](int32_t s, int32_t rel0) {
        const uint32_t sbase  = ring_base + (uint32_t)s * kPkSlotBytes;
        const uint32_t rel0pk = ((uint32_t)kPkSentinel & 0xffffu) | ((uint32_t)rel0 << 16);
        lds_store_u64(sbase + a1, P01, P23);
        lds_store_u64_lanes17(sbase + ga, SENT2, is_lane16 ? rel0pk : SENT2);
    };
    ring_write(0, 0); // row 0 into slot 0

    // horizontal max-plus scan of the row's candidates; leaves the finished row in P01/P23
    auto scan_row = [&](uint32_t s01, uint32_t s23, int32_t fe) {
        const uint32_t u01 = pk_sub(s01, K01), u23 = pk_sub(s23, K23);
        // in-lane prefix maxima: pm01 = (u0, max(u0,u1)), pm23 = (u2, max(u2,u3))
        const uint32_t pm01 = pk_max(u01, (u01 << 16) | 0x8000u);
        const uint32_t pm23 = pk_max(u23, (u23 << 16) | 0x8000u);
        const int32_t m3    = (int32_t)pk_max(pm01, pm23) >> 16; // max(u0..u3)
        const int32_t incl  = wave_inclusive_max(m3);
        const int32_t excl  = max(wave_shr1(incl, INT32_MIN), fe + gap_score); // carry-in is element t = -1
        const uint32_t ex2  = __builtin_amdgcn_perm((uint32_t)excl, (uint32_t)excl, 0x01000100u);
        const uint32_t m1b  = __builtin_amdgcn_perm(pm01, pm01, 0x03020302u); // max(u0,u1) in both halves
        P01 = pk_add(pk_max(pm01, ex2), K01);
        P23 = pk_add(pk_max(pk_max(pm23, m1b), ex2), K23);
    };
    // the finished row goes to HBM and the ring
    auto store_row = [&](int32_t bs, int32_t rel0_val) {
        row_out += stride * 2;
        slot = (slot + 1) & (kPkSlots - 1);
        *reinterpret_cast<uint2*>(row_out + out_off) = make_uint2(P01, P23);
        ring_write(slot, rel0_val);
        if (bs == 0) // only rows whose band starts at column 0 have a real left-boundary value in HBM
            global_store_u16_lane0(row_out + 2 * kRelShift, (uint32_t)rel0_val);
        hbm_dirty = true;
        prev_bs   = bs;
        prev_rel0 = rel0_val;
    };
    auto store_codes = [&](uint32_t code01, uint32_t code23, bool undecided) {
        code_out += stride;
        uint32_t c4 = __builtin_amdgcn_perm(code23, code01, 0x06040200u) & code_keep; // low byte of each half
        c4          = undecided ? 0u : c4;
        *reinterpret_cast<uint32_t*>(code_out + code_off) = c4;
    };
    // 0 where the halves are equal, 1 where they differ
    auto nz = [&](uint32_t a, uint32_t b) -> uint32_t { return pk_min_u16(pk_sub(a, b), ONE2); };
    // match / mismatch cost pairs of this row's base against the lane's four read characters
    auto costs = [&](uint32_t base, uint32_t& c01, uint32_t& c23) {
        const uint32_t x   = rd4 ^ (base * 0x01010101u);
        const uint32_t x01 = __builtin_amdgcn_perm(0u, x, 0x0c010c00u); // (byte0, byte1) zero-extended to halves
        const uint32_t x23 = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
        c01 = pk_mad_u16(pk_min_u16(x01, ONE2), DIF2, MAT2);
        c23 = pk_mad_u16(pk_min_u16(x23, ONE2), DIF2, MAT2);
    };
    // diagonal / vertical candidates of the four cells from one predecessor row: q01/q23 = its cells of columns
    // c+1..c+4, s0x = its cell of column c in the HIGH half
    auto from_pred = [&](uint32_t s0x, uint32_t q01, uint32_t q23, uint32_t c01, uint32_t c23, uint32_t& D01, uint32_t& D23,
                         uint32_t& V01, uint32_t& V23) {
        D01 = pk_add(__builtin_amdgcn_alignbit(q01, s0x, 16), c01); // from (col c, col c+1)
        D23 = pk_add(__builtin_amdgcn_alignbit(q23, q01, 16), c23); // from (col c+2, col c+3)
        V01 = pk_add(q01, GAP2);
        V23 = pk_add(q23, GAP2);
    };
    auto class_of = [&](const RowInfo<true>& w) -> uint32_t { return (uint32_t)(w.w >> kClassShift) & 3u; };

    // profiling (GWHIP_DEBUG bits 28-29): 1 cycles in class 3 rows, 2 their number x 1000 -- tested inside the (rare)
    // class 3 branch only; 3 cycles in class 2 rows needs a build with -DGWHIP_PROFILE_CLASS2 (a test per class 2 row
    // costs ~1 % of the kernel)
    const int32_t fsel = prof_acc ? (dbg >> 28) & 3 : 0;
    uint64_t facc      = 0;
    int32_t r        = 1;
    RowInfo<true> ri = uniform_row(rowinfo[1]);
    uint32_t cls     = class_of(ri);
    while (r <= graph_count)
    {
        // ========== streak of single-predecessor rows: class 0 (registers) and class 1 (LDS ring) ==========
        while (cls <= 1)
        {
            const RowInfo<true> nxt = rowinfo[min(r + 1, graph_count)]; // consumed after the arithmetic
            const int32_t bs        = ri.bs();
            uint32_t s0x, q01, q23;
            bool outside = false;
            int32_t fe   = min_score + gap_score;
            if (cls == 1)
            {
                a1  = (uint32_t)(2 * bs + lane8) & (kPkSlotBytes - 1);
                ga  = (a1 + guard_off) & (kPkSlotBytes - 1);
                rd4 = *reinterpret_cast<const uint32_t*>(lds_read + bs + lane4);
                const int32_t d     = r - ri.pred(0);
                const uint32_t pb   = ring_base + (uint32_t)((slot + 1 - d) & (kPkSlots - 1)) * kPkSlotBytes;
                s0x                 = lds_load_u32(pb + ((a1 - 4) & (kPkSlotBytes - 1)));
                const uint2 q       = lds_load_u64(pb + a1);
                q01 = q.x; q23 = q.y;
                outside = (q01 & 0xffffu) == ((uint32_t)kPkSentinel & 0xffffu); // chunk beyond the predecessor's band
                if (bs == 0)
                    fe = max(min_score, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(pb + kPkSlotBytes - 4) >> 16))) + gap_score;
            }
            else
            {
                s0x = (uint32_t)wave_shr1((int32_t)P23, (int32_t)((uint32_t)prev_rel0 << 16));
                q01 = P01; q23 = P23;
                if (bs == 0) fe = max(min_score, prev_rel0) + gap_score;
            }
            const int32_t rel0_val = bs == 0 ? fe : min_score;
            uint32_t c01, c23, D01, D23, V01, V23;
            costs((uint32_t)ri.base(), c01, c23);
            from_pred(s0x, q01, q23, c01, c23, D01, D23, V01, V23);
            const uint32_t s01 = pk_max(D01, V01), s23 = pk_max(D23, V23);
            scan_row(outside ? MIN2 : s01, outside ? MIN2 : s23, fe);
            // code = H == D ? diag : H == V ? vert : horiz  ==  2 + [H != D] * (3 - 4 * [H != V])
            const uint32_t code01 = pk_mad_u16(nz(P01, D01), pk_mad_u16(nz(P01, V01), NEG4, THREE2), TWO2);
            const uint32_t code23 = pk_mad_u16(nz(P23, D23), pk_mad_u16(nz(P23, V23), NEG4, THREE2), TWO2);
            const RowInfo<true> ri_n = uniform_row(nxt); // before this row's LDS stores: no wait behind them
            store_row(bs, rel0_val);
            store_codes(code01, code23, outside);
            r++;
            ri  = ri_n;
            cls = r <= graph_count ? class_of(ri) : 7u;
        }
        if (r > graph_count) break;

#ifdef GWHIP_PROFILE_CLASS2
        const uint64_t t_row2 = (fsel == 3 && cls == 2) ? clock64() : 0;
        const uint32_t cls_now = cls;
#endif
        const RowInfo<true> nxt = rowinfo[min(r + 1, graph_count)];
        const int32_t bs        = ri.bs();
        const uint32_t base     = (uint32_t)ri.base();
        a1  = (uint32_t)(2 * bs + lane8) & (kPkSlotBytes - 1);
        ga  = (a1 + guard_off) & (kPkSlotBytes - 1);
        rd4 = *reinterpret_cast<const uint32_t*>(lds_read + bs + lane4);
        if (cls == 2)
        {
            // ================= every predecessor (2..3, at most 7 rows back) from the LDS ring =================
            const int32_t cnt_all = ri.cnt();         // 2..6; predecessors 3..5 are in the side table
            const int32_t cnt     = min(cnt_all, 3);
            const int32_t my_slot = (slot + 1) & (kPkSlots - 1);
            const uint32_t a0     = (a1 - 4) & (kPkSlotBytes - 1); // dword whose high half is the cell of column c
            auto slot_base = [&](int32_t k) -> uint32_t {
                const int32_t d = r - ri.pred(k);
                return ring_base + (uint32_t)((my_slot - d) & (kPkSlots - 1)) * kPkSlotBytes;
            };
            const uint32_t b0 = slot_base(0);
            const uint32_t b1 = cnt > 1 ? slot_base(1) : b0;
            const uint32_t b2 = cnt > 2 ? slot_base(2) : b0;
            // all loads first (one LDS round trip), then the arithmetic
            const uint32_t x0 = lds_load_u32(b0 + a0);
            const uint2 q0    = lds_load_u64(b0 + a1);
            uint32_t x1 = 0, x2 = 0;
            uint2 q1 = make_uint2(0, 0), q2 = make_uint2(0, 0);
            if (cnt > 1)
            {
                x1 = lds_load_u32(b1 + a0);
                q1 = lds_load_u64(b1 + a1);
            }
            if (cnt > 2)
            {
                x2 = lds_load_u32(b2 + a0);
                q2 = lds_load_u64(b2 + a1);
            }
            int32_t fe = min_score + gap_score;
            if (bs == 0) // left boundary in band: carry-in from the predecessors' column-0 values (:293-326)
            {
                int32_t pen = max(min_score, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b0 + kPkSlotBytes - 4) >> 16)));
                if (cnt > 1) pen = max(pen, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b1 + kPkSlotBytes - 4) >> 16)));
                if (cnt > 2) pen = max(pen, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(b2 + kPkSlotBytes - 4) >> 16)));
                fe = pen + gap_score;
            }
            int32_t rel0_val = bs == 0 ? fe : min_score;
            const uint32_t sent16  = (uint32_t)kPkSentinel & 0xffffu;
            uint32_t c01, c23;
            costs(base, c01, c23);
            // best diagonal / vertical candidate over the predecessors and the first slot that attains it
            uint32_t D0a, D0b, V0a, V0b;
            from_pred(x0, q0.x, q0.y, c01, c23, D0a, D0b, V0a, V0b);
            bool undecided = (q0.x & 0xffffu) == sent16; // chunk beyond that predecessor's band (:139-156)
            D0a = undecided ? MIN2 : D0a; D0b = undecided ? MIN2 : D0b;
            V0a = undecided ? MIN2 : V0a; V0b = undecided ? MIN2 : V0b;
            uint32_t bD01 = D0a, bD23 = D0b, bV01 = V0a, bV23 = V0b;
            uint32_t kD01 = 0, kD23 = 0, kV01 = 0, kV23 = 0;
            if (cnt > 1)
            {
                uint32_t D1a, D1b, V1a, V1b;
                from_pred(x1, q1.x, q1.y, c01, c23, D1a, D1b, V1a, V1b);
                const bool out1 = (q1.x & 0xffffu) == sent16;
                undecided       = undecided | out1;
                D1a = out1 ? MIN2 : D1a; D1b = out1 ? MIN2 : D1b;
                V1a = out1 ? MIN2 : V1a; V1b = out1 ? MIN2 : V1b;
                bD01 = pk_max(bD01, D1a); bD23 = pk_max(bD23, D1b);
                bV01 = pk_max(bV01, V1a); bV23 = pk_max(bV23, V1b);
                if (cnt > 2)
                {
                    uint32_t D2a, D2b, V2a, V2b;
                    from_pred(x2, q2.x, q2.y, c01, c23, D2a, D2b, V2a, V2b);
                    const bool out2 = (q2.x & 0xffffu) == sent16;
                    undecided       = undecided | out2;
                    D2a = out2 ? MIN2 : D2a; D2b = out2 ? MIN2 : D2b;
                    V2a = out2 ? MIN2 : V2a; V2b = out2 ? MIN2 : V2b;
                    bD01 = pk_max(bD01, D2a); bD23 = pk_max(bD23, D2b);
                    bV01 = pk_max(bV01, V2a); bV23 = pk_max(bV23, V2b);
                    // first slot attaining the maximum: n0 * (1 + n1) with n_k = [slot k misses it]
                    const uint32_t n0a = nz(bD01, D0a), n0b = nz(bD23, D0b), m0a = nz(bV01, V0a), m0b = nz(bV23, V0b);
                    kD01 = pk_mad_u16(n0a, nz(bD01, D1a), n0a); kD23 = pk_mad_u16(n0b, nz(bD23, D1b), n0b);
                    kV01 = pk_mad_u16(m0a, nz(bV01, V1a), m0a); kV23 = pk_mad_u16(m0b, nz(bV23, V1b), m0b);
                }
                else
                {
                    kD01 = nz(bD01, D0a); kD23 = nz(bD23, D0b);
                    kV01 = nz(bV01, V0a); kV23 = nz(bV23, V0b);
                }
            }
            // move codes if the best candidate is attained by one of the first three predecessors: 2 + kD / 5 + kV
            uint32_t A01 = pk_add(kD01, TWO2), A23 = pk_add(kD23, TWO2), B01 = pk_add(kV01, FIVE2), B23 = pk_add(kV23, FIVE2);
            if (cnt_all > 3)
            {
                // predecessors 3..5 (rows from the side table, cells from the ring): they raise the maxima; where only
                // they attain a maximum the first attaining slot is >= 3, which a code cannot name -> that code is 0
                const uint64_t xe = wave_first64(xpred[r & 255]);
                uint32_t xD01 = MIN2, xD23 = MIN2, xV01 = MIN2, xV23 = MIN2;
                int32_t pen_x = min_score;
                for (int32_t k = 3; k < cnt_all; k++)
                {
                    const int32_t d   = r - xpred_row(xe, k);
                    const uint32_t bk = ring_base + (uint32_t)((my_slot - d) & (kPkSlots - 1)) * kPkSlotBytes;
                    const uint32_t xk = lds_load_u32(bk + a0);
                    const uint2 qk    = lds_load_u64(bk + a1);
                    uint32_t Da, Db, Va, Vb;
                    from_pred(xk, qk.x, qk.y, c01, c23, Da, Db, Va, Vb);
                    const bool outk = (qk.x & 0xffffu) == sent16;
                    undecided       = undecided | outk;
                    xD01 = pk_max(xD01, outk ? MIN2 : Da); xD23 = pk_max(xD23, outk ? MIN2 : Db);
                    xV01 = pk_max(xV01, outk ? MIN2 : Va); xV23 = pk_max(xV23, outk ? MIN2 : Vb);
                    if (bs == 0) pen_x = max(pen_x, (int32_t)(int16_t)wave_first((int32_t)(lds_load_u32(bk + kPkSlotBytes - 4) >> 16)));
                }
                if (bs == 0)
                {
                    fe       = max(fe - gap_score, pen_x) + gap_score;
                    rel0_val = fe;
                }
                const uint32_t fD01 = pk_max(bD01, xD01), fD23 = pk_max(bD23, xD23), fV01 = pk_max(bV01, xV01), fV23 = pk_max(bV23, xV23);
                // A *= [max of the first three == overall max]
                A01 = pk_mad_u16(nz(bD01, fD01), pk_sub(0u, A01), A01); A23 = pk_mad_u16(nz(bD23, fD23), pk_sub(0u, A23), A23);
                B01 = pk_mad_u16(nz(bV01, fV01), pk_sub(0u, B01), B01); B23 = pk_mad_u16(nz(bV23, fV23), pk_sub(0u, B23), B23);
                bD01 = fD01; bD23 = fD23; bV01 = fV01; bV23 = fV23;
            }
            scan_row(pk_max(bD01, bV01), pk_max(bD23, bV23), fe);
            // code = H == bestD ? A : H == bestV ? B : 1
            auto code_of = [&](uint32_t H, uint32_t bD, uint32_t bV, uint32_t A, uint32_t B) -> uint32_t {
                const uint32_t t1 = pk_mad_u16(nz(H, bV), pk_sub(ONE2, B), B);
                return pk_mad_u16(nz(H, bD), pk_sub(t1, A), A);
            };
            const uint32_t code01 = code_of(P01, bD01, bV01, A01, B01);
            const uint32_t code23 = code_of(P23, bD23, bV23, A23, B23);
            const RowInfo<true> ri_n = uniform_row(nxt);
            store_row(bs, rel0_val);
            store_codes(code01, code23, undecided);
            r++;
            ri = ri_n;
        }
        else
        {
            // ===== general row: 32-bit arithmetic, previous row from registers, any other row from the HBM matrix =====
            const uint64_t t_row3 = fsel == 1 ? clock64() : 0;
            if (fsel == 2) facc += 1000;
            const int32_t pred_count = ri.cnt();
            const int32_t c          = bs + lane4;
            const int32_t cp0 = ((rd4 & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp1 = (((rd4 >> 8) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp2 = (((rd4 >> 16) & 0xff) == base) ? match_score : mismatch_score;
            const int32_t cp3 = ((rd4 >> 24) == base) ? match_score : mismatch_score;
            const int32_t R0 = pk_lo(P01), R1 = pk_hi(P01), R2 = pk_lo(P23), R3 = pk_hi(P23);
            auto from_regs = [&](int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
                const int32_t q    = (bs - prev_bs) >> 2;
                const int32_t pend = min(prev_bs + band_width - kCellsPerLane, max_column);
                const int src      = lane + q;
                // the shuffle must run with every lane active (a lane that is masked off does not supply its value)
                const int32_t from_left = __shfl(R3, src - 1);
                const int32_t S0        = (q == 0 && lane == 0) ? prev_rel0 : from_left;
                const int32_t S1 = __shfl(R0, src), S2 = __shfl(R1, src), S3 = __shfl(R2, src), S4 = __shfl(R3, src);
                const bool valid = c <= pend;
                t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
                t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
                t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
                t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
            };
            auto from_hbm = [&](int32_t prow, int32_t& t0, int32_t& t1, int32_t& t2, int32_t& t3) {
                const int32_t pbs  = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                const int32_t pend = min(pbs + band_width - kCellsPerLane, max_column);
                const bool valid   = !(c > pend || c < pbs);
                if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
                int32_t S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0;
                if (valid)
                {
                    const int16_t* rowp = scores + (int64_t)prow * stride + (c - pbs) + kRelShift;
                    S0 = rowp[0];
                    const Quad<int16_t> qd = *reinterpret_cast<const Quad<int16_t>*>(rowp + 1);
                    S1 = qd.v[0]; S2 = qd.v[1]; S3 = qd.v[2]; S4 = qd.v[3];
                }
                if (pbs > 0 && c == pbs) S0 = min_score; // relative-0 slot of a row whose band starts past column 0
                t0 = valid ? max(S0 + cp0, S1 + gap_score) : min_score;
                t1 = valid ? max(S1 + cp1, S2 + gap_score) : min_score;
                t2 = valid ? max(S2 + cp2, S3 + gap_score) : min_score;
                t3 = valid ? max(S3 + cp3, S4 + gap_score) : min_score;
            };
            auto rel0_of = [&](int32_t prow) -> int32_t {
                if (prow == r - 1) return prev_rel0;
                const int32_t pbs = prow == 0 ? 0 : uniform_row(rowinfo[prow]).bs();
                if (pbs > 0) return min_score;
                if (hbm_dirty) { wave_sync(); hbm_dirty = false; }
                return wave_first((int32_t)scores[(int64_t)prow * stride + kRelShift]);
            };
            const int32_t node_id = (pred_count > 3) ? (int32_t)g.sorted_poa[r - 1] : 0;
            auto pred_row = [&](int32_t p) -> int32_t {
                if (pred_count == 0) return 0;
                if (p < 3) return ri.pred(p);
                return wave_first((int32_t)g.node_id_to_pos[g.incoming_edges[(int64_t)node_id * kEdges + p]] + 1);
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
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            const int32_t np = max(pred_count, 1);
            for (int32_t p = 0; p < np; p++)
            {
                const int32_t prow = pred_row(p);
                int32_t t0, t1, t2, t3;
                if (prow == r - 1 && !(dbg & 8192)) from_regs(t0, t1, t2, t3);
                else from_hbm(prow, t0, t1, t2, t3);
                if (p == 0) { s0 = t0; s1 = t1; s2 = t2; s3 = t3; }
                else { s0 = max(s0, t0); s1 = max(s1, t1); s2 = max(s2, t2); s3 = max(s3, t3); }
            }
            scan_row(pk_make(s0, s1), pk_make(s2, s3), fe);
            const RowInfo<true> ri_n = uniform_row(nxt);
            store_row(bs, rel0_val);
            store_codes(0, 0, true);
            r++;
            ri = ri_n;
            if (fsel == 1) facc += clock64() - t_row3;
        }
        cls = r <= graph_count ? class_of(ri) : 7u;
#ifdef GWHIP_PROFILE_CLASS2
        if (fsel == 3 && cls_now == 2) facc += clock64() - t_row2;
#endif
    }
    if (fsel && lane == 0) *prof_acc += facc;
}

} // namespace gwhip
