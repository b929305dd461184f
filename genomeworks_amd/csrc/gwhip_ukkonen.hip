// gwhip_ukkonen.hip -- AlignerGlobalUkkonen's device path for MI355X (gfx950): unit-cost NW restricted to Ukkonen's
// band (parameter p), int16 scores, followed by the backtrace over the stored band.
//
// Replaces cudaaligner/src/ukkonen_gpu.cu (ukkonen_gpu :313-327 = ukkonen_compute_score_matrix :214-258 +
// ukkonen_backtrace_kernel :66-143). Results are bit-identical by construction: the band is kept in the reference's
// anti-diagonal coordinates, slot (k, l) = ((j - i + p) / 2, i + j), because its backtrace indexes that storage
// directly (including the k == 0 aliasing of the diagonal just left of the band, see oracle/global_oracle.c).
//
// MI355X mapping (not the reference's): one wavefront per pair instead of a (band-wide x 1) thread block that
// re-reads the previous two anti-diagonals from the global matrix behind two block barriers per step. All slots of
// one anti-diagonal are independent, so a step is one pass of the 64 lanes over k; the two previous anti-diagonals
// (rows l-1 and l-2 of the storage) live in LDS, both sequences are staged in LDS once, and the matrix in HBM is
// written exactly once per slot with lane-contiguous (row-major in l) stores -- the init pass of the reference
// (one more full write of the matrix) is folded into the same store. The backtrace runs on the same wavefront
// right after, three slots per step fetched by three lanes in one round trip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <string>

#include "../../include/gwhip.h"

namespace gwhip
{
extern thread_local std::string g_last_error;

namespace
{

constexpr int kWave       = 64;
constexpr int32_t kMaxVal = INT16_MAX - 1; // numeric_limits<nw_score_t>::max() - 1 (ukkonen_gpu.cu:81,146)
constexpr size_t kLdsCap  = 64 * 1024;

struct PairDims
{
    int32_t m, n; // lengths + 1, m <= n after the swap
    int32_t bw, cols;
    bool swapped;
};

__host__ __device__ inline PairDims pair_dims(int32_t query_length, int32_t target_length, int32_t p)
{
    PairDims d;
    d.m       = query_length + 1;
    d.n       = target_length + 1;
    d.swapped = d.m > d.n;
    if (d.swapped)
    {
        const int32_t t = d.m;
        d.m             = d.n;
        d.n             = t;
    }
    d.bw   = (1 + d.n - d.m + 2 * p + 1) / 2; // :93,232
    d.cols = d.n + d.m;
    return d;
}

struct UkkonenArgs
{
    int32_t n_pairs;
    const char* sequences;
    const int64_t* starts;
    int8_t* results;
    int32_t* result_lengths;
    const int64_t* offsets; // [n_pairs + 1] element offsets of the band storage
    int16_t* band;
    int64_t band_capacity; // elements
    int32_t p;
    int32_t row_lds_elems; // LDS elements reserved per anti-diagonal row (>= widest band)
    int32_t seq_lds_bytes; // LDS bytes reserved per sequence; 0: read the sequences from global memory
};

// element counts of the band storage, prefix-summed over the pairs
__global__ __launch_bounds__(1024) void ukkonen_offsets_kernel(const int64_t* starts, int64_t* offsets, int32_t n, int32_t p)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int32_t base = 0; base < n; base += 1024)
    {
        const int32_t i = base + threadIdx.x;
        int64_t v       = 0;
        if (i < n)
        {
            const PairDims d = pair_dims((int32_t)(starts[2 * i + 1] - starts[2 * i]), (int32_t)(starts[2 * i + 2] - starts[2 * i + 1]), p);
            v                = (int64_t)d.bw * d.cols;
        }
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            const int64_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) offsets[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}

template <bool LDS_SEQ>
__global__ __launch_bounds__(kWave) void ukkonen_kernel(UkkonenArgs a)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int lane     = threadIdx.x;
    const int32_t pair = blockIdx.x;
    const int32_t p    = a.p;

    const int64_t q_begin = a.starts[2 * pair], t_begin = a.starts[2 * pair + 1], t_end = a.starts[2 * pair + 2];
    const int32_t qlen = (int32_t)(t_begin - q_begin), tlen = (int32_t)(t_end - t_begin);
    const PairDims d   = pair_dims(qlen, tlen, p);
    const int32_t m = d.m, n = d.n, bw = d.bw, cols = d.cols;
    // the shorter sequence runs along i (:224-228); the backtrace swaps the gap states with it (:87-91)
    const char* seq_a = a.sequences + (d.swapped ? t_begin : q_begin);
    const char* seq_b = a.sequences + (d.swapped ? q_begin : t_begin);
    const int8_t insertion = d.swapped ? 3 : 2, deletion = d.swapped ? 2 : 3;

    int16_t* row_even = reinterpret_cast<int16_t*>(smem); // rows with (p + l) even
    int16_t* row_odd  = row_even + a.row_lds_elems;
    if constexpr (LDS_SEQ)
    {
        char* la = reinterpret_cast<char*>(row_odd + a.row_lds_elems);
        char* lb = la + a.seq_lds_bytes;
        for (int32_t x = lane; x < m - 1; x += kWave) la[x] = seq_a[x];
        for (int32_t x = lane; x < n - 1; x += kWave) lb[x] = seq_b[x];
        seq_a = la;
        seq_b = lb;
    }
    __syncthreads();

    if (a.offsets[pair] + (int64_t)bw * cols > a.band_capacity) // workspace smaller than gwhip_ukkonen_workspace_bytes() asked for
    {
        if (lane == 0) a.result_lengths[pair] = 0;
        return;
    }
    int16_t* S = a.band + a.offsets[pair]; // S[l * bw + k]
    const int32_t kmax_odd  = (n - m + 2 * p - 1) / 2 + 1; // :233-234
    const int32_t kmax_even = (n - m + 2 * p) / 2 + 1;

    for (int32_t l = 0; l < cols; l++)
    {
        const int32_t odd  = (p + l) & 1; // the diagonals handled at l: d = 2k + odd (:236-257)
        int16_t* cur       = odd ? row_odd : row_even; // holds row l-2, receives row l
        const int16_t* prv = odd ? row_even : row_odd; // row l-1
        const int32_t half = (p + l) / 2;
        const int32_t kmax = odd ? kmax_odd : kmax_even;
        for (int32_t k = lane; k < bw; k += kWave)
        {
            const int32_t j = k - half + l; // to_matrix_indices :42-47
            const int32_t i = l - j;
            int32_t v       = kMaxVal; // ukkonen_init_score_matrix :189-208
            if (i == 0) v = (int16_t)j;
            else if (j == 0) v = (int16_t)i;
            const int32_t dg   = 2 * k + odd;
            const int32_t lmin = abs(dg - p);
            const int32_t lmax = dg <= p ? 2 * (m - p + dg) + lmin : 2 * min(m, n - dg + p) + lmin;
            if (k < kmax && lmin + 1 <= l && l < lmax)
            {
                const int32_t cost = seq_a[i - 1] == seq_b[j - 1] ? 0 : 1;
                const int32_t diag = l - 2 < 0 ? kMaxVal : (int32_t)(int16_t)(cur[k] + cost);
                int32_t left, above;
                if (!odd) // ukkonen_compute_score_matrix_even :167-187
                {
                    left  = (k - 1 < 0 || l - 1 < 0) ? kMaxVal : (int32_t)(int16_t)(prv[k - 1] + 1);
                    above = l - 1 < 0 ? kMaxVal : (int32_t)(int16_t)(prv[k] + 1);
                }
                else // ukkonen_compute_score_matrix_odd :145-165
                {
                    left  = l - 1 < 0 ? kMaxVal : (int32_t)(int16_t)(prv[k] + 1);
                    above = (l - 1 < 0 || k + 1 >= bw) ? kMaxVal : (int32_t)(int16_t)(prv[k + 1] + 1);
                }
                v = min(diag, min(left, above));
            }
            cur[k]                  = (int16_t)v;
            S[(int64_t)l * bw + k] = (int16_t)v;
        }
        __syncthreads();
    }

    // ---- backtrace (ukkonen_backtrace_kernel :66-143) on the same wavefront ----
    __threadfence();
    __syncthreads();
    int8_t* path = a.results + q_begin;
    auto fetch = [&](int32_t ii, int32_t jj) -> int32_t {
        const int32_t k = (jj - ii + p) / 2; // to_band_indices :49-54, truncating division
        const int32_t l = jj + ii;
        return (k < 0 || k >= bw || l < 0 || l >= cols) ? kMaxVal : (int32_t)S[(int64_t)l * bw + k];
    };
    int32_t i = m - 1, j = n - 1;
    int32_t myscore = __builtin_amdgcn_readfirstlane(fetch(i, j));
    int32_t pos     = 0;
    while (i > 0 && j > 0)
    {
        // lane 0: above (i-1, j), lane 1: diag (i-1, j-1), lane 2: left (i, j-1) -- one round trip for all three
        const int32_t mine  = fetch(i - (lane != 2 ? 1 : 0), j - (lane != 0 ? 1 : 0));
        const int32_t above = __builtin_amdgcn_readlane(mine, 0);
        const int32_t diag  = __builtin_amdgcn_readlane(mine, 1);
        const int32_t left  = __builtin_amdgcn_readlane(mine, 2);
        int8_t r;
        if (left + 1 == myscore)
        {
            r       = insertion;
            myscore = left;
            --j;
        }
        else if (above + 1 == myscore)
        {
            r       = deletion;
            myscore = above;
            --i;
        }
        else
        {
            r       = diag == myscore ? 0 : 1;
            myscore = diag;
            --i;
            --j;
        }
        if (lane == 0) path[pos] = r;
        ++pos;
    }
    for (int32_t x = lane; x < i; x += kWave) path[pos + x] = deletion;
    pos += i;
    for (int32_t x = lane; x < j; x += kWave) path[pos + x] = insertion;
    pos += j;
    if (lane == 0) a.result_lengths[pair] = pos;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

} // namespace
} // namespace gwhip

using namespace gwhip;

extern "C" {

size_t gwhip_ukkonen_workspace_bytes(int32_t n_alignments, const int64_t* sequence_starts_host, int32_t ukkonen_p)
{
    if (n_alignments <= 0 || ukkonen_p < 0) return 256;
    int64_t elems = 0;
    for (int32_t i = 0; i < n_alignments; i++)
    {
        const PairDims d = pair_dims((int32_t)(sequence_starts_host[2 * i + 1] - sequence_starts_host[2 * i]),
                                     (int32_t)(sequence_starts_host[2 * i + 2] - sequence_starts_host[2 * i + 1]), ukkonen_p);
        elems += (int64_t)d.bw * d.cols;
    }
    return 256 + align_up(((size_t)n_alignments + 1) * 8, 256) + align_up((size_t)elems * 2, 256) + 256;
}

int gwhip_ukkonen(const gwhip_ukkonen_args* args, gwhip_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->n_alignments < 0 || args->ukkonen_p < 0 || args->max_length_difference < 0 || args->max_sequence_length < 0)
    {
        g_last_error = "gwhip_ukkonen: invalid arguments";
        return (int)hipErrorInvalidValue;
    }
    const int32_t n = args->n_alignments;
    if (n == 0) return 0;
    uint8_t* ws          = (uint8_t*)args->workspace;
    const size_t off_band = align_up(((size_t)n + 1) * 8, 256);
    if (!ws || ((uintptr_t)ws & 255) != 0 || args->workspace_bytes < off_band + 256)
    {
        g_last_error = "gwhip_ukkonen: workspace missing, misaligned or too small";
        return (int)hipErrorInvalidValue;
    }
    UkkonenArgs ka{};
    ka.n_pairs        = n;
    ka.sequences      = args->sequences;
    ka.starts         = args->sequence_starts;
    ka.results        = args->results;
    ka.result_lengths = args->result_lengths;
    ka.offsets        = reinterpret_cast<int64_t*>(ws);
    ka.band           = reinterpret_cast<int16_t*>(ws + off_band);
    ka.band_capacity  = (int64_t)((args->workspace_bytes - off_band) / sizeof(int16_t));
    ka.p              = args->ukkonen_p;
    // widest band of the batch: bw = (1 + |n - m| + 2p + 1) / 2 (:275)
    ka.row_lds_elems = (int32_t)align_up((size_t)(1 + args->max_length_difference + 2 * args->ukkonen_p + 1) / 2 + 1, 8);
    const size_t row_bytes = (size_t)ka.row_lds_elems * 2 * sizeof(int16_t);
    if (row_bytes > kLdsCap)
    {
        g_last_error = "gwhip_ukkonen: band wider than the LDS row buffers (length difference too large)";
        return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(ukkonen_offsets_kernel, dim3(1), dim3(1024), 0, stream, args->sequence_starts,
                       const_cast<int64_t*>(ka.offsets), n, args->ukkonen_p);
    const size_t seq_bytes = align_up((size_t)args->max_sequence_length + 4, 16);
    if (row_bytes + 2 * seq_bytes <= kLdsCap)
    {
        ka.seq_lds_bytes = (int32_t)seq_bytes;
        hipLaunchKernelGGL(ukkonen_kernel<true>, dim3(n), dim3(kWave), row_bytes + 2 * seq_bytes, stream, ka);
    }
    else
    {
        ka.seq_lds_bytes = 0;
        hipLaunchKernelGGL(ukkonen_kernel<false>, dim3(n), dim3(kWave), row_bytes, stream, ka);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        g_last_error = std::string("gwhip_ukkonen: kernel launch: ") + hipGetErrorString(e);
        return (int)e;
    }
    return 0;
}

} // extern "C"
