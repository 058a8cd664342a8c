// Exact top-k selection inside one 1024-thread workgroup (shared by the RPN top-k and the seeded anchor / RoI sampling).
//   select_topk(sm, n, k, keys_scratch, key_fn): key_fn(i) -> uint32 for i in [0,n); 0 means "not a candidate".
//   Selects the min(k, #candidates) largest keys, ties by ascending index, and leaves them sorted by (key desc, index asc)
//   in sm.sorted[] as 64-bit words (key << 32) | (0xffffffff - index).  Returns the number selected.
// Method: 4 x 8-bit radix-select rounds with LDS histograms find the k-th largest key T; a wave-ballot compaction in index
// order places the keys > T and the first (k - #greater) keys == T; an LDS bitonic sort orders the <= 2048 survivors.
#pragma once
#include "common.h"

namespace amp {

constexpr int SELECT_THREADS = 1024;
constexpr int SELECT_MAX_K = 2048;
constexpr int SELECT_UNROLL = 8;      // elements in flight per thread in the passes of the memory version

constexpr int SELECT_HCOPIES = 16;      // private histograms (lane & 15): the top digits of a batch of logits fall into two or three bins,
constexpr int SELECT_HSTRIDE = 257;     // and 64 lanes adding to one LDS word are 64 serial passes; 257: the copies start in different banks
struct SelectSmem {
    unsigned int hist[256];
    unsigned int hist_p[SELECT_HCOPIES * SELECT_HSTRIDE];
    unsigned int prefix, remaining, ncand;
    unsigned int wave_gt[16], wave_eq[16];
    unsigned long long sorted[SELECT_MAX_K];
};

// In-LDS bitonic sort, descending, of N (power of two, >= 64) 64-bit words by all NT threads of the block.
// Compare-exchange steps with a stride below 64 stay inside a 64-word group: a wave holds the group in registers (one word per lane) and
// exchanges with __shfl_xor -- no barrier, no LDS round trip.  Only the steps with stride >= 64 go through LDS with a barrier each: 10 of
// the 55 steps of N = 1024, 28 of 91 for N = 8192 (every step used to cost a barrier of 16 waves: ~0.5 us).
__device__ __forceinline__ unsigned long long bitonic_wave_steps(unsigned long long v, int idx, int lane, int size, int first_stride) {
    for (int stride = first_stride; stride > 0; stride >>= 1) {
        const unsigned long long p = __shfl_xor(v, stride, 64);
        const bool take_max = ((lane & stride) == 0) == ((idx & size) == 0);       // the lower position of a descending pair keeps the larger word
        const unsigned long long mx = v > p ? v : p, mn = v > p ? p : v;
        v = take_max ? mx : mn;
    }
    return v;
}

template <int NT>
__device__ inline void bitonic_desc(unsigned long long* s, int N) {
    const int lane = threadIdx.x & 63;
    __syncthreads();
    for (int g0 = (int)(threadIdx.x & ~63u); g0 < N; g0 += NT) {           // sizes 2 .. 64: entirely inside the groups
        unsigned long long v = s[g0 + lane];
        for (int size = 2; size <= 64; size <<= 1) v = bitonic_wave_steps(v, g0 + lane, lane, size, size >> 1);
        s[g0 + lane] = v;
    }
    for (int size = 128; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride >= 64; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (N >> 1); t += NT) {
                const int lo = (t / stride) * (stride << 1) + (t % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = s[lo], b = s[hi];
                if ((a < b) == desc) { s[lo] = b; s[hi] = a; }
            }
        }
        __syncthreads();
        for (int g0 = (int)(threadIdx.x & ~63u); g0 < N; g0 += NT)
            s[g0 + lane] = bitonic_wave_steps(s[g0 + lane], g0 + lane, lane, size, 32);
    }
    __syncthreads();
}

// histogram plumbing shared by the two selections (all threads of the 1024-thread block call these)
__device__ __forceinline__ void select_hist_clear(SelectSmem& sm) {
    for (int i = threadIdx.x; i < SELECT_HCOPIES * SELECT_HSTRIDE; i += SELECT_THREADS) sm.hist_p[i] = 0;
}
__device__ __forceinline__ void select_hist_add(SelectSmem& sm, unsigned int digit) {
    atomicAdd(&sm.hist_p[(threadIdx.x & (SELECT_HCOPIES - 1)) * SELECT_HSTRIDE + digit], 1u);
}
// after a barrier: hist[d] = sum of the copies; then (another barrier inside) wave 0 finds the digit that holds the `remaining`-th largest
// key -- lane l owns the four bins 255 - 4 l ... 252 - 4 l, a wave scan gives the number of keys above its group -- and updates
// sm.prefix / sm.remaining exactly as the serial walk from bin 255 downwards did.
__device__ __forceinline__ void select_hist_pick(SelectSmem& sm, int shift) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        unsigned int c = 0;
#pragma unroll
        for (int q = 0; q < SELECT_HCOPIES; ++q) c += sm.hist_p[q * SELECT_HSTRIDE + tid];
        sm.hist[tid] = c;
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned int rem = sm.remaining;
        unsigned int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = sm.hist[255 - 4 * tid - q];
        const unsigned int sum = c[0] + c[1] + c[2] + c[3];
        unsigned int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int o = __shfl_up(incl, d, 64);
            if (tid >= d) incl += o;
        }
        const unsigned int excl = incl - sum;
        if (excl < rem && rem <= incl) {
            unsigned int r = rem - excl;
            int q = 3;
            if (c[0] >= r) q = 0;
            else {
                r -= c[0];
                if (c[1] >= r) q = 1;
                else {
                    r -= c[1];
                    if (c[2] >= r) q = 2; else r -= c[2];
                }
            }
            sm.prefix |= ((uint32_t)(255 - 4 * tid - q) << shift);
            sm.remaining = r;
        }
    }
    __syncthreads();
}

template <class KeyFn>
__device__ inline int select_topk(SelectSmem& sm, int n, int kmax, uint32_t* keys, KeyFn key_fn) {
    const int tid = threadIdx.x;
    __syncthreads();   // previous users of sm are done
    select_hist_clear(sm);
    if (tid == 0) sm.ncand = 0;
    __syncthreads();
    // (SELECT_UNROLL independent elements per trip in every pass over the keys: the passes are chains of L2 round trips -- 256 per thread at
    //  n = 262 144 -- and their latency, not the arithmetic, is the kernel)
    unsigned int mine = 0;
    for (int i0 = tid; i0 < n; i0 += SELECT_UNROLL * SELECT_THREADS) {
        uint32_t key[SELECT_UNROLL];
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {
            const int i = i0 + u * SELECT_THREADS;
            key[u] = (i < n) ? key_fn(i) : 0u;
        }
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {
            const int i = i0 + u * SELECT_THREADS;
            if (i < n) keys[i] = key[u];
            if (key[u]) { select_hist_add(sm, key[u] >> 24); ++mine; }
        }
    }
    if (mine) atomicAdd(&sm.ncand, mine);
    __syncthreads();
    const int k = min(kmax, (int)sm.ncand);
    if (k <= 0) return 0;   // uniform
    if (tid == 0) { sm.prefix = 0; sm.remaining = (unsigned)k; }
    __syncthreads();
    for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        if (round > 0) {
            select_hist_clear(sm);
            __syncthreads();
            const uint32_t prefix = sm.prefix;
            const uint32_t himask = 0xffffffffu << (shift + 8);
            for (int i0 = tid; i0 < n; i0 += SELECT_UNROLL * SELECT_THREADS) {
                uint32_t key[SELECT_UNROLL];
#pragma unroll
                for (int u = 0; u < SELECT_UNROLL; ++u) {
                    const int i = i0 + u * SELECT_THREADS;
                    key[u] = (i < n) ? keys[i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < SELECT_UNROLL; ++u)
                    if (key[u] && (key[u] & himask) == prefix) select_hist_add(sm, (key[u] >> shift) & 0xff);
            }
            __syncthreads();
        }
        select_hist_pick(sm, shift);
    }
    const uint32_t T = sm.prefix;                 // k-th largest key (> 0 because only candidates were counted)
    const unsigned int need_eq = sm.remaining;    // keys == T to take, smallest indices first
    const int wave = tid >> 6, lane = tid & 63;
    const int chunk = ((n + 16 * 64 - 1) / (16 * 64)) * 64;
    const int beg = wave * chunk, end = min(n, beg + chunk);
    unsigned int cgt = 0, ceq = 0;
    for (int i0 = beg; i0 < end; i0 += SELECT_UNROLL * 64) {
        uint32_t key[SELECT_UNROLL];
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {
            const int i = i0 + u * 64 + lane;
            key[u] = (i < end) ? keys[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {
            cgt += __popcll(__ballot(key[u] > T));
            ceq += __popcll(__ballot(key[u] == T));
        }
    }
    if (lane == 0) { sm.wave_gt[wave] = cgt; sm.wave_eq[wave] = ceq; }
    for (int i = tid; i < SELECT_MAX_K; i += SELECT_THREADS) sm.sorted[i] = 0ull;
    __syncthreads();
    unsigned int rgt = 0, req = 0, gt_total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { rgt += sm.wave_gt[w]; req += sm.wave_eq[w]; }
        gt_total += sm.wave_gt[w];
    }
    for (int j0 = beg; j0 < end; j0 += SELECT_UNROLL * 64) {
        uint32_t key4[SELECT_UNROLL];
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {
            const int i = j0 + u * 64 + lane;
            key4[u] = (i < end) ? keys[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < SELECT_UNROLL; ++u) {          // in index order, as before
            const int i = j0 + u * 64 + lane;
            const uint32_t key = key4[u];
            const bool gt = key > T, eq = key == T;
            const unsigned long long mgt = __ballot(gt), meq = __ballot(eq);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const unsigned long long word = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
            if (gt) sm.sorted[rgt + __popcll(mgt & below)] = word;
            if (eq) {
                const unsigned int r = req + __popcll(meq & below);
                if (r < need_eq) sm.sorted[gt_total + r] = word;
            }
            rgt += __popcll(mgt);
            req += __popcll(meq);
        }
    }
    __syncthreads();
    int N = 64;
    while (N < k) N <<= 1;
    bitonic_desc<SELECT_THREADS>(sm.sorted, N);
    return k;
}

// The same selection with the keys held in registers: n <= PER * 1024 keys, thread (wave w, lane l) owns
// i = w * PER * 64 + j * 64 + l for j < PER -- a contiguous range per wave, so the index-ordered compaction needs no second layout.
// The memory version re-reads its keys from L2 in every one of its seven passes; for the RPN's chunks of <= 49 152 logits that
// latency, not the arithmetic, was the kernel.
template <int PER, class KeyFn>
__device__ inline int select_topk_reg(SelectSmem& sm, int n, int kmax, KeyFn key_fn) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int base = wave * PER * 64 + lane;
    __syncthreads();   // previous users of sm are done
    select_hist_clear(sm);
    if (tid == 0) sm.ncand = 0;
    __syncthreads();
    uint32_t kreg[PER];
    unsigned int mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = base + j * 64;
        kreg[j] = (i < n) ? key_fn(i) : 0u;
        if (kreg[j]) { select_hist_add(sm, kreg[j] >> 24); ++mine; }
    }
    if (mine) atomicAdd(&sm.ncand, mine);
    __syncthreads();
    const int k = min(kmax, (int)sm.ncand);
    if (k <= 0) return 0;   // uniform
    if (tid == 0) { sm.prefix = 0; sm.remaining = (unsigned)k; }
    __syncthreads();
    for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        if (round > 0) {
            select_hist_clear(sm);
            __syncthreads();
            const uint32_t prefix = sm.prefix;
            const uint32_t himask = 0xffffffffu << (shift + 8);
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (kreg[j] && (kreg[j] & himask) == prefix) select_hist_add(sm, (kreg[j] >> shift) & 0xff);
            __syncthreads();
        }
        select_hist_pick(sm, shift);
    }
    const uint32_t T = sm.prefix;
    const unsigned int need_eq = sm.remaining;
    unsigned int cgt = 0, ceq = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        cgt += __popcll(__ballot(kreg[j] > T));
        ceq += __popcll(__ballot(kreg[j] == T));
    }
    if (lane == 0) { sm.wave_gt[wave] = cgt; sm.wave_eq[wave] = ceq; }
    for (int i = tid; i < SELECT_MAX_K; i += SELECT_THREADS) sm.sorted[i] = 0ull;
    __syncthreads();
    unsigned int rgt = 0, req = 0, gt_total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { rgt += sm.wave_gt[w]; req += sm.wave_eq[w]; }
        gt_total += sm.wave_gt[w];
    }
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t key = kreg[j];
        const bool gt = key > T, eq = key == T;
        const unsigned long long mgt = __ballot(gt), meq = __ballot(eq);
        const unsigned long long word = ((unsigned long long)key << 32) | (uint32_t)(0xffffffffu - (uint32_t)(base + j * 64));
        if (gt) sm.sorted[rgt + __popcll(mgt & below)] = word;
        if (eq) {
            const unsigned int r = req + __popcll(meq & below);
            if (r < need_eq) sm.sorted[gt_total + r] = word;
        }
        rgt += __popcll(mgt);
        req += __popcll(meq);
    }
    __syncthreads();
    int N = 64;
    while (N < k) N <<= 1;
    bitonic_desc<SELECT_THREADS>(sm.sorted, N);
    return k;
}

}  // namespace amp
