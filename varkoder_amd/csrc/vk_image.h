// vk_image.h -- K2: histograms -> rank-quantile images (sort and counting kernels), CGR table
// Part of the one translation unit vkimg.hip (device code for gfx950; see the notes there).
#ifndef VK_IMAGE_H
#define VK_IMAGE_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "vk_count.h"

namespace {

// --------------------------------------------------------------- K2 image ----

constexpr int kImgThreads = 1024;
constexpr uint32_t kTile = 16384;  // u32 elements sorted in LDS at a time (64 KiB)

__device__ __forceinline__ uint32_t revcomp_code(uint32_t c, int k) {
    // complement = 3 - b = ~b on 2 bits; reverse the k two-bit groups
    uint32_t x = ~c;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32 - 2 * k);
}

__device__ __forceinline__ void cmpswap(uint32_t& a, uint32_t& b, bool asc) {
    uint32_t lo = min(a, b), hi = max(a, b);
    a = asc ? lo : hi;
    b = asc ? hi : lo;
}

// All bitonic passes with stride < tile length, for merge size `size` and above
// (up to `maxsize`), on a tile that sits in LDS.  gbase = global index of tile[0].
__device__ void bitonic_tile(uint32_t* tile, uint32_t tlen, uint32_t gbase, uint32_t size_from,
                             uint32_t size_to, bool only_tail) {
    for (uint32_t size = size_from; size <= size_to; size <<= 1) {
        uint32_t s0 = only_tail ? tlen >> 1 : size >> 1;
        if (s0 > (tlen >> 1)) s0 = tlen >> 1;
        for (uint32_t stride = s0; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < (tlen >> 1); i += kImgThreads) {
                uint32_t lo = ((i / stride) * 2u * stride) + (i % stride);
                uint32_t hi = lo + stride;
                bool asc = ((gbase + lo) & size) == 0u;
                uint32_t a = tile[lo], b = tile[hi];
                cmpswap(a, b, asc);
                tile[lo] = a;
                tile[hi] = b;
            }
            __syncthreads();
        }
        if (only_tail) break;
    }
}

// One workgroup per sample.  scratch: [nsamples][2][npad] u32 (val, sorted).
// `only_if` (may be null): per-sample flags written by vk_image_count_kernel; a sample whose flag is 0
// is already done.
__global__ __launch_bounds__(kImgThreads) void vk_image_kernel(
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ pix, int k, uint32_t npix,
    uint32_t npad, uint32_t* __restrict__ scratch, uint8_t* __restrict__ img, const uint32_t* __restrict__ only_if) {
    __shared__ uint32_t tile[kTile];
    __shared__ unsigned long long bins[256];
    const uint32_t s = blockIdx.x;
    if (only_if && only_if[s] == 0u) return;  // uniform over the workgroup
    const uint32_t ncode = 1u << (2 * k);
    const uint32_t* h = hist + static_cast<uint64_t>(s) * ncode;
    uint32_t* val = scratch + static_cast<uint64_t>(s) * 2u * npad;
    uint32_t* srt = val + npad;
    const uint32_t tid = threadIdx.x;

    for (uint32_t i = tid; i < npad; i += kImgThreads) val[i] = 0u;
    __syncthreads();
    // strand merge + scatter: every code writes tot+1 to its own pixel; s and rc(s)
    // write the same value (to the same pixel for varKode, to two pixels for cgr)
    for (uint32_t c = tid; c < ncode; c += kImgThreads) {
        uint32_t r = revcomp_code(c, k);
        uint32_t tot = (r == c) ? h[c] : h[c] + h[r];
        val[pix[c]] = tot + 1u;
    }
    __syncthreads();
    for (uint32_t i = tid; i < npad; i += kImgThreads) srt[i] = (i < npix) ? val[i] : 0xFFFFFFFFu;
    __syncthreads();

    const uint32_t tlen = npad < kTile ? npad : kTile;
    const uint32_t ntiles = npad / tlen;
    // phase 1: sort every tile completely (directions follow the global index)
    for (uint32_t t = 0; t < ntiles; ++t) {
        for (uint32_t i = tid; i < tlen; i += kImgThreads) tile[i] = srt[t * tlen + i];
        __syncthreads();
        bitonic_tile(tile, tlen, t * tlen, 2u, tlen, false);
        if (ntiles > 1) {
            for (uint32_t i = tid; i < tlen; i += kImgThreads) srt[t * tlen + i] = tile[i];
            __syncthreads();
        }
    }
    // phase 2: merges wider than a tile: global passes, then the in-tile tail
    for (uint32_t size = tlen << 1; size <= npad && ntiles > 1; size <<= 1) {
        for (uint32_t stride = size >> 1; stride >= tlen; stride >>= 1) {
            for (uint32_t i = tid; i < (npad >> 1); i += kImgThreads) {
                uint32_t lo = ((i / stride) * 2u * stride) + (i % stride);
                uint32_t hi = lo + stride;
                bool asc = (lo & size) == 0u;
                uint32_t a = srt[lo], b = srt[hi];
                cmpswap(a, b, asc);
                srt[lo] = a;
                srt[hi] = b;
            }
            __syncthreads();
        }
        for (uint32_t t = 0; t < ntiles; ++t) {
            for (uint32_t i = tid; i < tlen; i += kImgThreads) tile[i] = srt[t * tlen + i];
            __syncthreads();
            bitonic_tile(tile, tlen, t * tlen, size, size, true);
            for (uint32_t i = tid; i < tlen; i += kImgThreads) srt[t * tlen + i] = tile[i];
            __syncthreads();
        }
    }
    const uint32_t* a = (ntiles > 1) ? srt : tile;

    // 256 quantile bins, scaled by 256 (exact integers)
    if (tid < 256) {
        unsigned long long pos = static_cast<unsigned long long>(tid) * (npix - 1u);
        uint32_t i = static_cast<uint32_t>(pos >> 8), g = static_cast<uint32_t>(pos & 255u);
        uint32_t i1 = (i + 1u < npix) ? i + 1u : npix - 1u;
        uint32_t ai = a[i], aj = a[i1];
        bins[tid] = 256ull * ai + static_cast<unsigned long long>(aj - ai) * g;
    }
    __syncthreads();
    uint8_t* out = img + static_cast<uint64_t>(s) * npix;
    for (uint32_t p = tid; p < npix; p += kImgThreads) {
        unsigned long long v = 256ull * val[p];
        // upper_bound over the non-decreasing bins; bins[0] = 256*min <= v, so the
        // answer lies in [1, 256]: 255 candidates to discard, 8 halvings
        uint32_t lo = 1, hi = 256;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            uint32_t mid = (lo + hi) >> 1;
            bool le = bins[mid] <= v;
            lo = le ? mid + 1u : lo;
            hi = le ? hi : mid;
        }
        out[p] = static_cast<uint8_t>(lo - 1u);
    }
}

// K2 for large images (k = 8, 9: 65k..262k pixels): the 256 quantile cut points need 512 order
// statistics, not a sorted array.  Pixel values below 2 x 32768 are COUNTED in a 32768-bin LDS
// histogram (one pass per half), a prefix scan turns the counts into ranks, and every wanted rank is
// looked up by binary search; the few larger values (outlier k-mers) are listed, sorted in LDS and
// indexed directly.  Exact like the sort, ~40x shorter for one 512 x 512 image.  A sample
// with more than 32768 values >= 65536 is left to vk_image_kernel (flag = 1).
constexpr uint32_t kCountBins = 32768;

// Strand merge + scatter of vk_image_count_kernel's samples as a launch of its own, <<<(4^k / 256, nsamples), 256>>> over
// a zeroed scratch: inside that kernel it is the work of ONE workgroup (one CU) per sample, and its reverse-complement
// gather touches 64 lines per wavefront -- 0.95 of the 1.4 ms the k = 9 images of 100 samples took.
__global__ __launch_bounds__(256) void vk_image_scatter_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ pix,
                                                               int k, uint32_t npad, uint32_t* __restrict__ scratch) {
    const uint32_t s = blockIdx.y, c = blockIdx.x * 256u + threadIdx.x;
    const uint32_t ncode = 1u << (2 * k);
    if (c >= ncode) return;
    const uint32_t* h = hist + static_cast<uint64_t>(s) * ncode;
    const uint32_t r = revcomp_code(c, k);
    const uint32_t tot = (r == c) ? h[c] : h[c] + h[r];
    scratch[static_cast<uint64_t>(s) * 2u * npad + pix[c]] = tot + 1u;
}

// scattered: vk_image_scatter_kernel has filled val already
__global__ __launch_bounds__(kImgThreads) void vk_image_count_kernel(
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ pix, int k, uint32_t npix,
    uint32_t npad, uint32_t* __restrict__ scratch, uint8_t* __restrict__ img, uint32_t* __restrict__ flags, int scattered) {
    __shared__ uint32_t cnt[kCountBins];
    __shared__ unsigned long long bins[256];
    __shared__ uint32_t order[512];   // [j] = a[i_j], [256 + j] = a[min(i_j + 1, npix - 1)]
    __shared__ uint32_t wsum[kImgThreads / 64];
    __shared__ uint32_t novf, any_hi;
    const uint32_t s = blockIdx.x;
    const uint32_t ncode = 1u << (2 * k);
    const uint32_t* h = hist + static_cast<uint64_t>(s) * ncode;
    uint32_t* val = scratch + static_cast<uint64_t>(s) * 2u * npad;
    uint32_t* ovf = val + npad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    if (tid == 0) { novf = 0u; any_hi = 0u; }
    if (!scattered) {
        for (uint32_t i = tid; i < npad; i += kImgThreads) val[i] = 0u;
        __syncthreads();
        for (uint32_t c = tid; c < ncode; c += kImgThreads) {  // strand merge + scatter, as in vk_image_kernel
            uint32_t r = revcomp_code(c, k);
            uint32_t tot = (r == c) ? h[c] : h[c] + h[r];
            val[pix[c]] = tot + 1u;
        }
    }
    __syncthreads();

    uint32_t my_rank = 0;  // thread t < 512 looks up one order statistic
    if (tid < 512) {
        const unsigned long long pos = static_cast<unsigned long long>(tid & 255u) * (npix - 1u);
        const uint32_t i = static_cast<uint32_t>(pos >> 8);
        my_rank = (tid < 256) ? i : ((i + 1u < npix) ? i + 1u : npix - 1u);
    }
    uint32_t base = 0;  // values counted by earlier passes
    for (uint32_t pass = 0; pass < 2; ++pass) {
        if (pass == 1 && __builtin_amdgcn_readfirstlane(static_cast<int>(any_hi)) == 0) break;  // uniform (and a scalar for hipcc: a wave scan follows): nothing in [32768, 65536)
        for (uint32_t i = tid; i < kCountBins; i += kImgThreads) cnt[i] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < npix; i += kImgThreads) {
            const uint32_t v = val[i], hi = v >> 15;
            if (hi == pass) {
                atomicAdd(&cnt[v & (kCountBins - 1u)], 1u);
            } else if (pass == 0) {
                if (hi == 1u) {
                    any_hi = 1u;
                } else {
                    const uint32_t at = atomicAdd(&novf, 1u);
                    ovf[at] = v;  // at < npix <= npad
                }
            }
        }
        __syncthreads();
        // inclusive scan of the counts, in place: 32 consecutive bins per thread
        uint32_t local = 0;
        const uint32_t b0 = tid * (kCountBins / kImgThreads);
#pragma unroll 8
        for (uint32_t b = 0; b < kCountBins / kImgThreads; ++b) local += cnt[b0 + b];
        const uint32_t incl = wave_inclusive_sum(local);
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - local;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        uint32_t run = before;
#pragma unroll 8
        for (uint32_t b = 0; b < kCountBins / kImgThreads; ++b) {
            run += cnt[b0 + b];
            cnt[b0 + b] = run;
        }
        __syncthreads();
        const uint32_t total = cnt[kCountBins - 1u];
        if (tid < 512 && my_rank >= base && my_rank - base < total) {
            const uint32_t r = my_rank - base;  // smallest bin with cumulative count > r
            uint32_t lo = 0, hi = kCountBins - 1u;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cnt[mid] > r) hi = mid; else lo = mid + 1u;
            }
            order[tid] = pass * kCountBins + lo;
        }
        base += total;
        __syncthreads();
    }
    const uint32_t n_over = novf;
    if (n_over > kCountBins) {  // uniform: too many large values for LDS, the sort kernel takes over
        if (tid == 0) flags[s] = 1u;
        return;
    }
    if (n_over > 0u) {
        uint32_t tlen = 2;
        while (tlen < n_over) tlen <<= 1;
        for (uint32_t i = tid; i < tlen; i += kImgThreads) cnt[i] = (i < n_over) ? ovf[i] : 0xFFFFFFFFu;
        __syncthreads();
        bitonic_tile(cnt, tlen, 0u, 2u, tlen, false);
        if (tid < 512 && my_rank >= base) order[tid] = cnt[my_rank - base];
        __syncthreads();
    }
    if (tid == 0) flags[s] = 0u;
    if (tid < 256) {
        const unsigned long long pos = static_cast<unsigned long long>(tid) * (npix - 1u);
        const uint32_t g = static_cast<uint32_t>(pos & 255u);
        const uint32_t ai = order[tid], aj = order[256u + tid];
        bins[tid] = 256ull * ai + static_cast<unsigned long long>(aj - ai) * g;
    }
    __syncthreads();
    uint8_t* out = img + static_cast<uint64_t>(s) * npix;
    for (uint32_t p = tid; p < npix; p += kImgThreads) {
        unsigned long long v = 256ull * val[p];
        uint32_t lo = 1, hi = 256;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            uint32_t mid = (lo + hi) >> 1;
            bool le = bins[mid] <= v;
            lo = le ? mid + 1u : lo;
            hi = le ? hi : mid;
        }
        out[p] = static_cast<uint8_t>(lo - 1u);
    }
}

__global__ void vk_cgr_lut_kernel(int k, uint32_t* __restrict__ pix) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 1u << (2 * k), side = 1u << k;
    if (c >= n) return;
    uint32_t x = 0, y = 0;
    for (int i = 0; i < k; ++i) {
        uint32_t b = (c >> (2 * (k - 1 - i))) & 3u;
        x |= ((b >> 1) & 1u) << i;
        y |= (((b >> 1) ^ b) & 1u) << i;
    }
    pix[c] = (side - 1u - y) * side + x;
}

}  // namespace

#endif  // VK_IMAGE_H
