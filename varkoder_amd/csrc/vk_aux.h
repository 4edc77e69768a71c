// vk_aux.h -- synthetic workload generator, `convert` remap, `query` input transform
// Part of the one translation unit vkimg.hip (device code for gfx950; see the notes there).
#ifndef VK_AUX_H
#define VK_AUX_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

// ------------------------------------------------------------------ synth ----

__device__ __host__ inline uint64_t vk_mix(uint64_t seed, uint64_t s, uint64_t r, uint64_t w, uint64_t stream) {
    uint64_t z = seed + s * 0x9E3779B97F4A7C15ull + r * 0xBF58476D1CE4E5B9ull + w * 0x94D049BB133111EBull +
                 stream * 0xD6E8FEB86659FD93ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline uint8_t synth_base(uint64_t seed, uint32_t s, uint32_t r, uint32_t i, uint32_t readlen, int dist) {
    const uint32_t w = i >> 4, j = i & 15u;
    uint32_t b;
    if (dist == 0) {
        b = static_cast<uint32_t>(vk_mix(seed, s, r, w, 1) >> (2 * j)) & 3u;
    } else {
        // GC content gq/16 per sample, gq in 4..10; 4 random bits per base
        uint32_t gq = 4u + static_cast<uint32_t>(vk_mix(seed, s, 0, 0, 3) % 7u);
        uint32_t u = static_cast<uint32_t>(vk_mix(seed, s, r, w, 1) >> (4 * j)) & 15u;
        uint32_t at = 16u - gq, a = (at + 1u) >> 1, cc = (gq + 1u) >> 1, g = gq >> 1;
        b = u < a ? 0u : (u < a + cc ? 1u : (u < a + cc + g ? 2u : 3u));
        // 1 read in 200 carries a homopolymer run of 20..60 bases
        uint64_t hr = vk_mix(seed, s, r, 0, 4);
        if (hr % 200u == 0u && readlen > 64u) {
            uint32_t rl = 20u + static_cast<uint32_t>((hr >> 16) % 41u);
            uint32_t st = static_cast<uint32_t>((hr >> 32) % (readlen - rl));
            if (i >= st && i < st + rl) b = static_cast<uint32_t>(hr >> 8) & 3u;
        }
    }
    uint64_t hn = vk_mix(seed, s, r, w, 2);
    if (((hn >> 8) & 63u) == 0u && (hn & 15u) == j) return 'N';
    return "ACGT"[b];
}

// one thread per 16 output bytes
__global__ void vk_synth_kernel(uint8_t* __restrict__ out, uint32_t sample0, uint32_t nsamples, uint32_t reads,
                                uint32_t readlen, uint64_t seed, int dist, uint64_t total16) {
    const uint64_t rec = 2ull * readlen + 20ull;
    for (uint64_t g = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; g < total16;
         g += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        uint8_t bytes[16];
        const uint64_t o0 = g * 16;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            uint64_t o = o0 + t;
            uint64_t ridx = o / rec;
            uint32_t off = static_cast<uint32_t>(o % rec);
            uint32_t sl = static_cast<uint32_t>(ridx / reads);
            uint32_t r = static_cast<uint32_t>(ridx % reads);
            uint32_t s = sample0 + sl;
            uint8_t ch;
            if (sl >= nsamples) {
                ch = 0;
            } else if (off < 16) {
                // "@sSSSSS.RRRRRRR\n"
                if (off == 0) ch = '@';
                else if (off == 1) ch = 's';
                else if (off < 7) {
                    uint32_t p10 = 1;
                    for (uint32_t e = 0; e < 6 - off; ++e) p10 *= 10;
                    ch = '0' + (s / p10) % 10;
                } else if (off == 7) ch = '.';
                else if (off < 15) {
                    uint32_t p10 = 1;
                    for (uint32_t e = 0; e < 14 - off; ++e) p10 *= 10;
                    ch = '0' + (r / p10) % 10;
                } else ch = '\n';
            } else if (off < 16 + readlen) {
                ch = synth_base(seed, s, r, off - 16, readlen, dist);
            } else if (off == 16 + readlen) ch = '\n';
            else if (off == 17 + readlen) ch = '+';
            else if (off == 18 + readlen) ch = '\n';
            else if (off < 19 + 2 * readlen) ch = 'I';
            else ch = '\n';
            bytes[t] = ch;
        }
        uint4 v;
        memcpy(&v, bytes, 16);
        *reinterpret_cast<uint4*>(out + o0) = v;
    }
}

// ---- dist 2: reads shaped like what step B of the reference hands to step D -------------------------
// fastp runs with --merge --include_unmerged and --disable_length_filtering (commands/image.py:405,426-427,
// 494-495): the files step D counts hold reads of every length from 0 to about twice the read length under
// long headers.  Per read (key stream 5): 65 % readlen bases, 20 % merged pairs of readlen+1 .. 2*readlen-10,
// 10 % trimmed reads of 45 .. readlen-1, 5 % of 0 .. 44 (empty ones included); header lines of 40 .. 70
// bytes; quality characters '!' .. 'I' (so '@' and '+' occur in them, also first in the line).  Bases as
// dist 0 (uniform, N at ~1e-3).  varkoder_amd/synth.py is the bit-identical host generator.
struct SynthRead {
    uint32_t hl;   // bytes of the header line, its newline included
    uint32_t len;  // bases
};

__device__ __host__ inline SynthRead synth_read_shape(uint64_t seed, uint32_t s, uint32_t r, uint32_t readlen) {
    const uint64_t h = vk_mix(seed, s, r, 0, 5);
    const uint32_t u = static_cast<uint32_t>(h % 100u), v = static_cast<uint32_t>((h >> 8) & 0xFFFFFFu);
    SynthRead o;
    o.len = u < 65u ? readlen : (u < 85u ? readlen + 1u + v % (readlen - 10u) : (u < 95u ? 45u + v % (readlen - 45u) : v % 45u));
    o.hl = 40u + static_cast<uint32_t>((h >> 40) % 31u);
    return o;
}

__device__ __host__ inline uint32_t synth_record_bytes(const SynthRead& sr) { return sr.hl + 2u * sr.len + 4u; }

// per sample: exclusive prefix sums of the record sizes, recoff[sample][0 .. reads] (u32: a sample stays below 4 GiB)
__global__ __launch_bounds__(1024) void vk_synth_shape_kernel(uint32_t* __restrict__ recoff, uint32_t sample0, uint32_t reads,
                                                              uint32_t readlen, uint64_t seed) {
    __shared__ uint32_t part[1024];
    const uint32_t s = sample0 + blockIdx.x, tid = threadIdx.x;
    const uint32_t per = (reads + 1023u) / 1024u;
    const uint32_t r0 = tid * per < reads ? tid * per : reads, r1 = r0 + per < reads ? r0 + per : reads;
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; ++r) sum += synth_record_bytes(synth_read_shape(seed, s, r, readlen));
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {   // inclusive scan of the threads' sums
        const uint32_t add = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t at = part[tid] - sum;
    uint32_t* out = recoff + static_cast<uint64_t>(blockIdx.x) * (reads + 1u);
    for (uint32_t r = r0; r < r1; ++r) {
        out[r] = at;
        at += synth_record_bytes(synth_read_shape(seed, s, r, readlen));
    }
    if (tid == 1023u) out[reads] = part[1023];
}

__device__ inline uint8_t synth_shaped_byte(uint64_t seed, uint32_t s, uint32_t r, uint32_t off, const SynthRead& sr) {
    if (off < sr.hl) {
        if (off == 0) return '@';
        if (off == 1) return 's';
        if (off < 7) {
            uint32_t p10 = 1;
            for (uint32_t e = 0; e < 6 - off; ++e) p10 *= 10;
            return static_cast<uint8_t>('0' + (s / p10) % 10);
        }
        if (off == 7) return '.';
        if (off < 15) {
            uint32_t p10 = 1;
            for (uint32_t e = 0; e < 14 - off; ++e) p10 *= 10;
            return static_cast<uint8_t>('0' + (r / p10) % 10);
        }
        if (off == sr.hl - 1u) return '\n';
        if (off == 15) return ' ';
        return static_cast<uint8_t>("ABCDEFGHIJKLMNOPQRSTUVWXYZ:_/=0123456789"[(off + r) % 40u]);
    }
    off -= sr.hl;
    if (off < sr.len) return synth_base(seed, s, r, off, sr.len, 0);
    off -= sr.len;
    if (off == 0) return '\n';
    if (off == 1) return '+';
    if (off == 2) return '\n';
    off -= 3;
    if (off < sr.len) return static_cast<uint8_t>(33u + static_cast<uint32_t>((vk_mix(seed, s, r, off >> 3, 6) >> (8u * (off & 7u))) & 0xFFu) % 41u);
    return '\n';
}

// one workgroup per (sample, 256 reads): the 16-byte granules whose FIRST byte lies in those reads
__global__ __launch_bounds__(256) void vk_synth_shaped_kernel(uint8_t* __restrict__ out, const uint64_t* __restrict__ soffs,
                                                              const uint32_t* __restrict__ recoff, uint32_t sample0,
                                                              uint32_t reads, uint32_t readlen, uint64_t seed) {
    __shared__ uint32_t off[258];
    const uint32_t chunks = (reads + 255u) / 256u;
    const uint32_t sl = blockIdx.x / chunks, c = blockIdx.x % chunks, s = sample0 + sl;
    const uint32_t r0 = c * 256u;
    const uint32_t* ro = recoff + static_cast<uint64_t>(sl) * (reads + 1u);
    const uint32_t total = ro[reads];
    for (uint32_t i = threadIdx.x; i < 258u; i += 256u) off[i] = r0 + i <= reads ? ro[r0 + i] : 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t nr = reads - r0 < 256u ? reads - r0 : 256u;      // reads of this chunk
    const uint32_t b0 = off[0], b1 = off[nr];                       // its bytes
    const uint32_t end = c + 1u == chunks ? (total + 15u) / 16u * 16u : b1;   // the last chunk pads the sample with zeros
    uint8_t* dst = out + soffs[sl];
    for (uint32_t g = (b0 + 15u) / 16u + threadIdx.x; g * 16u < end; g += 256u) {
        const uint32_t o0 = g * 16u;
        uint32_t lo = 0, hi = nr + 1u;   // the read (0 .. nr: one beyond the chunk may be reached) that holds byte o0
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= o0) lo = mid; else hi = mid;
        }
        uint32_t ri = lo;
        SynthRead sr = synth_read_shape(seed, s, r0 + ri, readlen);
        uint8_t bytes[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t o = o0 + t;
            if (o >= total) { bytes[t] = 0; continue; }
            if (o >= off[ri + 1u]) {   // (records are longer than 16 bytes: one step at most)
                ++ri;
                sr = synth_read_shape(seed, s, r0 + ri, readlen);
            }
            bytes[t] = synth_shaped_byte(seed, s, r0 + ri, o - off[ri], sr);
        }
        uint4 v;
        memcpy(&v, bytes, 16);
        *reinterpret_cast<uint4*>(dst + o0) = v;
    }
}

// ----------------------------------------------------------------- remap ----
// convert.py:34-77 as a gather: out[p] = in[src0[p]] (0xFFFFFFFF = unmapped -> 0), or with
// sum_rc the uint8-wrapping weighted sum of two source pixels followed by the reference's
// float64 min/max rescale.  One workgroup per image.
__global__ __launch_bounds__(256) void vk_remap_kernel(const uint8_t* __restrict__ in, uint32_t npix_in,
                                                        uint32_t npix_out, const uint32_t* __restrict__ src0,
                                                        const uint32_t* __restrict__ src1,
                                                        const uint8_t* __restrict__ w0, const uint8_t* __restrict__ w1,
                                                        int sum_rc, uint8_t* __restrict__ out) {
    __shared__ uint32_t red_min[256], red_max[256];
    const uint8_t* img = in + static_cast<uint64_t>(blockIdx.x) * npix_in;
    uint8_t* o = out + static_cast<uint64_t>(blockIdx.x) * npix_out;
    const uint32_t tid = threadIdx.x;
    if (!sum_rc) {
        for (uint32_t p = tid; p < npix_out; p += 256) {
            uint32_t s0 = src0[p];
            o[p] = s0 == 0xFFFFFFFFu ? 0 : img[s0];
        }
        return;
    }
    uint32_t mn = 255, mx = 0;
    for (uint32_t p = tid; p < npix_out; p += 256) {
        uint32_t s0 = src0[p], s1 = src1[p];
        uint32_t a = s0 == 0xFFFFFFFFu ? 0u : img[s0], b = s1 == 0xFFFFFFFFu ? 0u : img[s1];
        uint32_t v = (a * w0[p] + b * w1[p]) & 0xFFu;  // np.add.at on a uint8 array wraps
        o[p] = static_cast<uint8_t>(v);
        mn = min(mn, v);
        mx = max(mx, v);
    }
    red_min[tid] = mn;
    red_max[tid] = mx;
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            red_min[tid] = min(red_min[tid], red_min[tid + st]);
            red_max[tid] = max(red_max[tid], red_max[tid + st]);
        }
        __syncthreads();
    }
    mn = red_min[0];
    mx = red_max[0];
    for (uint32_t p = tid; p < npix_out; p += 256) {
        uint32_t v = o[p];
        // np.uint8((arr - arr.min()) / arr.max() * 255): float64 divide, multiply, truncate
        double r = mx ? static_cast<double>(v - mn) / static_cast<double>(mx) * 255.0 : 0.0;
        o[p] = static_cast<uint8_t>(static_cast<uint32_t>(r));
    }
}

// ------------------------------------------------------------ query preprocessing ----
// fastai's inference-time item/batch transforms for a fixed-input-size timm model
// (commands/train.py:236-245: Resize(squish, BOX) to the model's input size; IntToFloatTensor;
// Normalize(mean, std)): PIL's 8-bit BOX resample = two separable passes with 22-bit fixed-point
// coefficients and an 8-bit intermediate, then (v/255 - mean)/std in float32, grey replicated to
// 3 channels.  One workgroup per image; the [side][out] intermediate lives in LDS.
__global__ __launch_bounds__(256) void vk_preprocess_kernel(const uint8_t* __restrict__ img, uint32_t side,
                                                             uint32_t out, const int32_t* __restrict__ bounds,
                                                             const int32_t* __restrict__ coef, uint32_t kmax,
                                                             float mean, float stdv, float* __restrict__ dst) {
    extern __shared__ uint8_t tmp[];  // [side][out]
    const uint8_t* src = img + static_cast<uint64_t>(blockIdx.x) * side * side;
    float* o = dst + static_cast<uint64_t>(blockIdx.x) * 3u * out * out;
    for (uint32_t i = threadIdx.x; i < side * out; i += blockDim.x) {
        const uint32_t y = i / out, xx = i % out;
        const int32_t x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        int32_t ss = 1 << 21;
        for (int32_t k = 0; k < n; ++k) ss += static_cast<int32_t>(src[y * side + x0 + k]) * coef[xx * kmax + k];
        ss >>= 22;
        tmp[i] = static_cast<uint8_t>(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < out * out; i += blockDim.x) {
        const uint32_t yy = i / out, xx = i % out;
        const int32_t y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        int32_t ss = 1 << 21;
        for (int32_t k = 0; k < n; ++k) ss += static_cast<int32_t>(tmp[(y0 + k) * out + xx]) * coef[yy * kmax + k];
        ss >>= 22;
        const float v = static_cast<float>(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
        const float f = (v / 255.0f - mean) / stdv;
        o[i] = f;
        o[out * out + i] = f;
        o[2u * out * out + i] = f;
    }
}

}  // namespace

#endif  // VK_AUX_H
