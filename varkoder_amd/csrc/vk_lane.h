// vk_lane.h -- per-lane SWAR routines of the k-mer count kernel (K1).
//
// One lane owns 64 contiguous FASTQ bytes (16 dwords).  Everything a lane needs is
// derived with 32-bit SWAR arithmetic in a "2-bit geometry": position p (0..63) of the
// block lives at bits [2p, 2p+1] of a 128-bit string held in four dwords.
//   C   2-bit base codes (A0 C1 G2 T3; garbage where the byte is not a base)
//   IV  even bit set  <=> byte is not one of ACGTacgt
//   NL  even bit set  <=> byte is '\n'
// The dwords are first byte-transposed in groups of four (8 v_perm_b32), so that the
// four codes found in one dword are 8 bits apart and a shift-or of four such dwords
// yields 16 packed codes in position order -- no per-byte work anywhere.
//
// The same source compiles for the host (tests/test_lane_emulation.py builds a CPU
// emulation of a wavefront around it), where the gfx950 intrinsics have portable
// stand-ins.
#ifndef VK_LANE_H
#define VK_LANE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define VKL_FN __host__ __device__ __forceinline__
#else
#define VKL_FN inline
#endif

namespace vkl {

VKL_FN uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    // v_perm_b32: byte i of the result is byte sel[i] of the 8-byte pool {hi, lo}
    // (0..3 = lo, 4..7 = hi).  Only selectors 0..7 are used here.
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t pool = (static_cast<uint64_t>(hi) << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t s = (sel >> (8 * i)) & 7u;
        r |= static_cast<uint32_t>((pool >> (8 * s)) & 0xFFu) << (8 * i);
    }
    return r;
#endif
}

VKL_FN uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
    // ({hi, lo} >> sh)[31:0], sh in 0..31
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
#endif
}

VKL_FN uint32_t ffbl(uint32_t x) {
    // index of the lowest set bit, 0xFFFFFFFF for zero (v_ffbl_b32)
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__ffs(static_cast<int>(x)) - 1);
#else
    return x ? static_cast<uint32_t>(__builtin_ctz(x)) : 0xFFFFFFFFu;
#endif
}

VKL_FN uint32_t popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return static_cast<uint32_t>(__builtin_popcount(x));
#endif
}

VKL_FN uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

struct LaneBits {
    uint32_t C[4];
    uint32_t IV[4];
    uint32_t NL[4];
};

struct Mask128 {
    uint32_t w[4];
};

// bits [0, 2q) of the 128-bit string, q = 0..65 (65 behaves like 64).  The kernel
// keeps this table in LDS so that a lane gets a mask with one ds_read_b128.
VKL_FN Mask128 ones_below(uint32_t q) {
    Mask128 m;
    for (int g = 0; g < 4; ++g) {
        int n = 2 * static_cast<int>(q) - 32 * g;
        m.w[g] = n <= 0 ? 0u : (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u));
    }
    return m;
}

// Phase A: classify the 64 bytes.  Returns the number of newlines in the block.
VKL_FN uint32_t classify(const uint32_t d[16], LaneBits& o) {
    constexpr uint32_t kLutLo = 0x41204020u;  // low3 = 0..3 : inv, A(0x40|0), inv, C(0x40|1)
    constexpr uint32_t kLutHi = 0x42202053u;  // low3 = 4..7 : T(0x50|3), inv, inv, G(0x40|2)
    uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int g = 0; g < 4; ++g) {
        const uint32_t a0 = d[4 * g], a1 = d[4 * g + 1], a2 = d[4 * g + 2], a3 = d[4 * g + 3];
        // 4x4 byte transpose: byte i of T[j] = byte j of a_i  (position 16g + 4i + j)
        const uint32_t p01l = perm(a1, a0, 0x05010400u), p01h = perm(a1, a0, 0x07030602u);
        const uint32_t p23l = perm(a3, a2, 0x05010400u), p23h = perm(a3, a2, 0x07030602u);
        const uint32_t T[4] = {perm(p23l, p01l, 0x05040100u), perm(p23l, p01l, 0x07060302u),
                               perm(p23h, p01h, 0x05040100u), perm(p23h, p01h, 0x07060302u)};
        uint32_t C = 0, IV = 0, NN = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = T[j];
            // table lookup on the low 3 bits: expected high bits (case folded) | code
            const uint32_t L = perm(kLutHi, kLutLo, t & 0x07070707u);
            const uint32_t x = L ^ (t & 0xD8D8D8D8u);  // 0..3 for a base, else some bit of 0xFC
            C |= (x & 0x03030303u) << (2 * j);
            // bit 7 := byte is not a base: some bit of x & 0x7C set (carry trick) or bit 7 of x
            const uint32_t nz = ((x & 0x7C7C7C7Cu) + 0x7F7F7F7Fu) | x;
            IV |= (nz >> (7 - 2 * j)) & (0x01010101u << (2 * j));
            // bit 7 := byte is not '\n'  ((t7 ^ 0x0A) + 0x7F is one v_xad_u32)
            const uint32_t nn = (((t & 0x7F7F7F7Fu) ^ 0x0A0A0A0Au) + 0x7F7F7F7Fu) | t;
            NN |= (nn >> (7 - 2 * j)) & (0x01010101u << (2 * j));
        }
        o.C[g] = C;
        o.IV[g] = IV;
        o.NL[g] = ~NN & 0x55555555u;
        c += popc(o.NL[g]);
    }
    return c;
}

// Position (0..63) of the lowest newline recorded in nl[], 64 if there is none.
VKL_FN uint32_t first_newline(const uint32_t nl[4]) {
    uint32_t u0 = ffbl(nl[0]), u1 = ffbl(nl[1]) | 32u, u2 = ffbl(nl[2]) | 64u, u3 = ffbl(nl[3]) | 96u;
    uint32_t b = umin(umin(u0, u1), umin(u2, u3));  // 0xFFFFFFFF stays the maximum
    return umin(b >> 1, 64u);
}

// Sequence-line mask of a block with at most three newlines (the normal case: at most
// one stretch of sequence per 64 bytes).  lph = line phase at the block start
// (0 header, 1 sequence, 2 plus, 3 quality).  tbl(q) = ones_below(q).
template <typename Tbl>
VKL_FN Mask128 seq_mask_fast(const uint32_t NL[4], uint32_t lph, Tbl tbl) {
    const uint32_t d = (1u - lph) & 3u;  // newlines to skip before a sequence line starts
    const uint32_t p1 = first_newline(NL);
    Mask128 m = tbl(umin(p1 + 1u, 64u));
    uint32_t r[4] = {NL[0] & ~m.w[0], NL[1] & ~m.w[1], NL[2] & ~m.w[2], NL[3] & ~m.w[3]};
    const uint32_t p2 = first_newline(r);
    m = tbl(umin(p2 + 1u, 64u));
    r[0] &= ~m.w[0]; r[1] &= ~m.w[1]; r[2] &= ~m.w[2]; r[3] &= ~m.w[3];
    const uint32_t p3 = first_newline(r);
    // interval [s, e): after the d-th newline, up to the (d+1)-th
    const uint32_t s = d == 0 ? 0u : (d == 1 ? p1 + 1u : (d == 2 ? p2 + 1u : p3 + 1u));
    const uint32_t e = d == 0 ? p1 : (d == 1 ? p2 : (d == 2 ? p3 : 64u));
    const Mask128 ms = tbl(umin(s, 64u)), me = tbl(umin(e, 64u));
    Mask128 out;
    for (int g = 0; g < 4; ++g) out.w[g] = me.w[g] & ~ms.w[g];
    return out;
}

// Any number of newlines (degenerate FASTQ with very short lines): position by position.
VKL_FN Mask128 seq_mask_general(const uint32_t NL[4], uint32_t lph) {
    Mask128 out;
    uint32_t cur = lph & 3u;
    for (int g = 0; g < 4; ++g) {
        uint32_t acc = 0;
        const uint32_t nl = NL[g];
        for (uint32_t b = 0; b < 32; b += 2) {
            acc |= (cur == 1u ? 3u : 0u) << b;
            cur = (cur + ((nl >> b) & 1u)) & 3u;
        }
        out.w[g] = acc;
    }
    return out;
}

// BAD (even bits): the byte is not a base of a sequence line.
VKL_FN void bad_mask(const LaneBits& lb, const Mask128& seq, uint32_t bad[4]) {
    for (int g = 0; g < 4; ++g) bad[g] = (lb.IV[g] | ~seq.w[g]) & 0x55555555u;
}

// OK (even bit 2p set <=> the K-mer window ending at position p is countable) from the
// BAD string [badh | bad[0..3]], badh = BAD of the 16 positions before the block.
template <int K>
VKL_FN void ok_mask(uint32_t badh, const uint32_t bad[4], uint32_t ok[4]) {
    uint32_t w[5] = {badh, bad[0], bad[1], bad[2], bad[3]};
    int cover = 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int step = 0; step < 4; ++step) {
        if (cover < K) {
            const int shp = cover < K - cover ? cover : K - cover;  // positions
            const uint32_t sh = 2u * static_cast<uint32_t>(shp);
            for (int i = 4; i >= 1; --i) w[i] |= alignbit(w[i], w[i - 1], 32u - sh);
            w[0] |= w[0] << sh;
            cover += shp;
        }
    }
    for (int g = 0; g < 4; ++g) ok[g] = ~w[g + 1] & 0x55555555u;
}

// Window loop: for every position p with its OK bit set, emit(code << 2) where code is
// the K-mer ending at p, read from the code string [ch | C[0..3]] (ch = codes of the 16
// positions before the block).  First base most significant.
template <int K, typename Emit>
VKL_FN void windows(uint32_t ch, const uint32_t C[4], const uint32_t ok[4], Emit emit) {
    // Codes are packed with the EARLIEST position in the LOWEST bits, so a raw bit-field is
    // the k-mer with its first base least significant; the histogram convention wants the
    // first base most significant.  The kernel therefore counts into a bit-reversed index
    // space and un-reverses at flush time (see vk_count_kernel); here the raw field.
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    // OK bits are consumed from the top with x += x: the carry-out IS the lane predicate
    // (one v_add_co_u32 per position instead of and + compare).  OK lives on even bits, so two
    // dwords are interleaved (odd bits = the next dword) and every carry is a real position.
    uint32_t w[2] = {ok[0] | (ok[1] << 1), ok[2] | (ok[3] << 1)};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int t = 0; t < 32; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int h = 0; h < 2; ++h) {
            const int g = (t & 1) ? 2 * h : 2 * h + 1;   // bit 31-t: odd bits belong to dword 2h+1
            const int p = 16 * g + 15 - (t >> 1);
            const int o = 30 + 2 * (p - K + 1);           // bit offset of (field << 2) in v[]
            const int word = o >> 5, sh = o & 31;
            uint32_t val;
            if (sh + 2 * K + 2 <= 32) val = v[word] >> sh;
            else val = alignbit(v[word + 1], v[word], static_cast<uint32_t>(sh));
            const bool take = __builtin_add_overflow(w[h], w[h], &w[h]);
            if (take) emit(val & kMask4);
        }
    }
}

}  // namespace vkl
#endif  // VK_LANE_H
