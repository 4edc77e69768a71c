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
    // index of the lowest set bit, 0xFFFFFFFF for zero: exactly v_ffbl_b32 (spelled out: __ffs
    // wraps it in a compare + select for the zero case, which the hardware already handles)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_ffbl_b32 %0, %1" : "=v"(d) : "v"(x));
    return d;
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

// Three-operand VALU forms that hipcc does not select on gfx950 when one operand is a 32-bit
// literal (VOP3 cannot encode literals on gfx9): spelled out, with the constant in an SGPR.
// (Two-operand and / add / xor keep their constants as LITERALS: measured on gfx950 at 8 waves per
// SIMD, tools/issue_rate.hip, a VOP2 with a literal issues at the full rate, the same instruction
// with an SGPR operand at ~0.6 of it, any three-operand VOP3 / DPP / SDWA form at ~0.5.)
// Plain VGPR-to-VGPR ALU instructions: no memory access, no extra wait states.
VKL_FN uint32_t and_or_k(uint32_t a, uint32_t kmask, uint32_t c) {  // (a & kmask) | c
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(kmask), "v"(c));
    return d;
#else
    return (a & kmask) | c;
#endif
}

VKL_FN uint32_t xor_add_k(uint32_t a, uint32_t kx, uint32_t vadd) {  // (a ^ kx) + vadd
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(kx), "v"(vadd));
    return d;
#else
    return (a ^ kx) + vadd;
#endif
}

VKL_FN uint32_t xor_and_k(uint32_t a, uint32_t b, uint32_t kmask) {  // a ^ (b & kmask)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x78" : "=v"(d) : "v"(a), "v"(b), "s"(kmask));
    return d;
#else
    return a ^ (b & kmask);
#endif
}

// (a << sh) | c as one instruction (v_lshlrev_b32 alone issues at the slow rate); sh must fold to a constant
VKL_FN uint32_t lshl_or(uint32_t a, uint32_t sh, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(sh), "v"(c));
    return d;
#else
    return (a << sh) | c;
#endif
}

// (a << sh) + c with c wave-uniform (a scalar register: the one scalar operand a three-operand instruction may have)
VKL_FN uint32_t lshl_add_s(uint32_t a, uint32_t sh, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(sh), "s"(c));
    return d;
#else
    return (a << sh) + c;
#endif
}

// (a << sh) + c as one instruction; sh must fold to a constant
VKL_FN uint32_t lshl_add(uint32_t a, uint32_t sh, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(sh), "v"(c));
    return d;
#else
    return (a << sh) + c;
#endif
}

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

// Bit 7 of some byte set?  (FASTQ is ASCII; the all-ASCII case takes a shorter path.)
VKL_FN bool has_non_ascii(const uint32_t d[16]) {
    uint32_t o = 0;
    for (int i = 0; i < 16; ++i) o |= d[i];
    return (o & 0x80808080u) != 0u;
}

// Phase A: classify the 64 bytes.  Returns the number of newlines in the block.
// ASCII = true may only be used when no byte of the block has bit 7 set.
template <bool ASCII>
VKL_FN uint32_t classify(const uint32_t d[16], LaneBits& o) {
    constexpr uint32_t kLutLo = 0x41204020u;  // low3 = 0..3 : inv, A(0x40|0), inv, C(0x40|1)
    constexpr uint32_t kLutHi = 0x42202053u;  // low3 = 4..7 : T(0x50|3), inv, inv, G(0x40|2)
    const uint32_t k7f = 0x7F7F7F7Fu;
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
        uint32_t C = 0, IV = 0, NL = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = T[j];
            // table lookup on the low 3 bits: expected high bits (case folded) | code
            const uint32_t L = perm(kLutHi, kLutLo, t & 0x07070707u);
            const uint32_t x = xor_and_k(L, t, 0xD8D8D8D8u);  // 0..3 for a base, else some bit of 0xFC
            // (the first field of each string is a plain two-operand AND: half the issue cost of v_and_or)
            C = j == 0 ? (x & 0x03030303u) : lshl_or(x & 0x03030303u, 2 * j, C);
            uint32_t nz, eq;
            if (ASCII) {
                // every byte of t and x is below 0x80, so plain byte-wise adds cannot carry out:
                nz = x + 0x7C7C7C7Cu;                             // bit 7 := x >= 4 (not a base)
                eq = xor_add_k(t, 0x75757575u, 0x01010101u);      // bit 7 := t == '\n' (0x0A ^ 0x75 = 0x7F)
            } else {
                // bit 7 := some bit of x & 0x7C set (carry trick), or bit 7 of x
                nz = ((x & 0x7C7C7C7Cu) + k7f) | x;
                // bit 7 := low 7 bits equal 0x0A and bit 7 of t clear
                eq = ~(xor_add_k(t & k7f, 0x0A0A0A0Au, k7f) | t);
            }
            IV = j == 0 ? ((nz >> 7) & 0x01010101u) : and_or_k(nz >> (7 - 2 * j), 0x01010101u << (2 * j), IV);
            NL = j == 0 ? ((eq >> 7) & 0x01010101u) : and_or_k(eq >> (7 - 2 * j), 0x01010101u << (2 * j), NL);
        }
        o.C[g] = C;
        o.IV[g] = IV;
        o.NL[g] = NL;
        c += popc(NL);
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
// (0 header, 1 sequence, 2 plus, 3 quality).  below(q) = ones_below(q), above(q) = ~below(q);
// the kernel serves both from LDS tables (one ds_read_b128 each).
// s_raw receives the unclamped start of the interval: 0 when the block begins inside a sequence
// line, p+1 (1..64) when the newline at position p -- the end of a header line -- opens one here,
// more than 64 when no sequence line starts in this block.
template <typename Below, typename Above>
VKL_FN Mask128 seq_mask_fast(const uint32_t NL[4], uint32_t lph, Below below, Above above, uint32_t& s_raw) {
    const uint32_t d = (1u - lph) & 3u;  // newlines to skip before a sequence line starts
    const uint32_t p1 = first_newline(NL);
    Mask128 m = above(umin(p1 + 1u, 64u));
    uint32_t r[4] = {NL[0] & m.w[0], NL[1] & m.w[1], NL[2] & m.w[2], NL[3] & m.w[3]};
    const uint32_t p2 = first_newline(r);
    m = above(umin(p2 + 1u, 64u));
    r[0] &= m.w[0]; r[1] &= m.w[1]; r[2] &= m.w[2]; r[3] &= m.w[3];
    const uint32_t p3 = first_newline(r);
    // interval [s, e): after the d-th newline, up to the (d+1)-th.  The five candidates
    // (-1, p1, p2, p3, 64) sit in consecutive bytes; a funnel shift by 8d picks the pair.
    const uint32_t lo = 0xFFu | (p1 << 8) | (p2 << 16) | (p3 << 24);
    const uint32_t pr = alignbit(64u, lo, 8u * d);
    const uint32_t s = ((pr & 0xFFu) + 1u) & 0xFFu;
    const uint32_t e = (pr >> 8) & 0xFFu;
    s_raw = s;
    const Mask128 ms = above(umin(s, 64u)), me = below(umin(e, 64u));
    Mask128 out;
    for (int g = 0; g < 4; ++g) out.w[g] = me.w[g] & ms.w[g];
    return out;
}

template <typename Below, typename Above>
VKL_FN Mask128 seq_mask_fast(const uint32_t NL[4], uint32_t lph, Below below, Above above) {
    uint32_t s_raw;
    return seq_mask_fast(NL, lph, below, above, s_raw);
}

// The same mask by COUNTING instead of searching (at most three newlines in the block): position p
// is in a sequence line iff the number of newlines at positions <= p equals d, the number of line
// ends still to pass at the block start.  The 2-bit slot of every position holds that count: four
// shift-adds per dword give the prefix sums inside a dword (a total of at most three never carries
// out of a slot), the newlines of the dwords before are added replicated into every slot, and one
// xor against d replicated leaves a zero slot exactly where the counts agree.  Pure VALU work with
// four independent dword chains -- no table lookups, so the piece loop never waits for LDS behind
// its own histogram atomics.  (A newline's own slot already counts it: the newline that ends a
// header line is marked, but it is not a base and BAD takes it out; callers that need sites
// mask with ~NL.)  Result on the even bits only.
VKL_FN uint32_t rep_slots(uint32_t v) {  // v = 0..3 replicated into all sixteen 2-bit slots
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t r = __umul24(v, 0x555555u);
    return r | (r << 8);
#else
    return v * 0x55555555u;
#endif
}

VKL_FN Mask128 seq_mask_count(const uint32_t NL[4], uint32_t lph) {
    const uint32_t want = rep_slots((1u - lph) & 3u);
    Mask128 out;
    uint32_t before = 0;  // newlines in the dwords before this one
    for (int g = 0; g < 4; ++g) {
        uint32_t x = NL[g];
        x += x << 2;
        x += x << 4;
        x += x << 8;
        x += x << 16;
        if (g) x += rep_slots(before);
        before += popc(NL[g]);
        const uint32_t z = x ^ want;
        out.w[g] = ~(z | (z >> 1)) & 0x55555555u;
    }
    return out;
}

// The same with room for a fourth newline (reads shorter than ~45 bases: a 64-byte block can hold
// the ends of all four lines of a record).  One more newline search and table lookup than
// seq_mask_fast, so the kernel only takes it when some block of the piece has exactly four.
// s_raw is as in seq_mask_fast and describes the FIRST interval only.
template <typename Below, typename Above>
VKL_FN Mask128 seq_mask_fast4(const uint32_t NL[4], uint32_t lph, Below below, Above above, uint32_t& s_raw) {
    const uint32_t d = (1u - lph) & 3u;
    const uint32_t p1 = first_newline(NL);
    Mask128 m = above(umin(p1 + 1u, 64u));
    uint32_t r[4] = {NL[0] & m.w[0], NL[1] & m.w[1], NL[2] & m.w[2], NL[3] & m.w[3]};
    const uint32_t p2 = first_newline(r);
    m = above(umin(p2 + 1u, 64u));
    r[0] &= m.w[0]; r[1] &= m.w[1]; r[2] &= m.w[2]; r[3] &= m.w[3];
    const uint32_t p3 = first_newline(r);
    m = above(umin(p3 + 1u, 64u));
    r[0] &= m.w[0]; r[1] &= m.w[1]; r[2] &= m.w[2]; r[3] &= m.w[3];
    const uint32_t p4 = first_newline(r);
    // candidates (-1, p1, p2, p3, p4, 64) in consecutive bytes; d = 0..3 picks the pair
    const uint32_t lo = 0xFFu | (p1 << 8) | (p2 << 16) | (p3 << 24);
    const uint32_t hi = p4 | (64u << 8);
    const uint32_t pr = alignbit(hi, lo, 8u * d);
    const uint32_t s = ((pr & 0xFFu) + 1u) & 0xFFu;
    const uint32_t e = (pr >> 8) & 0xFFu;
    s_raw = s;
    const Mask128 ms = above(umin(s, 64u)), me = below(umin(e, 64u));
    // A block that starts inside a sequence line (d = 0) and holds four newlines ends inside the
    // NEXT record's sequence line: a second interval from p4 + 1 on.
    const Mask128 m2 = above(d == 0u ? umin(p4 + 1u, 64u) : 64u);
    Mask128 out;
    for (int g = 0; g < 4; ++g) out.w[g] = (me.w[g] & ms.w[g]) | m2.w[g];
    return out;
}

// ---- read subsampling (vk_count_sampled_device) ------------------------------------------------
// A read is identified by the sample offset of its ANCHOR, the newline that ends its header line,
// and is counted iff sample_hash(seed, anchor) < threshold (threshold in [0, 2^32]): a pure function
// of the file's bytes, so every split of a sample into byte ranges selects the same reads.
VKL_FN uint32_t sample_hash(uint64_t seed, uint64_t anchor) {
    uint32_t h = static_cast<uint32_t>(anchor) ^ static_cast<uint32_t>(seed);
    h += static_cast<uint32_t>(anchor >> 32) * 0x9E3779B1u + static_cast<uint32_t>(seed >> 32);
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

VKL_FN bool sample_take(uint64_t seed, uint64_t anchor, uint64_t threshold) {
    return static_cast<uint64_t>(sample_hash(seed, anchor)) < threshold;
}

// Inclusion strings of one 64-byte block that starts at sample offset `base` in line phase lph, for
// any number of newlines: `first` = positions up to and including the first anchor (or all 64 when
// the block has none) -- they belong to the read the block was entered in; `inc` = positions after
// an anchor whose read is taken.  Returns anchors seen; last_take = decision of the last one.
VKL_FN uint32_t sample_strings_general(const uint32_t NL[4], uint32_t lph, uint64_t base, uint64_t seed,
                                       uint64_t threshold, uint32_t first[4], uint32_t inc[4], uint32_t& last_take,
                                       uint32_t& last_anchor) {
    uint32_t cur = lph & 3u, anchors = 0, take = 0;
    last_take = 0;
    last_anchor = 0;  // block position of the last anchor (meaningful when the return value is not 0)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int g = 0; g < 4; ++g) {
        uint32_t f = 0, n = 0;
        const uint32_t nl = NL[g];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll  // the rare path: keep it small, its registers would otherwise crowd the piece loop
#endif
        for (uint32_t b = 0; b < 32; b += 2) {
            if (anchors == 0) f |= 3u << b;
            else if (take) n |= 3u << b;
            if ((nl >> b) & 1u) {
                if (cur == 0u) {
                    ++anchors;
                    take = sample_take(seed, base + 16u * static_cast<uint32_t>(g) + (b >> 1), threshold) ? 1u : 0u;
                    last_take = take;
                    last_anchor = 16u * static_cast<uint32_t>(g) + (b >> 1);
                }
                cur = (cur + 1u) & 3u;
            }
        }
        first[g] = f;
        inc[g] = n;
    }
    return anchors;
}

// reformat.sh breaks reads longer than this into pieces before anything else sees them (the reference
// runs it with breaklength=500, commands/image.py:586-588): no k-mer of a subsample spans a multiple of
// 500 bases of its read.  Block positions p whose window would: ((r + p) mod 500) <= K - 2, r = read
// position of the block's first byte mod 500 -- at most two stretches in 64 positions.  Returns the
// bounds of [0, q1) and [lo2, hi2) as the mask tables take them.
constexpr uint32_t kBreakLength = 500;
template <int K>
VKL_FN void break_stretches(uint32_t r, uint32_t& q1, uint32_t& lo2, uint32_t& hi2) {
    q1 = r <= static_cast<uint32_t>(K - 2) ? static_cast<uint32_t>(K - 1) - r : 0u;
    lo2 = umin(kBreakLength - r, 64u);
    hi2 = umin(lo2 + static_cast<uint32_t>(K - 1), 64u);
}

VKL_FN Mask128 ones_not_below(uint32_t q) {
    Mask128 m = ones_below(q);
    for (int g = 0; g < 4; ++g) m.w[g] = ~m.w[g];
    return m;
}

// Any number of newlines (degenerate FASTQ with very short lines): position by position.
VKL_FN Mask128 seq_mask_general(const uint32_t NL[4], uint32_t lph) {
    Mask128 out;
    uint32_t cur = lph & 3u;
    for (int g = 0; g < 4; ++g) {
        uint32_t acc = 0;
        const uint32_t nl = NL[g];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll  // the rare path: unrolled, its sixteen shifted constants would sit in registers across the piece loop
#endif
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
// (after_group() runs after each 16-position group)
// the K-mer ending at p, read from the code string [ch | C[0..3]] (ch = codes of the 16
// positions before the block).  First base most significant.
template <int K, typename Emit, typename Hook>
VKL_FN void windows(uint32_t ch, const uint32_t C[4], const uint32_t ok[4], Emit emit, Hook after_group) {
    // Codes are packed with the EARLIEST position in the LOWEST bits, so a raw bit-field is
    // the k-mer with its first base least significant; the histogram convention wants the
    // first base most significant.  The kernel therefore counts into a bit-reversed index
    // space and un-reverses at flush time (see vk_count_kernel); here the raw field.
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    // OK bits are consumed from the top with x += x: the carry-out IS the lane predicate (one
    // v_add_co_u32 per position instead of and + compare).  OK lives on even bits; the odd bits
    // of w carry the same dword rotated by 8 positions, so that consecutive carries deliver
    // positions i-8 and i of one dword, whose fields lie 16 bits apart in ONE extracted word
    // (for K <= 7): one funnel shift serves two positions, the upper one through an SDWA and.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int g = 0; g < 4; ++g) {
        uint32_t w = ok[g] | (alignbit(ok[g], ok[g], 16u) << 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 15; i >= 8; --i) {
            // bit 31 first: odd bit 2i+1 = position (i + 8) % 16 = i - 8; then even bit 2i = position i
            const int plo = 16 * g + i - 8, phi = 16 * g + i;
            const int o = 30 + 2 * (plo - K + 1);  // bit offset of (field << 2) of plo in v[]
            const int word = o >> 5, sh = o & 31;
            if (2 * K + 2 <= 16) {
                uint32_t x;  // 32 bits from offset o: field(plo) << 2 at [0,16), field(phi) << 2 at [16,32)
                if (sh == 0) x = v[word];
                else if (word == 4) x = v[4] >> sh;  // the last fields end exactly at bit 160
                else x = alignbit(v[word + 1], v[word], static_cast<uint32_t>(sh));
                const bool take_lo = __builtin_add_overflow(w, w, &w);
                if (take_lo) emit(x & kMask4);
                const bool take_hi = __builtin_add_overflow(w, w, &w);
                if (take_hi) emit((x >> 16) & kMask4);
            } else {
                const int o2 = o + 16, word2 = o2 >> 5, sh2 = o2 & 31;
                const bool take_lo = __builtin_add_overflow(w, w, &w);
                if (take_lo) {
                    const uint32_t x = (sh + 2 * K + 2 <= 32) ? (v[word] >> sh)
                                                               : alignbit(v[word + 1], v[word], static_cast<uint32_t>(sh));
                    emit(x & kMask4);
                }
                const bool take_hi = __builtin_add_overflow(w, w, &w);
                if (take_hi) {
                    const uint32_t x = (sh2 + 2 * K + 2 <= 32) ? (v[word2] >> sh2)
                                                                : alignbit(v[word2 + 1], v[word2], static_cast<uint32_t>(sh2));
                    emit(x & kMask4);
                }
            }
        }
        after_group();  // 16 positions of every lane done (the bucket path drains its queues here)
    }
}

// ---- the sequence-only ("dense") stage of the K <= 7 count kernel --------------------------------
// Header, '+' and quality lines are 53 % of a FASTQ's bytes and need nothing but their newlines found.
// The piece loop therefore runs in two levels: a cheap LINE pass over all 64 bytes of a lane (newline
// flags -> ordered 64-bit mask -> the lane's stretch of sequence line), which picks the 16-byte
// GRANULES that hold sequence bytes, and the heavy stage (transposes, classification, window masks,
// histogram updates) on those granules only, one granule per lane, 64 granules per round.

VKL_FN uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {  // sum of the four byte products + c
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    uint32_t r = c;
    for (int i = 0; i < 4; ++i) r += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return r;
#endif
}

VKL_FN uint32_t ascii_or(const uint32_t d[16]) {  // bit 7 of some byte set <=> the block is not all ASCII
    uint32_t o = 0;
    for (int i = 0; i < 16; i += 2) o |= d[i] | d[i + 1];
    return o & 0x80808080u;
}

// Bit p of {hi, lo} set <=> byte p of the all-ASCII block is '\n'.  Per dword: the byte-wise add of
// the ASCII classifier puts "is newline" on bit 7, one AND isolates the flags (0x80 each), and one
// v_dot4_u32_u8 with the weights 1, 2, 4, 8 (16 .. 128 for the odd dword of a pair, accumulated)
// lays them down in position order: 8 ordered bits per two dwords, no transposes, no shifts.
VKL_FN void newline_mask64(const uint32_t d[16], uint32_t& lo, uint32_t& hi) {
    uint32_t v[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
        // (xor and add as ONE v_xad_u32: measured on MI355X, a three-operand op costs ~1.8 ns of a SIMD's issue
        // time at 8 waves per SIMD, two two-operand ops with literals ~2.1)
        const uint32_t f0 = xor_add_k(d[2 * k], 0x75757575u, 0x01010101u) & 0x80808080u;
        const uint32_t f1 = xor_add_k(d[2 * k + 1], 0x75757575u, 0x01010101u) & 0x80808080u;
        v[k] = udot4(f1, 0x80402010u, udot4(f0, 0x08040201u, 0u));  // (8 ordered flags) << 7
    }
    // (plain C on purpose: hipcc puts the wait states a v_dot4 result needs in front of its OWN instructions, not
    // in front of an asm statement that reads it -- the shifts fused by hand into v_lshl_or_b32 read stale masks)
    lo = (v[0] >> 7) | (v[1] << 1) | (v[2] << 9) | (v[3] << 17);
    hi = (v[4] >> 7) | (v[5] << 1) | (v[6] << 9) | (v[7] << 17);
}

VKL_FN uint32_t first_bit64(uint32_t lo, uint32_t hi) {  // index of the lowest set bit, 64 if none
    return umin(umin(ffbl(lo), ffbl(hi) | 32u), 64u);    // 0xFFFFFFFF | 32 stays above 64
}

// The lane's stretch [s, e] of positions whose line phase is 1 (e = the newline that ends the sequence
// line, 64 if it lies beyond the block; s = 64: none), from the newline mask, the newline count c and
// the line phase lph at the block start: the stretch begins behind the dn-th newline, dn = the line ends
// still to pass (0 .. 3), and ends at the next one.  A block that begins in a quality line and holds that
// line's end and a whole header line (a header under 64 bytes: every FASTQ whose records are not a
// multiple of 64 bytes has such blocks all over) is as ordinary as one that begins in a header.
// Returns false for what one stretch of granules with a start tag cannot describe: four or more newlines,
// or three with the stretch behind the second (reads shorter than ~20-45 bases, by the header's length);
// a granule (16 positions) that holds both the start of a sequence line (not at its first position) and
// the end of that line (reads under 15 bases); a granule that holds two line ends in front of the start
// (headers under 15 bytes).  The caller counts the windows of such a lane apart (vk_count.h, "lanes set
// aside").
// s_raw: the stretch's start before clamping -- 1 .. 64 exactly when a sequence line begins behind a newline of this
// block (that newline, at s_raw - 1, is its read's anchor), 0 when the block begins inside one, 65 when none begins.
VKL_FN bool seq_span(uint32_t lo, uint32_t hi, uint32_t c, uint32_t lph, uint32_t& s, uint32_t& e, uint32_t& s_raw) {
    const uint32_t dn = (1u - lph) & 3u;  // line ends to pass before a sequence line starts
    const uint32_t p1 = first_bit64(lo, hi);
    const uint32_t lo1 = lo & (lo - 1u), hi1 = lo ? hi : (hi & (hi - 1u));
    const uint32_t p2 = first_bit64(lo1, hi1);
    // the candidates (-1, p1, p2, 64, 64) in consecutive bytes; a funnel shift by 8 dn picks (start - 1, end)
    const uint32_t cand = 0xFFu | (p1 << 8) | (p2 << 16) | (64u << 24);
    const uint32_t pr = alignbit(64u, cand, 8u * dn);
    s_raw = (pr + 1u) & 0xFFu;
    s = umin(s_raw, 64u);
    e = (pr >> 8) & 0xFFu;
    const bool both = (s >> 4) == (e >> 4) && (s & 15u) != 0u;  // (s = 64: s & 15 == 0)
    // A tagged granule's sequence bytes are those behind its FIRST newline: with dn = 2 the line end before the one
    // the stretch starts behind (p1) must lie in an earlier granule.
    const bool two = dn == 2u && (s & 15u) != 0u && (p1 >> 4) == (p2 >> 4);
    return c <= (dn >= 2u ? 2u : 3u) && !both && !two;
}

VKL_FN bool seq_span(uint32_t lo, uint32_t hi, uint32_t c, uint32_t lph, uint32_t& s, uint32_t& e) {
    uint32_t s_raw;
    return seq_span(lo, hi, c, lph, s, e, s_raw);
}

// The granules of a lane that go to the heavy stage: every granule with a position of line phase 1,
// the newline that ends the line included -- so that where two granules follow each other in the
// heavy stage without being neighbours in the file, the first one's last position is never a base.
VKL_FN uint32_t span_first(uint32_t s) { return s >> 4; }
VKL_FN uint32_t span_last(uint32_t e) { return umin(e, 63u) >> 4; }
VKL_FN uint32_t span_count(uint32_t s, uint32_t e) { return s < 64u ? span_last(e) - span_first(s) + 1u : 0u; }

// A granule whose sequence line STARTS inside it (after the newline of a header line) travels with bit 7
// of its first byte set (the bytes are ASCII); every other granule's sequence bytes are those before
// its first newline, or all of them.
constexpr uint32_t kGranuleStartTag = 0x80u;
VKL_FN bool span_starts_inside(uint32_t s) { return s < 64u && (s & 15u) != 0u; }

// Heavy stage, one all-ASCII granule (start tag already taken off): codes, invalid flags and the
// sequence-byte mask in the 2-bit geometry (one dword each; SEQ on both bits of a position).
VKL_FN void classify_granule(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, bool starts_inside,
                             uint32_t& Cout, uint32_t& IVout, uint32_t& SEQout) {
    constexpr uint32_t kLutLo = 0x41204020u, kLutHi = 0x42202053u;  // as in classify()
    const uint32_t p01l = perm(a1, a0, 0x05010400u), p01h = perm(a1, a0, 0x07030602u);
    const uint32_t p23l = perm(a3, a2, 0x05010400u), p23h = perm(a3, a2, 0x07030602u);
    const uint32_t T[4] = {perm(p23l, p01l, 0x05040100u), perm(p23l, p01l, 0x07060302u),
                           perm(p23h, p01h, 0x05040100u), perm(p23h, p01h, 0x07060302u)};
    uint32_t C = 0, IV = 0, NL = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = T[j];
        const uint32_t L = perm(kLutHi, kLutLo, t & 0x07070707u);
        const uint32_t x = xor_and_k(L, t, 0xD8D8D8D8u);
        C = j == 0 ? (x & 0x03030303u) : lshl_or(x & 0x03030303u, 2 * j, C);
        const uint32_t nz = x + 0x7C7C7C7Cu;
        const uint32_t eq = xor_add_k(t, 0x75757575u, 0x01010101u);
        IV = j == 0 ? ((nz >> 7) & 0x01010101u) : and_or_k(nz >> (7 - 2 * j), 0x01010101u << (2 * j), IV);
        NL = j == 0 ? ((eq >> 7) & 0x01010101u) : and_or_k(eq >> (7 - 2 * j), 0x01010101u << (2 * j), NL);
    }
    const uint32_t low = (NL - 1u) & ~NL;  // everything below the first newline (all ones without one)
    Cout = C;
    IVout = IV;
    SEQout = starts_inside ? ~((low << 2) | 3u) : low;
}

// ---- round 6: the line pass tells the heavy stage where a granule's sequence bytes are ------------------------
// The line pass knows every lane's stretch [s, e] already; classify_granule() found it again from the granule's own
// newline flags (four v_xad, four shifts, four v_and_or, the mask below the first newline and a select: 21 of the
// heavy stage's 116 vector instructions per round).  Now the two EDGE granules of a stretch carry a 5-bit note in
// bit 7 of their first five bytes (the bytes are ASCII: bit 7 is free; the classification below does not look at it):
//   bytes 0..3  p, a position 0..15, one bit per byte (bit i of p on byte i)
//   byte 4      kGranuleEnd: the sequence bytes are the positions BELOW p (p = the newline that ends the line);
//               clear: the positions FROM p on (p = s & 15, the first base; 0 = all sixteen: a granule without a
//               note is all sequence).
// A separator granule (sixteen newlines) needs no note: a newline is not a base.
constexpr uint32_t kGranuleNoteBits = 0x80808080u;   // dword 0: p
constexpr uint32_t kGranuleEnd = 0x80u;              // dword 1

// bit i of p (0..15) on bit 7 of byte i
VKL_FN uint32_t note_spread(uint32_t p) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t;   // (spelled out: hipcc folds the shift into the factor and takes the quarter-rate v_mul_lo_u32 for it)
    asm("v_mul_u32_u24 %0, 0x204081, %1" : "=v"(t) : "v"(p));
    return (t & 0x01010101u) << 7;
#else
    return ((p * 0x204081u) & 0x01010101u) << 7;
#endif
}

// One note per lane: a stretch may start inside a granule, or end in the lane, not both.
VKL_FN bool span_one_note(uint32_t s, uint32_t e) { return (s & 15u) == 0u || e >= 64u; }

// seq_span() for the note scheme: the same stretch [s, e], and what one note per lane can describe -- seq_span's two
// granule rules (start and end in one granule; two line ends in front of a start) were the tag's limits and are
// gone: the note names the position itself.  What stays: at most three newlines (two when two or three line ends
// must pass first: the candidates stop at the second newline), and span_one_note.
// dn: the line ends to pass before a sequence line starts, (1 - line phase) & 3 (the caller has it one subtraction sooner
// than the phase).
VKL_FN bool seq_span_note(uint32_t lo, uint32_t hi, uint32_t c, uint32_t dn, uint32_t& s, uint32_t& e, uint32_t& s_raw) {
    const uint32_t p1 = first_bit64(lo, hi);
    const uint32_t lo1 = lo & (lo - 1u), hi1 = lo ? hi : (hi & (hi - 1u));
    const uint32_t p2 = first_bit64(lo1, hi1);
    const uint32_t cand = 0xFFu | (p1 << 8) | (p2 << 16) | (64u << 24);
    const uint32_t pr = alignbit(64u, cand, 8u * dn);
    s_raw = (pr + 1u) & 0xFFu;
    s = umin(s_raw, 64u);
    e = (pr >> 8) & 0xFFu;
    // c <= (dn >= 2 ? 2 : 3), as one comparison: 2 c + dn <= 7
    return 2u * c + dn <= 7u && span_one_note(s, e);
}

// Heavy stage, one all-ASCII granule with its note: codes (2-bit geometry) and BAD -- not a base of the sequence
// line -- on the EVEN bits (the odd bits hold garbage: every user shifts by whole positions and masks at the end).
// seq (both bits of a position, for the read index): the positions the note calls sequence bytes.
VKL_FN void classify_granule_note(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& Cout, uint32_t& BADout, uint32_t& SEQout) {
    constexpr uint32_t kLutLo = 0x41204020u, kLutHi = 0x42202053u;  // as in classify()
    const uint32_t p01l = perm(a1, a0, 0x05010400u), p01h = perm(a1, a0, 0x07030602u);
    const uint32_t p23l = perm(a3, a2, 0x05010400u), p23h = perm(a3, a2, 0x07030602u);
    const uint32_t T[4] = {perm(p23l, p01l, 0x05040100u), perm(p23l, p01l, 0x07060302u),
                           perm(p23h, p01h, 0x05040100u), perm(p23h, p01h, 0x07060302u)};
    uint32_t C = 0, IV = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = T[j];
        const uint32_t L = perm(kLutHi, kLutLo, t & 0x07070707u);
        const uint32_t x = xor_and_k(L, t, 0x58585858u);   // (bit 7 left out: the note; x < 0x80, the add below never carries)
        C = j == 0 ? (x & 0x03030303u) : lshl_or(x & 0x03030303u, 2 * j, C);
        const uint32_t nz = x + 0x7C7C7C7Cu;
        IV = j == 0 ? ((nz >> 7) & 0x01010101u) : and_or_k(nz >> (7 - 2 * j), 0x01010101u << (2 * j), IV);
    }
    const uint32_t m = udot4(a0 & kGranuleNoteBits, 0x20100804u, 0u);   // p << 9
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t S;   // all ones: an end granule (one v_bfe_i32; hipcc makes it v_bfe_u32 + v_add)
    asm("v_bfe_i32 %0, %1, 7, 1" : "=v"(S) : "v"(a1));
#else
    const uint32_t S = static_cast<uint32_t>(static_cast<int32_t>(a1 << 24) >> 31);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // positions from p on: ~0 << 2p, the shift count being byte 1 of m (one SDWA shift instead of a shift and a shift).  By
    // hand, so the wait states a v_dot4 result needs in front of its reader are written out (tools/asm_lint.py: DOT, 3).
    uint32_t X;
    asm("s_nop 2\n\tv_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
        : "=v"(X) : "v"(m), "v"(0xFFFFFFFFu));
#else
    const uint32_t X = 0xFFFFFFFFu << ((m >> 8) & 31u);                  // positions from p on
#endif
    Cout = C;
    SEQout = X ^ S;
    BADout = IV | ~(X ^ S);
}

// The same for any bytes (no ASCII precondition), with the newline flags instead of a sequence mask: the subsample
// walker (vk_ladder.h) knows where its read starts and only asks where it ends.
// ASCII: the caller has seen that no byte of the wave's granules has bit 7 set (the byte tests of classify_granule).
template <bool ASCII>
VKL_FN void classify_granule_nl(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& Cout, uint32_t& IVout, uint32_t& NLout) {
    constexpr uint32_t kLutLo = 0x41204020u, kLutHi = 0x42202053u;  // as in classify()
    const uint32_t k7f = 0x7F7F7F7Fu;
    const uint32_t p01l = perm(a1, a0, 0x05010400u), p01h = perm(a1, a0, 0x07030602u);
    const uint32_t p23l = perm(a3, a2, 0x05010400u), p23h = perm(a3, a2, 0x07030602u);
    const uint32_t T[4] = {perm(p23l, p01l, 0x05040100u), perm(p23l, p01l, 0x07060302u),
                           perm(p23h, p01h, 0x05040100u), perm(p23h, p01h, 0x07060302u)};
    uint32_t C = 0, IV = 0, NL = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        const uint32_t t = T[j];
        const uint32_t L = perm(kLutHi, kLutLo, t & 0x07070707u);
        const uint32_t x = xor_and_k(L, t, 0xD8D8D8D8u);
        C = j == 0 ? (x & 0x03030303u) : lshl_or(x & 0x03030303u, 2 * j, C);
        const uint32_t nz = ASCII ? x + 0x7C7C7C7Cu : (((x & 0x7C7C7C7Cu) + k7f) | x);
        const uint32_t eq = ASCII ? xor_add_k(t, 0x75757575u, 0x01010101u) : ~(xor_add_k(t & k7f, 0x0A0A0A0Au, k7f) | t);
        IV = j == 0 ? ((nz >> 7) & 0x01010101u) : and_or_k(nz >> (7 - 2 * j), 0x01010101u << (2 * j), IV);
        NL = j == 0 ? ((eq >> 7) & 0x01010101u) : and_or_k(eq >> (7 - 2 * j), 0x01010101u << (2 * j), NL);
    }
    Cout = C;
    IVout = IV;
    NLout = NL;
}

VKL_FN void classify_granule_any(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& Cout, uint32_t& IVout, uint32_t& NLout) {
    classify_granule_nl<false>(a0, a1, a2, a3, Cout, IVout, NLout);
}

// OK of the 16 positions of one granule (even bits) from its BAD string and the one before it.
template <int K>
VKL_FN uint32_t ok_mask1(uint32_t badh, uint32_t bad) {
    uint32_t w0 = badh, w1 = bad;
    int cover = 1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int step = 0; step < 4; ++step) {
        if (cover < K) {
            const int shp = cover < K - cover ? cover : K - cover;
            const uint32_t sh = 2u * static_cast<uint32_t>(shp);
            w1 |= alignbit(w1, w0, 32u - sh);
            w0 |= w0 << sh;
            cover += shp;
        }
    }
    return ~w1 & 0x55555555u;
}

// windows() for one granule: code string [ch | C]
template <int K, typename Emit>
VKL_FN void windows1(uint32_t ch, uint32_t C, uint32_t ok, Emit emit) {
    const uint64_t v = (static_cast<uint64_t>(C) << 32) | ch;
    for (int p = 0; p < 16; ++p)
        if ((ok >> (2 * p)) & 1u) emit(static_cast<uint32_t>((v >> (32 + 2 * (p - K + 1))) & ((1u << (2 * K)) - 1u)) << 2);
}

}  // namespace vkl
#endif  // VK_LANE_H
