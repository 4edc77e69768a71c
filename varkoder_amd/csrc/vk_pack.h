// vk_pack.h -- FASTQ text -> the packed sequence stream (2-bit codes + per-position masks)
// Part of the one translation unit vkimg.hip (device code for gfx950; see the notes there).
//
// Counting at k = 8, 9 and counting subsamples need the FASTQ's sequence lines several times over or next to a
// partition stage that has no registers to spare for line logic; 53 % of the text is header, '+' and quality
// lines.  vk_pack_kernel therefore streams the text ONCE, with the k <= 7 kernel's front end (line pass over all
// bytes, the 16-byte granules that hold sequence bytes handed through the wave's exchange buffer, one granule
// per lane classified per round, lanes the line pass cannot describe set aside), and writes, per granule that
// holds sequence bytes, one RECORD of the packed stream:
//     C  u32  the 16 positions' 2-bit codes (first base lowest; garbage where the byte is not a base)
//     M  u32  even bits: BAD  -- the position is not a base of a sequence line (N, other letters, newline,
//                               the bytes of the other lines that share the granule)
//             odd bits:  SITE -- the position is a byte of a sequence line other than its newline
// in file order, 8 bytes for 16 bytes of text that matter: 0.26 of the text's size for 150-base reads.
// The stream is K-independent: a consumer takes a record's K - 1 bases of context from the record before it in
// the stream.  Where two records follow each other without being neighbours in the file the first one's last
// position is BAD (a granule is packed as soon as ONE of its positions has line phase 1, the newline that ends
// the line included; the general path packs a group when it has a base or the next group begins with one), so
// no window reaches across.
// Every wavefront packs its own byte range (the ranges of vk_count_kernel) into its own segment of the
// sample's record arrays; a range that does not begin at the sample's start opens with one record of context
// (the 16 positions before the range: flag kPackHasPre in the wave's count word) whose own windows belong to
// the wave before.
#ifndef VK_PACK_H
#define VK_PACK_H

#include "vk_count.h"

namespace {

// The general path of one piece for the pack kernel (first and last piece of a range, bytes >= 0x80): all 64
// bytes of every lane classified (vk_count_kernel's front end), the groups that matter stored behind the
// records already written.  Returns the new (pph, records written) -- wave-uniform.
struct PackGeneral {
    uint32_t pph, nrec;
};

__device__ __attribute__((noinline)) PackGeneral pack_general_piece(uint4 q0, uint4 q1, uint4 q2, uint4 q3, uint32_t pph_in, uint32_t nrec,
                                                                    uint32_t flags, uint32_t ph0, uint32_t* pc, uint32_t* pm) {
    const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const bool first = (flags & 1u) != 0u, has_pre = (flags & 2u) != 0u;
    const uint32_t lane = lane_now();
    uint32_t pph = pph_in;
    vkl::LaneBits lb;
    const uint32_t c = __any(vkl::has_non_ascii(d)) ? vkl::classify<false>(d, lb) : vkl::classify<true>(d, lb);
    const uint32_t incl = wave_inclusive_sum(c);
    const uint32_t total = lane_bcast(incl, 63);
    if (first) pph = has_pre ? ph0 - lane_bcast(c, 0) : 0u;
    const uint32_t lph = (pph + incl - c) & 3u;
    vkl::Mask128 seq;
    const bool degenerate = __any(c > 4u);
    uint32_t s_raw = 0;
    const bool four = !degenerate && __any(c > 3u);
    if (degenerate) seq = vkl::seq_mask_general(lb.NL, lph);
    else if (four) seq = vkl::seq_mask_fast4(lb.NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw);
    else seq = vkl::seq_mask_count(lb.NL, lph);
    uint32_t bad[4], m[4];
    vkl::bad_mask(lb, seq, bad);
#pragma unroll
    for (int g = 0; g < 4; ++g) m[g] = bad[g] | (((seq.w[g] & ~lb.NL[g]) & 0x55555555u) << 1);
    // a group is packed when it has a base, or when the group behind it begins with one (it is that group's context)
    const uint32_t next0 = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(1, static_cast<int>(bad[0] & 1u), 0x130, 0xF, 0xF, false));  // wave_shl:1; lane 63: "bad", packed anyway below
    uint32_t emit[4];
    emit[0] = (bad[0] != 0x55555555u || (bad[1] & 1u) == 0u) ? 1u : 0u;
    emit[1] = (bad[1] != 0x55555555u || (bad[2] & 1u) == 0u) ? 1u : 0u;
    emit[2] = (bad[2] != 0x55555555u || (bad[3] & 1u) == 0u) ? 1u : 0u;
    emit[3] = (bad[3] != 0x55555555u || (next0 & 1u) == 0u || lane == 63u) ? 1u : 0u;   // the piece's last group always: the next piece cannot be asked
    if (first && has_pre && lane == 0u) {   // the pre-block: its last group only, as the range's record of context
        emit[0] = 0u; emit[1] = 0u; emit[2] = 0u; emit[3] = 1u;
    }
    const uint32_t n = emit[0] + emit[1] + emit[2] + emit[3];
    const uint32_t inc = wave_inclusive_sum(n);
    uint32_t at = nrec + inc - n;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (emit[g]) {
            pc[at] = lb.C[g];
            pm[at] = m[g];
            ++at;
        }
    }
    PackGeneral o;
    o.pph = pph + total;
    o.nrec = nrec + lane_bcast(inc, 63);
    return o;
}

__global__ __launch_bounds__(kCountThreads, VK_K1_OCC) void vk_pack_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ lens, uint32_t nsamples,
    uint32_t parts, PackParams pk, uint32_t* __restrict__ wavephase, uint32_t* __restrict__ aside, uint32_t aside_cap,
    uint32_t* __restrict__ aside_n) {
    __shared__ uint4 xbuf[kWaves][64];     // per wave: 64 granules on their way to the classifying round

    const uint32_t unit = blockIdx.x;
    const uint32_t smp = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const WaveRange wr = wave_range(lens[smp], parts, part, wave);
    uint32_t ph_start = 0, ph_end = 0, aside_count = 0, records = 0;
    if (!wr.empty) {
        uint4* const xb = &xbuf[wave][0];
        const uint8_t* sbase = reinterpret_cast<const uint8_t*>(uniform64(reinterpret_cast<uint64_t>(fastq + offs[smp])));
        const uint64_t len = uniform64(lens[smp]), w0 = uniform64(wr.w0), w1 = uniform64(wr.w1);
        const uint32_t ph0 = (w0 != 0) ? sync_phase(sbase, w0, len, reinterpret_cast<uint64_t*>(xb), lane) : 0u;
        ph_start = ph0;
        const bool has_pre = w0 != 0;
        const uint64_t o0 = has_pre ? w0 - 64 : 0;
        const uint64_t span = w1 - o0;
        const uint32_t npieces = static_cast<uint32_t>((span + kPiece - 1) / kPiece);
        const uint32_t tail_bytes = static_cast<uint32_t>(span % kPiece);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(sbase + o0), 0, static_cast<int>((span + 15) & ~15ull), 0x00020000);
        const uint64_t seg = uniform64(pack_segment(pk, smp, part, wave, w0));
        uint32_t* const pc = pk.c + seg;
        uint32_t* const pm = pk.m + seg;
        uint4 r0, r1, r2, r3;
        auto load_piece = [&](uint32_t piece) {
            const uint32_t soff = piece * static_cast<uint32_t>(kPiece);
            const uint32_t lane64 = lane_now() << 6;
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 16u, 0);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 32u, 0);
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 48u, 0);
            r0 = make_uint4(a.x, a.y, a.z, a.w);
            r1 = make_uint4(b.x, b.y, b.z, b.w);
            r2 = make_uint4(c.x, c.y, c.z, c.w);
            r3 = make_uint4(d.x, d.y, d.z, d.w);
        };
        auto clip_granule = [](uint4& v, int n) {
            if (n >= 16) return;
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int kb = n - 4 * d;
                w[d] &= kb >= 4 ? 0xFFFFFFFFu : (kb <= 0 ? 0u : ((1u << (8 * kb)) - 1u));
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        };

        uint32_t pph = 0;     // line phase at the start of the current piece
        uint32_t npend = 0;   // granules waiting in xb[0 .. npend), < 64 between pieces
        uint32_t nrec = 0;    // records written
        uint32_t* const alist = aside + (static_cast<uint64_t>(unit) * kWaves + static_cast<uint32_t>(wave)) * aside_cap;
        uint32_t naside = 0;
        bool aside63 = false;

        // one granule per lane (the first n lanes) -> one record each
        auto round_pack = [&](uint32_t n, uint4 q) __attribute__((always_inline)) {
            uint32_t C, IV, SEQ;
            vkl::classify_granule(q.x & ~vkl::kGranuleStartTag, q.y, q.z, q.w, (q.x & vkl::kGranuleStartTag) != 0u, C, IV, SEQ);
            const uint32_t M = ((IV | ~SEQ) & 0x55555555u) | (SEQ & 0xAAAAAAAAu);
            const uint32_t ln = lane_now();
            if (ln < n) {
                pc[nrec + ln] = C;
                pm[nrec + ln] = M;
            }
            nrec += n;
        };
        auto flush = [&]() __attribute__((always_inline)) {
            uint4 q = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
            const uint32_t ln = lane_now();
            if (ln < npend) q = xb[ln];
            round_pack(npend, q);
            npend = 0u;
        };

        load_piece(0);
        for (uint32_t it = 0; it < npieces; ++it) {
            if (it + 1 == npieces && (tail_bytes & 15u) != 0u) {
                uint32_t tb = tail_bytes;
                asm volatile("" : "+s"(tb));
                const int n = static_cast<int>(tb) - static_cast<int>(lane_now() << 6);
                clip_granule(r0, n);
                clip_granule(r1, n - 16);
                clip_granule(r2, n - 32);
                clip_granule(r3, n - 48);
            }
            const uint4 q0 = r0, q1 = r1, q2 = r2, q3 = r3;
            const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                    q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            // ---- line pass (vk_count_dense_kernel's) ----
            bool fast = it != 0 && it + 1 != npieces && !__any(vkl::ascii_or(d) != 0u);
            uint32_t total = 0, s = 64, e = 64;
            if (fast) {
                uint32_t mlo, mhi;
                vkl::newline_mask64(d, mlo, mhi);
                const uint32_t c = vkl::popc(mlo) + vkl::popc(mhi);
                const uint32_t incl = wave_inclusive_sum(c);
                total = lane_bcast(incl, 63);
                const uint32_t lph = (pph + incl - c) & 3u;
                const bool plain = vkl::seq_span(mlo, mhi, c, lph, s, e);
                const unsigned long long am = __ballot(!plain);
                if (am != 0ull) {   // rare: lanes set aside (vk_count.h), or too many of them
                    const uint32_t na = static_cast<uint32_t>(__builtin_popcountll(am));
                    if (na > kSetAside || naside + na > aside_cap) {
                        fast = false;
                    } else {
                        if (!plain) {
                            const uint32_t ln = lane_now();
                            const bool before = ln == 0u ? aside63 : ((am >> (ln - 1u)) & 1ull) != 0ull;
                            alist[naside + static_cast<uint32_t>(__builtin_popcountll(am & ((1ull << ln) - 1ull)))] =
                                ((it * 64u + ln) << 3) | (before ? 4u : 0u) | lph;
                            r0 = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);   // the separator
                            s = 0u;
                            e = 15u;
                        }
                        naside += na;
                    }
                }
                if (fast) aside63 = (am >> 63) != 0ull;
            }
            if (!fast) {
                aside63 = false;
                if (npend != 0u) flush();
                const PackGeneral pg = pack_general_piece(q0, q1, q2, q3, pph, nrec, (it == 0 ? 1u : 0u) | (has_pre ? 2u : 0u), ph0, pc, pm);
                pph = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pg.pph)));
                nrec = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pg.nrec)));
                if (it + 1 < npieces) load_piece(it + 1);
                continue;
            }
            // ---- the granules with sequence bytes, 64 at a time, through the exchange buffer ----
            const uint32_t gs = vkl::span_first(s), n = vkl::span_count(s, e);
            const uint32_t incl = wave_inclusive_sum(n);
            const uint32_t tot = npend + lane_bcast(incl, 63);
            const uint32_t first = npend + incl - n - gs;
            uint32_t wp[4];
#pragma unroll
            for (uint32_t g = 0; g < 4; ++g) wp[g] = (g - gs < n) ? first + g : 0xFFFFFFC0u;
            const uint32_t wtag = vkl::span_starts_inside(s) ? first + gs : 0xFFFFFFC0u;
            const uint32_t rounds = tot >> 6;
            auto put = [&](uint32_t r) __attribute__((always_inline)) {
                if ((wp[0] >> 6) == r) xb[wp[0] & 63u] = r0;
                if ((wp[1] >> 6) == r) xb[wp[1] & 63u] = r1;
                if ((wp[2] >> 6) == r) xb[wp[2] & 63u] = r2;
                if ((wp[3] >> 6) == r) xb[wp[3] & 63u] = r3;
                if ((wtag >> 6) == r) atomicOr(&xb[wtag & 63u].x, vkl::kGranuleStartTag);
            };
            if (rounds >= 2u) {
                for (uint32_t r = 0; r + 2u < rounds; ++r) {
                    put(r);
                    const uint4 q = xb[lane];
                    round_pack(64u, q);
                }
                put(rounds - 2u);
                const uint4 qa = xb[lane];
                put(rounds - 1u);
                const uint4 qb = xb[lane];
                put(rounds);
                load_piece(it + 1);
                round_pack(64u, qa);
                round_pack(64u, qb);
            } else if (rounds == 1u) {
                put(0u);
                const uint4 q = xb[lane];
                put(1u);
                load_piece(it + 1);
                round_pack(64u, q);
            } else {
                put(0u);
                load_piece(it + 1);
            }
            npend = tot & 63u;
            pph += total;
        }
        // (the last piece of a range takes the general path: nothing is pending here)
        ph_end = pph & 3u;
        aside_count = naside;
        records = nrec | (has_pre ? kPackHasPre : 0u);
    }
    if (lane == 0) {
        wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));
        aside_n[unit * kWaves + wave] = aside_count;
        pk.count[unit * kWaves + wave] = records;
    }
}

}  // namespace

#endif  // VK_PACK_H
