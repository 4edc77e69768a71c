// vk_ladder.h -- subsampled counts without streaming the text once per subsample: a read index per sample
// (vk_index_kernel) and a walker over the reads a subsample takes (vk_walk_kernel).
// Part of the one translation unit vkimg.hip (device code for gfx950; see the notes there).
//
// The reference draws up to ~12 subsamples of a cleaned file (split_fastq's 1-2-5 ladder, commands/image.py:
// 682-695, one `reformat.sh samplebasestarget=...` per size, :577-627) and counts each with dsk.  Together the
// ladder takes about a quarter of the reads, but a count kernel that streams the text pays for every byte of it
// once per subsample (rounds 2-3: 12 passes at 0.40 of HBM peak -- 18x the plain count).
//  * vk_index_kernel streams the text ONCE per sample (a line pass only: no classification, no windows): every
//    newline that ends a header line is an ANCHOR -- the identity of its read, the thing sample_hash takes --
//    and goes into the sample's anchor list (u32 offsets, per-wavefront segments); the bytes of all sequence
//    lines (`nsites`, which the ladder's thresholds need, image.py:669-675) fall out as the sum of the positions
//    of the line ends behind the anchors minus the anchors'.
//  * vk_walk_kernel, one workgroup per (subsample, part): its wavefronts go through the anchor list, 64 anchors
//    at a time, keep the reads the subsample takes (sample_hash(seed, anchor) < threshold: the rule of
//    vk_count_sampled_device), and when 64 of them have come together they are WALKED from their anchors to their
//    lines' ends, sixteen reads at a time, a quad of lanes per read, a 64-byte sector per step: the count kernels'
//    granule classification and window blocks on the reads' own bytes; no window over a byte that is not a base or
//    over a multiple of 500 bases of the read (reformat.sh's breaklength=500, image.py:586-588); histogram in LDS for
//    k <= 7, the subsample's row in HBM for k = 8, 9.  Work and traffic are those of the reads taken; whatever the
//    text looks like (any line lengths, CRLF, a last line without newline) the walk is exact.
#ifndef VK_LADDER_H
#define VK_LADDER_H

#include "vk_count.h"

namespace {

__global__ __launch_bounds__(kCountThreads, VK_K1_OCC) void vk_index_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ lens, uint32_t nsamples,
    uint32_t parts, IndexParams ip, uint32_t* __restrict__ wavephase) {
    __shared__ uint64_t scratch[kWaves][8];   // sync_phase: newline positions after a range start
    const uint32_t unit = blockIdx.x;
    const uint32_t smp = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const WaveRange wr = wave_range(lens[smp], parts, part, wave);
    uint32_t ph_start = 0, ph_end = 0, nanch = 0;
    if (!wr.empty) {
        const uint8_t* sbase = reinterpret_cast<const uint8_t*>(uniform64(reinterpret_cast<uint64_t>(fastq + offs[smp])));
        const uint64_t len = uniform64(lens[smp]), w0 = uniform64(wr.w0), w1 = uniform64(wr.w1);
        const uint32_t ph0 = (w0 != 0) ? sync_phase(sbase, w0, len, &scratch[wave][0], lane) : 0u;
        ph_start = ph0;
        const uint64_t span = w1 - w0;
        const uint32_t npieces = static_cast<uint32_t>((span + kPiece - 1) / kPiece);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        // (bytes at and beyond w1 read as zeros: the descriptor ends at w1, rounded up to a dword -- zeros are no newlines)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(sbase + w0), 0, static_cast<int>((span + 3) & ~3ull), 0x00020000);
        const uint32_t tail = static_cast<uint32_t>(span & 3u);   // bytes of the last dword that belong to the range (0: all)
        uint32_t* const seg = ip.anchors + uniform64(index_segment(ip, smp, w0, part, wave));
        const uint32_t cap = static_cast<uint32_t>(index_segment_cap(span));
        uint32_t pph = ph0;
        bool full = false;
        long long acc = 0;   // per lane: positions of the sequence lines' ends - positions of their anchors - 1
        uint4 r0, r1, r2, r3;
        auto load_piece = [&](uint32_t piece) {
            const uint32_t soff = piece * static_cast<uint32_t>(kPiece);
            const uint32_t lane64 = static_cast<uint32_t>(lane) << 6;
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 16u, 0);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 32u, 0);
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff + 48u, 0);
            r0 = make_uint4(a.x, a.y, a.z, a.w);
            r1 = make_uint4(b.x, b.y, b.z, b.w);
            r2 = make_uint4(c.x, c.y, c.z, c.w);
            r3 = make_uint4(d.x, d.y, d.z, d.w);
        };
        load_piece(0);
        for (uint32_t it = 0; it < npieces; ++it) {
            uint32_t d[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
            if (it + 1 < npieces) load_piece(it + 1);
            if (it + 1 == npieces && tail != 0u) {
                // the range ends inside a dword (the sample's last bytes): what the load brought of the bytes behind it does not count
                const uint32_t last = static_cast<uint32_t>((span - 1u) & (kPiece - 1u)) >> 2;   // that dword's index in the piece
                const uint32_t keep = (1u << (8u * tail)) - 1u;
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i)
                    if (static_cast<uint32_t>(lane) * 16u + i == last) d[i] &= keep;
            }
            uint32_t mlo, mhi;
            newline_mask64_any(d, mlo, mhi);
            const uint32_t c = vkl::popc(mlo) + vkl::popc(mhi);
            const uint32_t incl = wave_inclusive_sum(c);
            const uint32_t total = lane_bcast(incl, 63);
            const uint32_t lph = (pph + incl - c) & 3u;   // phase of the line the lane's first byte lies in
            // sample offset of the lane's block
            const uint64_t blk = w0 + static_cast<uint64_t>(it) * kPiece + static_cast<uint32_t>(lane) * 64u;
            index_block(mlo, mhi, lph, blk, true, seg, cap, nanch, full, acc);
            pph += total;
        }
        ph_end = pph & 3u;
        // a sample that ends inside a sequence line (no final newline there): the line ends at the end of the text
        if (w1 == len && ph_end == 1u && lane == 0) acc += static_cast<long long>(len);
        // per-wave sum -> the sample's site count
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const uint32_t lo = __shfl_xor(static_cast<uint32_t>(acc), sft), hi = __shfl_xor(static_cast<uint32_t>(static_cast<unsigned long long>(acc) >> 32), sft);
            acc += static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
        }
        if (lane == 0) {
            atomicAdd(&ip.sites[smp], static_cast<unsigned long long>(acc));
            if (full || len >= (1ull << 32)) atomicOr(&ip.overflow[smp], 1u);
        }
    }
    if (lane == 0) {
        wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));
        ip.count[unit * kWaves + wave] = nanch;
    }
}

// ---- the walker ---------------------------------------------------------------------------------------
struct WalkParams {
    const uint32_t* isample;      // [npairs] the pair's sample in the index
    const uint64_t* seeds;        // [npairs]
    const uint64_t* thresholds;   // [npairs], in [0, 2^32]
    unsigned long long* sites;    // [npairs][2]: bytes of all sequence lines (from the index), of the reads taken; may be null
    uint32_t iparts;              // workgroups per sample of the index launch (the walker's grid uses the same parts)
};

constexpr uint32_t kWalkQueue = 128;   // anchors a wave's queue of taken reads holds

template <int K>
__global__ __launch_bounds__(kCountThreads, VK_K1_OCC) void vk_walk_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs, const uint64_t* __restrict__ lens, uint32_t npairs,
    IndexParams ip, WalkParams wp, uint32_t* __restrict__ hist_out, int atomic_flush) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr bool LDSH = NCODE <= kMaxBins;
    __shared__ uint32_t hist[LDSH ? NCODE : 1];     // raw-field order (first base least significant), as in vk_count_kernel
    __shared__ uint32_t queue[kWaves][kWalkQueue];
    const uint32_t unit = blockIdx.x;
    const uint32_t pair = unit / wp.iparts, part = unit % wp.iparts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (LDSH) {
        for (uint32_t i = tid; i < NCODE; i += kCountThreads) hist[i] = 0u;
        __syncthreads();
    }
    const uint32_t hist_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)hist));
    const uint32_t smp = wp.isample[pair];
    const uint8_t* sbase = fastq + offs[pair];
    const uint64_t len = lens[pair];
    const WaveRange wr = wave_range(len, wp.iparts, part, wave);
    uint32_t* const out = hist_out + static_cast<uint64_t>(pair) * NCODE;
    if (!wr.empty) {
        const uint32_t* seg = ip.anchors + index_segment(ip, smp, wr.w0, part, wave);
        const uint32_t n = ip.count[(smp * wp.iparts + part) * kWaves + static_cast<uint32_t>(wave)];
        const uint64_t seed = wp.seeds[pair], thr = wp.thresholds[pair];
        uint32_t* const q = &queue[wave][0];
        uint32_t npend = 0;
        uint32_t taken_sites = 0;   // per lane
        // The reads behind the anchors q[0 .. nb) are walked sixteen at a time, FOUR LANES PER READ: a quad takes the read's
        // text in 64-byte sectors -- lane g of the quad the g-th 16-byte granule, so that a sector is fetched once, by one
        // coalesced request (a lane per read, a granule per step, fetched every sector four times: 24.6 GB for a ladder that
        // takes 6 GB of reads, and the launch ran at HBM speed) -- classifies its granule in the 2-bit geometry of the count
        // kernels (vk_lane.h), keeps the bytes from the read's first (known) up to the first newline at or behind it (found
        // across the quad with one ballot), takes its K - 1 bases of context from the lane before it (the quad's last lane
        // of the sector before for lane 0), takes out the windows that would span a multiple of 500 bases of the read,
        // and counts its 16 positions' windows with the hand-written block of the k <= 7 kernels (windows_lds1) or, for
        // k = 8, 9, one by one into the subsample's row.
        // (positions as 32-bit offsets: an indexed sample is shorter than 4 GiB less a margin -- vkimg.hip count_walk)
        const uint32_t len32 = static_cast<uint32_t>(len), len16 = (len32 + 15u) & ~15u;
        const uint32_t g = static_cast<uint32_t>(lane) & 3u;
        auto walk = [&](uint32_t nb) __attribute__((always_inline)) {
            for (uint32_t sb = 0; sb < nb; sb += 16u) {
                const uint32_t r = sb + (static_cast<uint32_t>(lane) >> 2);
                bool active = r < nb;
                const uint32_t p = active ? q[r] + 1u : 0u;   // first byte of the sequence line
                if (p >= len32) active = false;
                uint32_t at = (p & ~63u) + 16u * g;          // this lane's granule of the read's first sector
                // read position of the granule's position 0, modulo 500 (negative before the read's first byte)
                uint32_t r500 = at >= p ? at - p : vkl::kBreakLength - (p - at);
                uint32_t carry_c = 0u, carry_bad = 0x55555555u;
                // positions before the read's first byte: in its first sector only (the reads of a batch start together)
                const int d0 = static_cast<int>(p - at);
                uint32_t below_start = d0 <= 0 ? 0u : (d0 >= 16 ? 0xFFFFFFFFu : ((1u << (2u * static_cast<uint32_t>(d0))) - 1u));
                while (__any(active)) {
                    uint4 v = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
                    if (active && at < len16) v = *reinterpret_cast<const uint4*>(sbase + at);
                    uint32_t C, IV, NL;
                    if (__any(((v.x | v.y | v.z | v.w) & 0x80808080u) != 0u)) vkl::classify_granule_nl<false>(v.x, v.y, v.z, v.w, C, IV, NL);
                    else vkl::classify_granule_nl<true>(v.x, v.y, v.z, v.w, C, IV, NL);
                    // positions at and beyond the end of the text (the last sector of a sample only)
                    uint32_t below_end = 0xFFFFFFFFu;
                    if (__any(at + 16u > len32)) below_end = at + 16u <= len32 ? 0xFFFFFFFFu : (at >= len32 ? 0u : ((1u << (2u * (len32 - at))) - 1u));
                    const uint32_t stop = (NL | ~below_end) & ~below_start & 0x55555555u;   // line ends at or behind the first byte
                    // has the line ended in a granule of this sector before mine?
                    const unsigned long long sm = __ballot(active && stop != 0u);
                    const uint32_t quad = static_cast<uint32_t>(sm >> (static_cast<uint32_t>(lane) & ~3u)) & 15u;
                    const bool over = (quad & ((1u << g) - 1u)) != 0u;
                    const uint32_t low = (stop - 1u) & ~stop;                  // everything below my first line end (all ones without one)
                    const uint32_t SEQ = (active && !over) ? (low & ~below_start) : 0u;
                    below_start = 0u;
                    taken_sites += vkl::popc(SEQ & 0x55555555u);
                    const uint32_t bad = (IV | ~SEQ) & 0x55555555u;
                    const uint32_t c_before = dpp_or_zero<0x111, 0xF>(C), bad_before = dpp_or_zero<0x111, 0xF>(bad);   // row_shr:1: the lane before
                    const uint32_t prev_c = g == 0u ? carry_c : c_before, prev_bad = g == 0u ? carry_bad : bad_before;
                    carry_c = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(C), 0xFF, 0xF, 0xF, true));      // quad_perm [3,3,3,3]
                    carry_bad = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(bad), 0xFF, 0xF, 0xF, true));
                    uint32_t ok = vkl::ok_mask1<K>(prev_bad, bad);
                    // breaklength: positions t of this granule with (r500 + t) % 500 == 0 start a new piece of the read: the
                    // windows ending at t .. t + K - 2 span the break
                    if (__any(r500 + 16u + static_cast<uint32_t>(K) > vkl::kBreakLength || r500 < static_cast<uint32_t>(K))) {
                        const uint32_t t0 = (vkl::kBreakLength - r500) % vkl::kBreakLength;    // first break at or behind position 0
                        uint32_t kill = 0u;
#pragma unroll
                        for (int rep = 0; rep < 2; ++rep) {   // the break at t0, and the one 500 before it, whose windows may still end in here
                            const int t = rep == 0 ? static_cast<int>(t0) : static_cast<int>(t0) - static_cast<int>(vkl::kBreakLength);
                            const int lo = t < 0 ? 0 : t, hi = t + K - 2;    // windows ending at lo .. hi
                            if (hi >= 0 && lo < 16) {
                                const uint32_t hi1 = static_cast<uint32_t>(hi < 15 ? hi : 15) + 1u;
                                const uint32_t mhi = hi1 >= 16u ? 0xFFFFFFFFu : ((1u << (2u * hi1)) - 1u);
                                const uint32_t mlo = (1u << (2u * static_cast<uint32_t>(lo))) - 1u;
                                kill |= mhi & ~mlo;
                            }
                        }
                        ok &= ~kill;   // (the read's first base is a break too, but nothing counts before it anyway)
                    }
                    if constexpr (LDSH) {
                        uint32_t pa;
                        unsigned long long pm;
                        windows_lds1<K>(prev_c, C, ok, hist_base, pa, pm);
                    } else {
                        vkl::windows1<K>(prev_c, C, ok, [&](uint32_t field4) { atomicAdd(&out[pair_reverse(field4 >> 2, K)], 1u); });
                    }
                    r500 = r500 + 64u >= vkl::kBreakLength ? r500 + 64u - vkl::kBreakLength : r500 + 64u;
                    at += 64u;
                    if (quad != 0u || (at & ~63u) >= len32) active = false;   // the line ended in this sector, or the text did
                    // The lanes meet HERE, not at the loop's head.  For k = 8, 9 the body ends in windows1's per-lane
                    // `if (ok bit 15) atomicAdd`, and what follows it is plain arithmetic: hipcc 7.2 threaded the two ways
                    // out of that `if` into two back edges, made an inner and an outer loop of them, and the lanes without
                    // a sixteenth window ran on through the ballot and the DPP moves at the head while the others waited at
                    // the outer latch -- context read from lanes outside EXEC, line ends voted by a part of the quad
                    // (tools/walker_fence_probe.py).  A convergent operation can be neither duplicated into
                    // the two paths nor made to depend on the `if`, so the body keeps one latch behind the join.
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
        for (uint32_t i = 0; i < n; i += 64u) {
            const bool have = i + static_cast<uint32_t>(lane) < n;
            const uint32_t a = have ? seg[i + lane] : 0u;
            const bool take = have && vkl::sample_take(seed, static_cast<uint64_t>(a), thr);
            const unsigned long long tm = __ballot(take);
            if (take) q[npend + static_cast<uint32_t>(__builtin_popcountll(tm & ((1ull << lane) - 1ull)))] = a;
            npend += static_cast<uint32_t>(__builtin_popcountll(tm));
            wave_lds_fence();
            if (npend >= 64u) {
                walk(64u);
                wave_lds_fence();
                const uint32_t rest = npend - 64u;   // < 64: move to the front
                const uint32_t keep = static_cast<uint32_t>(lane) < rest ? q[64 + lane] : 0u;
                wave_lds_fence();
                if (static_cast<uint32_t>(lane) < rest) q[lane] = keep;
                npend = rest;
                wave_lds_fence();
            }
        }
        if (npend != 0u) walk(npend);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
        if (wp.sites) {
            const uint32_t tot = lane_bcast(wave_inclusive_sum(taken_sites), 63);
            if (lane == 0 && tot != 0u) atomicAdd(&wp.sites[2ull * pair + 1], static_cast<unsigned long long>(tot));
        }
    }
    if (wp.sites && tid == 0 && part == 0) wp.sites[2ull * pair] = ip.sites[smp];
    if (LDSH) {
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        for (uint32_t code = tid; code < NCODE; code += kCountThreads) {
            const uint32_t v = hist[pair_reverse(code, K)];
            if (atomic_flush) {
                if (v) atomicAdd(&out[code], v);
            } else {
                out[code] = v;
            }
        }
    }
}

}  // namespace

#endif  // VK_LADDER_H
