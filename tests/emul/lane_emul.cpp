// lane_emul.cpp -- CPU emulation of vk_count_kernel's wavefront algorithm (test only).
//
// Re-uses the product's per-lane SWAR routines (varkoder_amd/csrc/vk_lane.h, host
// build) and re-states the surrounding wave logic of csrc/vkimg.hip sequentially:
// byte-range split over parts x 16 waves, local line-phase recovery, pieces that start
// one block before the range, newline prefix over the 64 lanes, halo carried from lane
// to lane.  Lets the CPU test-suite check the algorithm against the oracle without a GPU.
#include <cstdint>
#include <cstring>
#include <vector>

#include "vk_lane.h"

namespace {

constexpr int kWaves = 16;
constexpr int kPiece = 4096;

uint32_t pair_reverse(uint32_t c, int k) {
    uint32_t r = 0;
    for (int i = 0; i < k; ++i) { r = (r << 2) | (c & 3u); c >>= 2; }
    return r;
}

// csrc/vkimg.hip sync_rule() / sync_phase(), scalar
uint32_t sync_rule(const uint8_t* s, uint64_t at, uint64_t len) {
    uint64_t nl[6];
    uint32_t n = 0;
    for (uint64_t p = at; p < len && n < 6; ++p)
        if (s[p] == '\n') nl[n++] = p;
    for (uint32_t i = 0; i + 2 < n && i < 4; ++i) {
        uint64_t li = nl[i] + 1, lj = nl[i + 2] + 1;
        if (lj < len && s[li] == '@' && s[lj] == '+') return (3u - i) & 3u;
    }
    return 4u;
}

uint32_t sync_phase(const uint8_t* s, uint64_t w0, uint64_t len) {
    uint32_t ph = sync_rule(s, w0, len);
    if (ph < 4u) return ph;
    const uint64_t back = w0 > 65536 ? w0 - 65536 : 0;
    uint32_t cnt = 0;
    if (back != 0) {
        ph = sync_rule(s, back, len);
        if (ph < 4u) {
            for (uint64_t p = back; p < w0; ++p) cnt += s[p] == '\n';
            return (ph + cnt) & 3u;
        }
    }
    for (uint64_t p = 0; p < w0; ++p) cnt += s[p] == '\n';
    return cnt & 3u;
}

// wave_stream's SUB branch needs the newline before the pre-block when a range is entered inside a
// sequence line (last_newline_before in vkimg.hip)
static uint64_t last_newline_before(const uint8_t* s, uint64_t end) {
    while (end > 0)
        if (s[--end] == '\n') return end;
    return ~0ull;
}

struct Sub {
    bool on;
    uint64_t seed, threshold;
    uint64_t sites[2];
};

template <int K>
int count_impl(const uint8_t* s, uint64_t len, uint32_t parts, uint32_t* hist, uint32_t* status, Sub* sub = nullptr) {
    const uint32_t ncode = 1u << (2 * K);
    std::vector<uint32_t> raw(ncode, 0u);
    const uint64_t nblk = (len + 63) >> 6;
    const uint64_t bwg = (nblk + parts - 1) / parts;
    const uint64_t bw = (bwg + kWaves - 1) / kWaves;
    uint32_t st = 0, prev_end = 0;
    if (len && s[0] != '@') st |= 1u;
    {
        uint32_t seen = 0;
        const uint64_t lim = len < 65536 ? len : 65536;
        for (uint64_t p = 0; p < lim; ++p)
            if (s[p] == '\n' && ++seen == 2) {
                if (p + 1 < len && s[p + 1] != '+') st |= 1u;
                break;
            }
    }
    for (uint32_t part = 0; part < parts; ++part)
        for (int wave = 0; wave < kWaves; ++wave) {
            uint64_t blk0 = (uint64_t)part * bwg + (uint64_t)wave * bw;
            uint64_t blk1 = (uint64_t)part * bwg + std::min<uint64_t>((uint64_t)(wave + 1) * bw, bwg);
            if (blk1 > nblk) blk1 = nblk;
            if (blk0 >= blk1) continue;
            const uint64_t w0 = blk0 << 6, w1 = std::min<uint64_t>(blk1 << 6, len);
            const uint32_t ph0 = w0 ? sync_phase(s, w0, len) : 0u;
            if (ph0 != prev_end) st |= 2u;
            const long long o0 = (long long)w0 - 64;
            const uint64_t npieces = (w1 - w0 + 64 + kPiece - 1) / kPiece;
            uint32_t carry_c = 0, carry_bad = 0x55555555u, pph = 0, sub_carry = 0;
            long long read_start_rel = 0;  // where the sequence line running into the piece starts, relative to the piece
            for (uint64_t it = 0; it < npieces; ++it) {
                uint8_t piece[kPiece];
                for (int i = 0; i < kPiece; ++i) {
                    long long off = o0 + (long long)it * kPiece + i;
                    piece[i] = (off >= 0 && (uint64_t)off < w1) ? s[off] : 0;
                }
                vkl::LaneBits lb[64];
                uint32_t c[64], total = 0;
                bool non_ascii = false;  // the kernel's __any(has_non_ascii)
                for (int lane = 0; lane < 64; ++lane) {
                    uint32_t d[16];
                    memcpy(d, piece + 64 * lane, 64);
                    non_ascii |= vkl::has_non_ascii(d);
                }
                for (int lane = 0; lane < 64; ++lane) {
                    uint32_t d[16];
                    memcpy(d, piece + 64 * lane, 64);
                    c[lane] = non_ascii ? vkl::classify<false>(d, lb[lane]) : vkl::classify<true>(d, lb[lane]);
                    total += c[lane];
                }
                if (it == 0) pph = ph0 - c[0];
                bool any_gt3 = false, any_eq4 = false;  // the kernel's wave-uniform tiers
                for (int lane = 0; lane < 64; ++lane) { any_gt3 |= c[lane] > 4; any_eq4 |= c[lane] > 3; }
                uint32_t excl = 0;
                if (sub && sub->on && it == 0 && w0 != 0 && (pph & 3u) == 1u) {
                    const uint64_t a = last_newline_before(s, (uint64_t)o0);
                    sub_carry = (a != ~0ull && vkl::sample_take(sub->seed, a, sub->threshold)) ? 1u : 0u;
                    read_start_rel = a != ~0ull ? (long long)(a + 1) - o0 : 0;
                }
                long long rstart = read_start_rel;  // lanes run in order: the kernel's fetch from the nearest anchor lane
                for (int lane = 0; lane < 64; ++lane) {
                    const uint32_t lph = (pph + excl) & 3u;
                    excl += c[lane];
                    uint32_t s_raw = 0;
                    vkl::Mask128 seq = any_gt3 ? vkl::seq_mask_general(lb[lane].NL, lph)
                                       : any_eq4 ? vkl::seq_mask_fast4(lb[lane].NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw)
                                       : (sub && sub->on) ? vkl::seq_mask_fast(lb[lane].NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw)
                                                          : vkl::seq_mask_count(lb[lane].NL, lph);
                    uint32_t bad[4], ok[4];
                    vkl::bad_mask(lb[lane], seq, bad);
                    const uint32_t badh = carry_bad, ch = carry_c;  // lane-1's (or last piece's lane 63)
                    carry_bad = bad[3];
                    carry_c = lb[lane].C[3];
                    vkl::ok_mask<K>(badh, bad, ok);
                    if (it == 0 && lane == 0) ok[0] = ok[1] = ok[2] = ok[3] = 0;
                    if (sub && sub->on) {
                        // lanes run in order here, so the max-scan of the kernel is a running value
                        const uint64_t base = (uint64_t)(o0 + (long long)it * kPiece) + 64ull * lane;
                        uint32_t first[4], inc[4], anchors, take, seq_at;
                        if (any_gt3 || any_eq4) {
                            uint32_t la;
                            anchors = vkl::sample_strings_general(lb[lane].NL, lph, base, sub->seed, sub->threshold,
                                                                  first, inc, take, la);
                            seq_at = la + 1u;
                        } else {
                            anchors = (lph != 1u && s_raw <= 64u) ? 1u : 0u;
                            take = (anchors && vkl::sample_take(sub->seed, base + s_raw - 1u, sub->threshold)) ? 1u : 0u;
                            for (int g = 0; g < 4; ++g) { first[g] = anchors ? 0u : ~0u; inc[g] = take ? ~0u : 0u; }
                            seq_at = s_raw;
                        }
                        {   // breaklength: no window over a multiple of 500 bases of the read the block begins in
                            const uint32_t rel0 = (uint32_t)(64ll * lane - rstart);
                            uint32_t q1, lo2, hi2;
                            vkl::break_stretches<K>(rel0 % vkl::kBreakLength, q1, lo2, hi2);
                            const vkl::Mask128 m1 = vkl::ones_below(q1), m2a = vkl::ones_not_below(lo2), m2b = vkl::ones_below(hi2);
                            for (int g = 0; g < 4; ++g) ok[g] &= ~(((m2a.w[g] & m2b.w[g]) | m1.w[g]) & first[g]);
                            if (anchors) rstart = 64ll * lane + seq_at;
                        }
                        const uint32_t inh = 0u - sub_carry;
                        if (anchors) sub_carry = take;
                        const bool mine = !(it == 0 && lane == 0) && base < w1;
                        for (int g = 0; g < 4; ++g) {
                            const uint32_t takem = (first[g] & inh) | inc[g];
                            ok[g] &= takem;
                            if (mine) {
                                const uint32_t sites = seq.w[g] & ~lb[lane].NL[g] & 0x55555555u;
                                sub->sites[0] += vkl::popc(sites);
                                sub->sites[1] += vkl::popc(sites & takem);
                            }
                        }
                    }
                    vkl::windows<K>(ch, lb[lane].C, ok, [&](uint32_t a4) { raw[a4 >> 2]++; }, [] {});
                }
                pph += total;
                read_start_rel = rstart - kPiece;
            }
            prev_end = pph & 3u;
        }
    if (len) {
        uint32_t want = s[len - 1] == '\n' ? 0u : 3u;
        if (prev_end != want) st |= 2u;
    }
    for (uint32_t i = 0; i < ncode; ++i) hist[pair_reverse(i, K)] += raw[i];
    *status = st;
    return 0;
}

// csrc/vk_count.h vk_count_dense_kernel, sequentially: line pass per piece, granules with sequence bytes
// handed through the 64-granule exchange buffer in file order (start tag on bit 7 of the first byte),
// rounds of 64, the pending granules counted before a piece takes the general path, context from the
// previous granule of the stream.  `stats` (optional): [0] pieces on the fast path, [1] pieces in all,
// [2] rounds, [3] granules counted in rounds, [4] lanes set aside.
template <int K>
int count_dense_impl(const uint8_t* s, uint64_t len, uint32_t parts, uint32_t* hist, uint32_t* status, uint64_t* stats) {
    const uint32_t ncode = 1u << (2 * K);
    std::vector<uint32_t> raw(ncode, 0u);
    const uint64_t nblk = (len + 63) >> 6;
    const uint64_t bwg = (nblk + parts - 1) / parts;
    const uint64_t bw = (bwg + kWaves - 1) / kWaves;
    uint32_t st = 0, prev_end = 0;
    if (len && s[0] != '@') st |= 1u;
    {
        uint32_t seen = 0;
        const uint64_t lim = len < 65536 ? len : 65536;
        for (uint64_t p = 0; p < lim; ++p)
            if (s[p] == '\n' && ++seen == 2) {
                if (p + 1 < len && s[p + 1] != '+') st |= 1u;
                break;
            }
    }
    struct Granule { uint32_t a[4]; };
    for (uint32_t part = 0; part < parts; ++part)
        for (int wave = 0; wave < kWaves; ++wave) {
            uint64_t blk0 = (uint64_t)part * bwg + (uint64_t)wave * bw;
            uint64_t blk1 = (uint64_t)part * bwg + std::min<uint64_t>((uint64_t)(wave + 1) * bw, bwg);
            if (blk1 > nblk) blk1 = nblk;
            if (blk0 >= blk1) continue;
            const uint64_t w0 = blk0 << 6, w1 = std::min<uint64_t>(blk1 << 6, len);
            const uint32_t ph0 = w0 ? sync_phase(s, w0, len) : 0u;
            if (ph0 != prev_end) st |= 2u;
            const bool has_pre = w0 != 0;
            const uint64_t o0 = has_pre ? w0 - 64 : 0;
            const uint64_t span = w1 - o0;
            const uint32_t npieces = (uint32_t)((span + kPiece - 1) / kPiece);
            uint32_t ctx_c = 0, ctx_bad = 0x55555555u, pph = 0, npend = 0;
            std::vector<uint32_t> alist;   // lanes set aside
            bool aside63 = false;
            Granule xb[64];
            auto round = [&](uint32_t n) {  // xb[0 .. n) are real, lanes beyond idle along on newlines
                uint32_t C[64], bad[64];
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    Granule q = {{0x0A0A0A0Au, 0x0A0A0A0Au | vkl::kGranuleEnd, 0x0A0A0A0Au, 0x0A0A0A0Au}};
                    if (lane < n) q = xb[lane];
                    uint32_t SEQ;
                    vkl::classify_granule_note(q.a[0], q.a[1], q.a[2], q.a[3], C[lane], bad[lane], SEQ);
                    if (stats) stats[5] += vkl::popc(SEQ & 0x55555555u);   // sequence bytes the notes name
                }
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t badh = lane ? bad[lane - 1] : ctx_bad, ch = lane ? C[lane - 1] : ctx_c;
                    const uint32_t ok = vkl::ok_mask1<K>(badh, bad[lane]);
                    vkl::windows1<K>(ch, C[lane], ok, [&](uint32_t a4) { raw[a4 >> 2]++; });
                }
                ctx_bad = bad[n - 1];
                ctx_c = C[n - 1];
                if (stats) { stats[2]++; stats[3] += n; }
            };
            for (uint32_t it = 0; it < npieces; ++it) {
                uint8_t piece[kPiece];
                for (int i = 0; i < kPiece; ++i) {
                    const uint64_t rel = (uint64_t)it * kPiece + i;
                    piece[i] = rel < span ? s[o0 + rel] : 0;  // the registers: the last piece is clipped at w1
                }
                if (stats) stats[1]++;
                bool fast = it != 0 && it + 1 != npieces;
                uint32_t sp[64], ep[64], total = 0;
                if (fast) {
                    for (int lane = 0; lane < 64 && fast; ++lane) {
                        uint32_t d[16];
                        memcpy(d, piece + 64 * lane, 64);
                        if (vkl::ascii_or(d)) fast = false;
                    }
                }
                bool plainl[64];
                if (fast) {
                    uint32_t excl = 0, na = 0, lphs[64];
                    for (int lane = 0; lane < 64; ++lane) {
                        uint32_t d[16], mlo, mhi;
                        memcpy(d, piece + 64 * lane, 64);
                        vkl::newline_mask64(d, mlo, mhi);
                        const uint32_t c = vkl::popc(mlo) + vkl::popc(mhi);
                        lphs[lane] = (pph + excl) & 3u;
                        excl += c;
                        uint32_t s_raw;
                        plainl[lane] = vkl::seq_span_note(mlo, mhi, c, (1u - lphs[lane]) & 3u, sp[lane], ep[lane], s_raw);
                        na += plainl[lane] ? 0u : 1u;
                    }
                    total = excl;
                    if (na > 6) fast = false;    // kSetAside
                    else {
                        // lanes set aside: a separator granule in the stream, an entry in the wave's list
                        for (int lane = 0; lane < 64; ++lane) {
                            if (plainl[lane]) continue;
                            const bool before = lane == 0 ? aside63 : !plainl[lane - 1];
                            alist.push_back(((it * 64u + (uint32_t)lane) << 3) | (before ? 4u : 0u) | lphs[lane]);
                            sp[lane] = 0;
                            ep[lane] = 0;
                            if (stats) stats[4]++;
                        }
                        aside63 = !plainl[63];
                    }
                }
                if (!fast) aside63 = false;
                if (fast) {
                    if (stats) stats[0]++;
                    // the stream of this piece's granules, behind the pending ones; 64 at a time
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const uint32_t sl = sp[lane], el = ep[lane];
                        const uint32_t gs = vkl::span_first(sl), n = vkl::span_count(sl, el);
                        for (uint32_t g = gs; g < gs + n; ++g) {
                            Granule q;
                            memcpy(q.a, piece + 64 * lane + 16 * g, 16);
                            // the notes of the edge granules (vk_count.h: set on the copy in the exchange buffer)
                            if (g == gs && vkl::span_starts_inside(sl)) q.a[0] |= vkl::note_spread(sl & 15u);
                            if (el < 64u && g == vkl::span_last(el)) {
                                q.a[0] |= vkl::note_spread(el & 15u);
                                q.a[1] |= vkl::kGranuleEnd;
                            }
                            xb[npend++] = q;
                            if (npend == 64) {
                                round(64);
                                npend = 0;
                            }
                        }
                    }
                    pph += total;
                    continue;
                }
                // general path
                if (npend != 0) {
                    round(npend);
                    npend = 0;
                }
                vkl::LaneBits lb[64];
                uint32_t c[64];
                bool non_ascii = false;
                for (int lane = 0; lane < 64; ++lane) {
                    uint32_t d[16];
                    memcpy(d, piece + 64 * lane, 64);
                    non_ascii |= vkl::has_non_ascii(d);
                }
                total = 0;
                for (int lane = 0; lane < 64; ++lane) {
                    uint32_t d[16];
                    memcpy(d, piece + 64 * lane, 64);
                    c[lane] = non_ascii ? vkl::classify<false>(d, lb[lane]) : vkl::classify<true>(d, lb[lane]);
                    total += c[lane];
                }
                if (it == 0) pph = has_pre ? ph0 - c[0] : 0u;
                bool any_gt3 = false, any_eq4 = false;
                for (int lane = 0; lane < 64; ++lane) { any_gt3 |= c[lane] > 4; any_eq4 |= c[lane] > 3; }
                uint32_t excl = 0;
                for (int lane = 0; lane < 64; ++lane) {
                    const uint32_t lph = (pph + excl) & 3u;
                    excl += c[lane];
                    uint32_t s_raw = 0;
                    vkl::Mask128 seq = any_gt3 ? vkl::seq_mask_general(lb[lane].NL, lph)
                                       : any_eq4 ? vkl::seq_mask_fast4(lb[lane].NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw)
                                                 : vkl::seq_mask_count(lb[lane].NL, lph);
                    uint32_t bad[4], ok[4];
                    vkl::bad_mask(lb[lane], seq, bad);
                    const uint32_t badh = ctx_bad, ch = ctx_c;
                    ctx_bad = bad[3];
                    ctx_c = lb[lane].C[3];
                    vkl::ok_mask<K>(badh, bad, ok);
                    if (it == 0 && lane == 0 && has_pre) ok[0] = ok[1] = ok[2] = ok[3] = 0;
                    vkl::windows<K>(ch, lb[lane].C, ok, [&](uint32_t a4) { raw[a4 >> 2]++; }, [] {});
                }
                pph += total;
            }
            if (npend != 0) return 2;  // cannot happen: the last piece of a range takes the general path
            // the lanes set aside, counted exactly (vk_count.h set_aside_count): three blocks per entry through the
            // front end of the classic kernel; every window that touches the lane
            for (uint32_t w : alist) {
                const uint64_t blk = w >> 3;
                uint32_t cprev = 0, carry_bad = 0x55555555u, carry_c = 0;
                for (uint32_t j = 0; j < 3; ++j) {
                    uint8_t bytes[64];
                    for (int i = 0; i < 64; ++i) {
                        const uint64_t rel = (blk - 1 + j) * 64 + i;
                        bytes[i] = rel < span ? s[o0 + rel] : 0;
                    }
                    uint32_t d[16];
                    memcpy(d, bytes, 64);
                    vkl::LaneBits lb;
                    const uint32_t c = vkl::classify<false>(d, lb);
                    const uint32_t lph = ((w & 3u) + (j == 0 ? 0u - c : (j == 2 ? cprev : 0u))) & 3u;
                    cprev = c;
                    const vkl::Mask128 seq = vkl::seq_mask_general(lb.NL, lph);
                    uint32_t bad[4], ok[4];
                    vkl::bad_mask(lb, seq, bad);
                    vkl::ok_mask<K>(carry_bad, bad, ok);
                    const uint32_t ch = carry_c;
                    carry_bad = bad[3];
                    carry_c = lb.C[3];
                    const uint32_t kBack = (1u << (2 * (K - 1))) - 1u;
                    if (j == 0) ok[0] = ok[1] = ok[2] = ok[3] = 0;
                    if (j == 1 && (w & 4u)) ok[0] &= ~kBack;
                    if (j == 2) { ok[0] &= kBack; ok[1] = ok[2] = ok[3] = 0; }
                    vkl::windows<K>(ch, lb.C, ok, [&](uint32_t a4) { raw[a4 >> 2]++; }, [] {});
                }
            }
            prev_end = pph & 3u;
        }
    if (len) {
        uint32_t want = s[len - 1] == '\n' ? 0u : 3u;
        if (prev_end != want) st |= 2u;
    }
    for (uint32_t i = 0; i < ncode; ++i) hist[pair_reverse(i, K)] += raw[i];
    *status = st;
    return 0;
}

}  // namespace

extern "C" int emul_count_dense(const uint8_t* s, uint64_t len, int k, uint32_t parts, uint32_t* hist, uint32_t* status,
                                uint64_t* stats) {
    switch (k) {
        case 5: return count_dense_impl<5>(s, len, parts, hist, status, stats);
        case 6: return count_dense_impl<6>(s, len, parts, hist, status, stats);
        case 7: return count_dense_impl<7>(s, len, parts, hist, status, stats);
        default: return 1;
    }
}

extern "C" int emul_count(const uint8_t* s, uint64_t len, int k, uint32_t parts, uint32_t* hist, uint32_t* status) {
    switch (k) {
        case 5: return count_impl<5>(s, len, parts, hist, status);
        case 6: return count_impl<6>(s, len, parts, hist, status);
        case 7: return count_impl<7>(s, len, parts, hist, status);
        case 8: return count_impl<8>(s, len, parts, hist, status);
        case 9: return count_impl<9>(s, len, parts, hist, status);
        default: return 1;
    }
}

extern "C" int emul_count_sampled(const uint8_t* s, uint64_t len, int k, uint32_t parts, uint64_t seed,
                                  uint64_t threshold, uint32_t* hist, uint32_t* status, uint64_t* sites) {
    Sub sub = {true, seed, threshold, {0, 0}};
    int rc;
    switch (k) {
        case 5: rc = count_impl<5>(s, len, parts, hist, status, &sub); break;
        case 6: rc = count_impl<6>(s, len, parts, hist, status, &sub); break;
        case 7: rc = count_impl<7>(s, len, parts, hist, status, &sub); break;
        case 8: rc = count_impl<8>(s, len, parts, hist, status, &sub); break;
        case 9: rc = count_impl<9>(s, len, parts, hist, status, &sub); break;
        default: return 1;
    }
    sites[0] = sub.sites[0];
    sites[1] = sub.sites[1];
    return rc;
}
