// vk_count.h -- K1: FASTQ text -> forward-strand k-mer histograms (count, spill and check kernels)
// Part of the one translation unit vkimg.hip (device code for gfx950; see the notes there).
#ifndef VK_COUNT_H
#define VK_COUNT_H

#include <hip/hip_runtime.h>

#include <cstdint>

#include "vk_lane.h"

namespace {

constexpr int kWaves = 16;             // wavefronts per count workgroup
constexpr int kCountThreads = kWaves * 64;
constexpr int kPiece = 4096;           // bytes per wave iteration (64 lanes x 64 B)
constexpr uint32_t kMaxBins = 16384;   // u32 LDS histogram bins per workgroup (64 KiB)
#ifndef VK_K1_OCC
#define VK_K1_OCC 8                     // waves per SIMD the LDS-histogram kernel is compiled for (two workgroups per CU)
#endif
#ifndef VK_K1_PF
#define VK_K1_PF 2                      // when the next piece is loaded: 0 at its own start, 1 after classify, 2 before the windows (fewest registers held longest)
#endif

// ---------------------------------------------------------------- helpers ----

__device__ __forceinline__ uint32_t nl_flags(uint32_t w) {
    // bit 7 of every byte that equals '\n' (exact, no borrow artefacts)
    uint32_t z = w ^ 0x0A0A0A0Au;
    uint32_t t = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;  // bit7 set iff byte != 0
    return ~t & 0x80808080u;
}

__device__ __forceinline__ uint32_t nl_count16(uint4 v) {
    return __popc(nl_flags(v.x)) + __popc(nl_flags(v.y)) + __popc(nl_flags(v.z)) + __popc(nl_flags(v.w));
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront execute in order; this only stops the
    // compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// --- DPP cross-lane moves (gfx9: row_shr, row_bcast15/31, wave_shr) ---------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
    // lanes whose source is out of range or whose row is masked receive 0
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, false));
}

// inclusive prefix sum over the 64 lanes: 4 row_shr steps inside rows of 16, then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 -- six v_add_u32 with DPP operands
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += dpp_or_zero<0x111, 0xF>(x);
    x += dpp_or_zero<0x112, 0xF>(x);
    x += dpp_or_zero<0x114, 0xF>(x);
    x += dpp_or_zero<0x118, 0xF>(x);
    x += dpp_or_zero<0x142, 0xA>(x);
    x += dpp_or_zero<0x143, 0xC>(x);
    return x;
}

// value of lane-1 (lane 0 receives `first`): wave_shr:1
__device__ __forceinline__ uint32_t wave_prev_lane(uint32_t x, uint32_t first) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(first), static_cast<int>(x), 0x138, 0xF, 0xF, false));
}

__device__ __forceinline__ uint32_t lane_bcast(uint32_t x, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), lane));
}

__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }

// 16 bytes at sample offset `off`, zero for every byte at or beyond `lim`.
// Requires the buffer to be readable up to the 16-byte rounded end (ABI contract).
__device__ __forceinline__ uint4 load_granule(const uint8_t* sbase, uint64_t off, uint64_t lim) {
    if (off + 16 <= lim) return *reinterpret_cast<const uint4*>(sbase + off);
    if (off >= lim) return zero4();
    uint4 v = *reinterpret_cast<const uint4*>(sbase + off);
    uint32_t keep = static_cast<uint32_t>(lim - off);  // 1..15 valid bytes
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int kb = static_cast<int>(keep) - 4 * d;
        uint32_t m = kb >= 4 ? 0xFFFFFFFFu : (kb <= 0 ? 0u : ((1u << (8 * kb)) - 1u));
        w[d] &= m;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Line phase (0 header, 1 sequence, 2 plus, 3 quality) of the line that contains
// byte w0 > 0 of a 4-line FASTQ, recovered from the bytes at and after w0:
// among four consecutive line starts exactly one is a header, and a line start l_i
// is a header iff byte[l_i] == '@' and byte[l_{i+2}] == '+' (a quality line may
// start with '@', but then l_{i+2} is a sequence line, which never starts with '+').
// Falls back to counting the newlines of [0, w0) when fewer than six newlines
// follow w0.  Wave-uniform; `slot` is this wave's private LDS scratch.
// The '@' / '+' rule at byte position `at`: phase of the line holding `at`, or 4 if fewer than
// the needed newlines follow.
__device__ uint32_t sync_rule(const uint8_t* sbase, uint64_t at, uint64_t len, uint64_t* slot, int lane) {
    uint32_t n = 0;
    uint64_t pos = at;
    while (n < 6 && pos < len) {
        uint8_t b = (pos + lane < len) ? sbase[pos + lane] : 0;
        unsigned long long m = __ballot(b == '\n');
        while (m && n < 6) {
            int j = __builtin_ctzll(m);
            slot[n] = pos + j;
            ++n;
            m &= m - 1;
        }
        pos += 64;
    }
    wave_lds_fence();
    uint32_t ph = 4u;
    for (uint32_t i = 0; i + 2 < n && i < 4; ++i) {
        uint64_t li = slot[i] + 1, lj = slot[i + 2] + 1;
        if (lj < len && sbase[li] == '@' && sbase[lj] == '+') {
            ph = (3u - i) & 3u;  // the line holding `at` is line -1: phase (-1 - i) mod 4
            break;
        }
    }
    wave_lds_fence();
    return ph;
}

// newlines in [from, to), both multiples of 16, counted by the whole wave
__device__ uint32_t count_newlines(const uint8_t* sbase, uint64_t from, uint64_t to, int lane) {
    uint32_t cnt = 0;
    for (uint64_t off = from + static_cast<uint64_t>(lane) * 16; off + 16 <= to; off += 1024)
        cnt += nl_count16(*reinterpret_cast<const uint4*>(sbase + off));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    return cnt;
}

__device__ uint32_t sync_phase(const uint8_t* sbase, uint64_t w0, uint64_t len, uint64_t* slot, int lane) {
    uint32_t ph = sync_rule(sbase, w0, len, slot, lane);
    if (ph < 4u) return ph;
    // Too few lines after w0 (a range at the very end of a sample): apply the rule 64 KiB earlier and
    // count the newlines in between; only a sample with lines longer than that falls through to
    // counting every newline before w0.
    const uint64_t back = w0 > 65536 ? w0 - 65536 : 0;
    if (back != 0) {
        ph = sync_rule(sbase, back, len, slot, lane);
        if (ph < 4u) return (ph + count_newlines(sbase, back, w0, lane)) & 3u;
    }
    return count_newlines(sbase, 0, w0, lane) & 3u;
}

// --------------------------------------------------------------- K1 count ----

// reverse the order of the K two-bit groups of a code (no complement)
__device__ __forceinline__ uint32_t pair_reverse(uint32_t c, int k) {
    // all 32 bits reversed (one v_bfrev_b32), then the two bits of every group swapped back
    const uint32_t x = __brev(c);
    return (((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1)) >> (32 - 2 * k);
}

// 16 bytes at signed sample offset `off`: zero before the sample and at or beyond `lim`.
__device__ __forceinline__ uint4 load_granule_s(const uint8_t* sbase, long long off, uint64_t lim) {
    if (off < 0) return zero4();
    return load_granule(sbase, static_cast<uint64_t>(off), lim);
}

// Byte range [w0, w1) of workgroup `part` of `parts`, wave `wave` of kWaves (64-byte blocks).
struct WaveRange {
    uint64_t w0, w1;
    bool empty;
};

__device__ __forceinline__ WaveRange wave_range(uint64_t len, uint32_t parts, uint32_t part, int wave) {
    const uint64_t nblk = (len + 63) >> 6;
    const uint64_t bwg = (nblk + parts - 1) / parts;
    const uint64_t bw = (bwg + kWaves - 1) / kWaves;
    uint64_t blk0 = static_cast<uint64_t>(part) * bwg + static_cast<uint64_t>(wave) * bw;
    uint64_t blk1 = static_cast<uint64_t>(part) * bwg + umin64(static_cast<uint64_t>(wave + 1) * bw, bwg);
    if (blk1 > nblk) blk1 = nblk;
    WaveRange r;
    r.empty = blk0 >= blk1;
    r.w0 = blk0 << 6;
    r.w1 = r.empty ? r.w0 : umin64(blk1 << 6, len);
    return r;
}

// One wavefront streams the FASTQ bytes [w0, w1) of a sample and hands every 64-byte block's code
// string and countable-window mask to windows(ch, C[4], ok[4]) (raw fields: first base least
// significant, vk_lane.h).  `scratch` is a few wave-private LDS words for the range-start sync,
// below/above the shared mask tables.  Returns the line phase at w0 and at w1.
// Window loop of the LDS-histogram kernels (K <= 7): same arithmetic as vkl::windows<K>, with the
// predicated histogram update written out.  hipcc lowers `if (carry) atomicAdd(...)` to
//   v_add_co -> s_and_saveexec -> s_cbranch_execz -> address VALU -> ds_add -> s_or exec
// per position: one long VALU -> SALU -> branch -> VALU -> LDS dependency chain (measured ~76
// cycles per position, 59 % of the kernel).  Here eight positions form one block:
//   8 x v_add_co_u32 w, s[pair_j], w, w      carry-outs (= lane predicates) parked in SGPR pairs
//   8 x { s_mov_b64 exec, s[pair_j] ; ds_add_u32 addr_j, one }
//   s_mov_b64 exec, -1
// so the VALU -> SALU hand-over is paid once per block and the addresses are ordinary VALU work
// the scheduler hoists.  Requires EXEC = all ones on entry (wave_stream runs with the whole wave
// active) and leaves it so.  lds_base = byte offset of the histogram in LDS.
// probe_addr / probe_mask: the histogram address and the lane predicate of ONE position of the piece (the
// last one the loop handles), for the caller's low-complexity test.
template <int K>
__device__ __forceinline__ void windows_lds(uint32_t ch, const uint32_t C[4], const uint32_t ok[4],
                                            uint32_t lds_base, uint32_t& probe_addr, unsigned long long& probe_mask) {
    static_assert(2 * K + 2 <= 16, "paired extraction needs the field << 2 to fit 16 bits");
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    const uint32_t one = 1u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // even bits: OK of positions 0..15 of this dword; odd bits: the same rotated by 8
        // positions -> shifting out the top bit twice yields positions (i - 8, i), i = 15..8
        uint32_t w = vkl::lshl_or(vkl::alignbit(ok[g], ok[g], 16u), 1u, ok[g]);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t a[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 15 - 4 * half - j;
                const int plo = 16 * g + i - 8;
                const int o = 30 + 2 * (plo - K + 1);
                const int word = o >> 5, sh = o & 31;
                uint32_t x;
                if (sh == 0) x = v[word];
                else if (word == 4) x = v[4] >> sh;
                else x = vkl::alignbit(v[word + 1], v[word], static_cast<uint32_t>(sh));
                a[2 * j] = (x & kMask4) + lds_base;              // position i - 8 (first carry)
                a[2 * j + 1] = ((x >> 16) & kMask4) + lds_base;  // position i
            }
            unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
            const unsigned long long exec_in = __builtin_amdgcn_read_exec();  // restored behind the block (the callers run with every lane active)
            asm volatile(
                "v_add_co_u32_e64 %0, %1, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %2, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %3, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %4, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %5, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %6, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %7, %0, %0\n\t"
                "v_add_co_u32_e64 %0, %8, %0, %0\n\t"
                "s_mov_b64 exec, %1\n\tds_add_u32 %9, %17\n\t"
                "s_mov_b64 exec, %2\n\tds_add_u32 %10, %17\n\t"
                "s_mov_b64 exec, %3\n\tds_add_u32 %11, %17\n\t"
                "s_mov_b64 exec, %4\n\tds_add_u32 %12, %17\n\t"
                "s_mov_b64 exec, %5\n\tds_add_u32 %13, %17\n\t"
                "s_mov_b64 exec, %6\n\tds_add_u32 %14, %17\n\t"
                "s_mov_b64 exec, %7\n\tds_add_u32 %15, %17\n\t"
                "s_mov_b64 exec, %8\n\tds_add_u32 %16, %17\n\t"
                "s_mov_b64 exec, %18"
                : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(one), "s"(exec_in)
                : "memory");
            if (g == 3 && half == 1) {
                probe_addr = a[7];
                probe_mask = m7;
            }
        }
    }
}

// Low-complexity input (poly-A tails, short tandem repeats): when the lanes of a position hold one or a few
// distinct k-mers, their ds_add serialise on a few LDS counters (64 lanes on one: ~15x the cost of a scattered
// add).  The piece loop therefore probes one position per piece -- do at least 7 of 8 counting lanes agree? --
// and, every sixteenth piece, whether the counting lanes fall into at most eight groups (a tandem repeat of
// period <= 8 seen at different phases); while the data looks like that, pieces are counted by
// windows_lds_hot: whole homopolymer pieces with one add per lane, lane groups (16 positions) that repeat with
// a short period with one add per residue class, everything else as usual.  Exact either way; ordinary reads
// never enter it.
__device__ __forceinline__ bool probe_is_hot(uint32_t addr, unsigned long long mask) {
    if (__builtin_popcountll(mask) < 8) return false;
    const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(addr), __builtin_ctzll(mask)));
    const unsigned long long eq = __ballot(addr == first) & mask;
    // an eighth of the counting lanes (and at least four) on ONE counter: never among unrelated k-mers, always in
    // a wave that holds repeats beside ordinary reads (7 of 8 lanes, the first rule, missed those mixtures)
    return __builtin_popcountll(eq) >= 4 && __builtin_popcountll(eq) * 8 >= __builtin_popcountll(mask);
}

__device__ __forceinline__ bool probe_few_groups(uint32_t addr, unsigned long long mask) {
    if (__builtin_popcountll(mask) < 24) return false;
    unsigned long long rem = mask;
    for (int n = 0; n < 8 && rem; ++n) {
        const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(addr), __builtin_ctzll(rem)));
        rem &= ~__ballot(addr == first);
    }
    return rem == 0ull;
}

// The two probes together; `tick` counts the calls of one wave.
__device__ __forceinline__ bool probe_low_complexity(uint32_t addr, unsigned long long mask_in, uint32_t& tick) {
    // (the mask is a ballot, one value for the wave, but general_piece has it behind a branch on a by-value argument and
    // hipcc keeps it in vector registers there: the probes' loop over groups then came out as a loop the lanes leave
    // one by one, with its v_readlane inside; a scalar again at no cost where it already was one)
    const unsigned long long mask = static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(mask_in)))) |
        static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(mask_in >> 32)))) << 32;
    if (probe_is_hot(addr, mask)) return true;
    return ((++tick) & 15u) == 0u && probe_few_groups(addr, mask);
}

// Is, in EVERY lane of the wave, every base under a counting window one and the same base?  Then every
// window of a lane is the same homopolymer k-mer: n = the lane's window count, base = its base (0..3).
template <int K>
__device__ __forceinline__ bool piece_is_homopolymer(uint32_t ch, const uint32_t C[4], const uint32_t ok[4], uint32_t& n,
                                                     uint32_t& base) {
    // bases covered by a counting window: OK smeared towards lower positions over K - 1 positions
    uint32_t w[5] = {0u, ok[0], ok[1], ok[2], ok[3]};
    int cover = 1;
#pragma unroll
    for (int step = 0; step < 4; ++step) {
        if (cover < K) {
            const int shp = cover < K - cover ? cover : K - cover;
            const uint32_t sh = 2u * static_cast<uint32_t>(shp);
            for (int i = 0; i < 4; ++i) w[i] |= vkl::alignbit(w[i + 1], w[i], sh);
            w[4] |= w[4] >> sh;
            cover += shp;
        }
    }
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    uint32_t one0 = 0, zero0 = 0, one1 = 0, zero1 = 0;  // does a covered base have bit 0 (bit 1) set / clear?
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        one0 |= v[i] & w[i];
        zero0 |= ~v[i] & w[i];
        one1 |= (v[i] >> 1) & w[i];
        zero1 |= ~(v[i] >> 1) & w[i];
    }
    const bool homo = !((one0 != 0u && zero0 != 0u) || (one1 != 0u && zero1 != 0u));
    n = vkl::popc(ok[0]) + vkl::popc(ok[1]) + vkl::popc(ok[2]) + vkl::popc(ok[3]);
    base = (one0 != 0u ? 1u : 0u) | (one1 != 0u ? 2u : 0u);
    return __all(homo);
}

// Tandem repeats.  The period P (1 .. min(8, K) bases) is read off one lane whose first 32 positions are all
// countable; a lane group g (16 positions, every window countable) repeats with it when the bases its windows cover
// -- the last K - 1 of the word before and its own sixteen -- equal themselves P positions on.  Then the window
// ending at position 16g + j (j < P) stands for every position of the group that is congruent to j, and the window
// ending at 16g (K >= P bases of a P-periodic string) names the repeat unit and its phase: the lanes of the wave that
// hold the same one there hold the same windows at every j.  They are found ONCE per group (up to eight classes;
// what is left stands alone), and one lane per class calls add(field, n) for all of them, n = the positions the
// window stands for x the lanes of the class -- 64 adds on one counter would be executed one after the other.
// Such groups leave ok[]; what remains (read ends, other reads) is for the caller to count as usual.
// Returns (wave-uniform) how many groups of the wave went this way.
template <int K, typename Add>
__device__ __forceinline__ uint32_t count_repeats(const uint32_t (&v)[5], uint32_t (&ok)[4], int lane, Add add) {
    constexpr uint32_t kMaxP = K < 8 ? K : 8;
    uint32_t P = 0, handled = 0;
    {
        // (up to four lanes are asked: in a wave that holds repeats beside ordinary reads the first one may be ordinary)
        unsigned long long cand = __ballot(ok[0] == 0x55555555u && ok[1] == 0x55555555u);
        for (int tries = 0; tries < 4 && cand != 0ull && P == 0u; ++tries) {
            const int src = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            const unsigned long long x = (static_cast<unsigned long long>(lane_bcast(v[2], src)) << 32) | lane_bcast(v[1], src);
#pragma unroll
            for (uint32_t p = kMaxP; p >= 1; --p)
                if ((((x >> (2u * p)) ^ x) << (2u * p)) == 0ull) P = p;  // bits [0, 64 - 2p) of the difference
        }
    }
    if (P == 0u) return 0u;
    const uint32_t sh2 = 2u * P;
    const uint32_t himask = 0xFFFFFFFFu >> sh2;
    constexpr uint32_t kField = (1u << (2 * K)) - 1u;
    // positions of a group (16) congruent to j modulo P: q + (j < r), q = 16 / P, r = 16 % P -- from a table: a division
    // by the run-time P, one per window, was 140 of the hot path's vector instructions
    const uint32_t q16 = static_cast<uint32_t>((0x1084321510ull >> (5u * (P - 1u))) & 31ull);  // 16, 8, 5, 4, 3, 2, 2, 2
    const uint32_t r16 = 16u - q16 * P;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t lo = v[g], hi = v[g + 1];
        const uint32_t dlo = (vkl::alignbit(hi, lo, sh2) ^ lo) & (0xFFFFFFFFu << (32 - 2 * (K - 1)));
        const uint32_t dhi = ((hi >> sh2) ^ hi) & himask;
        const bool rep = ok[g] == 0x55555555u && (dlo | dhi) == 0u;
        const unsigned long long repmask = __ballot(rep);
        if (repmask == 0ull) continue;
        handled += static_cast<uint32_t>(__builtin_popcountll(repmask));
        // classes of lanes that hold the same unit at the same phase: leaders (one lane each) and sizes
        const uint32_t key = vkl::alignbit(hi, lo, static_cast<uint32_t>(32 - 2 * (K - 1))) & kField;  // the window ending at 16g
        uint32_t cnt = 1u;                        // lanes beyond the eighth class stand alone
        unsigned long long leaders = 0ull, left = repmask;
        for (uint32_t t = 0; t < 8 && left != 0ull; ++t) {
            const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), __builtin_ctzll(left)));
            const unsigned long long eq = __ballot(key == first) & left;   // (every lane with this key is still in `left`)
            if (key == first) cnt = static_cast<uint32_t>(__builtin_popcountll(eq));
            leaders |= eq & (~eq + 1ull);
            left &= ~eq;
        }
        leaders |= left;
        if ((leaders >> lane) & 1ull) {
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                if (j < P) {  // (P is wave-uniform)
                    const int sh = 32 + 2 * (static_cast<int>(j) - K + 1);   // bit offset of the window in [lo | hi]
                    const uint32_t f = (sh < 32 ? vkl::alignbit(hi, lo, static_cast<uint32_t>(sh)) : (hi >> (sh - 32))) & kField;
                    add(f, __umul24(q16 + (j < r16 ? 1u : 0u), cnt));          // positions of the group congruent to j x lanes
                }
            }
        }
        if (rep) ok[g] = 0u;
    }
    return handled;
}

// did at least a third of the wave's groups that had windows go the short way?  (handled: count_repeats' result)
__device__ __forceinline__ bool repeats_dominate(uint32_t handled, const uint32_t ok_in[4]) {
    if (handled == 0u) return false;
    const uint32_t had = static_cast<uint32_t>(__builtin_popcountll(__ballot(ok_in[0] != 0u)) + __builtin_popcountll(__ballot(ok_in[1] != 0u)) +
                                               __builtin_popcountll(__ballot(ok_in[2] != 0u)) + __builtin_popcountll(__ballot(ok_in[3] != 0u)));
    return handled * 3u >= had;
}

// Returns what the piece was: 0 not low-complexity after all, 1 homopolymer, 2 tandem repeats (the caller stays in
// this mode while it is not 0).  try_homopolymer: false skips the homopolymer test (the caller passes it while the
// pieces are tandem repeats, with a look every eighth piece: poly-A is then still exact, as period 1).
template <int K>
__device__ __forceinline__ void windows_lds1(uint32_t ch, uint32_t C, uint32_t ok, uint32_t lds_base, uint32_t& probe_addr,
                                             unsigned long long& probe_mask);   // (defined with the dense kernel below)

// xb: 64 free uint4 slots of this wave in LDS (the dense kernel's exchange buffer, empty while a piece takes the general
// path), or null.  With it, the lane groups the shortcuts leave over -- the ends of the repeats' reads: one group in ten
// -- are packed into ONE round of window blocks; without it they take the eight blocks of a whole piece, whose few live
// lanes all hit the repeat's handful of counters (3.3 of 11.9 ms on (ACGT)n, profiles/ab/r05_low_complexity.txt).
template <int K>
__device__ uint32_t windows_lds_hot(uint32_t ch, const uint32_t C[4], const uint32_t ok_in[4], uint32_t* hist, uint32_t lds_base,
                                    int lane, bool try_homopolymer, uint32_t& probe_addr, unsigned long long& probe_mask,
                                    uint4* xb = nullptr) {
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    // the probe for the next piece: the window that ends at position 40 of every lane, however it gets counted
    {
        constexpr int Q = 40, o = 30 + 2 * (Q - K + 1), word = o >> 5, sh = o & 31;
        const uint32_t x = sh == 0 ? v[word] : vkl::alignbit(v[word + 1 < 5 ? word + 1 : 4], v[word], static_cast<uint32_t>(sh));
        probe_addr = x & kMask4;
        probe_mask = __ballot(((ok_in[Q >> 4] >> (2 * (Q & 15))) & 1u) != 0u);
    }
    // Homopolymer pieces (poly-A / poly-G tails, the common low-complexity case): every window of a lane is
    // the same k-mer and the lane adds its window count once -- one ds_add per lane and piece instead of 64.
    if (try_homopolymer) {
        uint32_t n, base;
        if (piece_is_homopolymer<K>(ch, C, ok_in, n, base)) {
            if (n != 0u) atomicAdd(&hist[base * (((1u << (2 * K)) - 1u) / 3u)], n);
            return 1u;
        }
    }
    uint32_t ok[4] = {ok_in[0], ok_in[1], ok_in[2], ok_in[3]};
    const uint32_t handled = count_repeats<K>(v, ok, lane, [&](uint32_t f, uint32_t n) {
        __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(lds_base + 4u * f)),
                               n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    });
#ifndef VK_DIAG_HOT_NO_LEFTOVER   // timing only: what the window blocks of the groups the shortcuts left over cost
    if (__any((ok[0] | ok[1] | ok[2] | ok[3]) != 0u)) {
        uint32_t pa;
        unsigned long long pm;
        bool packed = false;
        if (xb != nullptr) {
            const uint32_t n = (ok[0] != 0u) + (ok[1] != 0u) + (ok[2] != 0u) + (ok[3] != 0u);
            const uint32_t incl = wave_inclusive_sum(n);
            const uint32_t tot = lane_bcast(incl, 63);
            if (tot <= 64u) {
                uint32_t at = incl - n;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (ok[g] != 0u) xb[at++] = make_uint4(v[g], v[g + 1], ok[g], 0u);
                wave_lds_fence();
                uint4 e = make_uint4(0u, 0u, 0u, 0u);
                if (static_cast<uint32_t>(lane) < tot) e = xb[lane];
                windows_lds1<K>(e.x, e.y, e.z, lds_base, pa, pm);
                wave_lds_fence();
                packed = true;
            }
        }
        if (!packed) windows_lds<K>(ch, C, ok, lds_base, pa, pm);
    }
#endif
    return repeats_dominate(handled, ok_in) ? 2u : 0u;
}

// ---- read subsampling (vk_count_sampled_device; vk_lane.h: sample_hash) ------------------------
struct SubParams {
    const uint64_t* seeds;        // [nsamples]
    const uint64_t* thresholds;   // [nsamples], in [0, 2^32]
    unsigned long long* sites;    // [nsamples][2]: bytes of sequence lines, of which in taken reads; may be null
};

struct SubWave {                  // per-wave state of a subsampling launch
    uint64_t seed, threshold;
    uint32_t sites, sites_taken;  // per-lane partial sums
};

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t x) {
    x = max(x, dpp_or_zero<0x111, 0xF>(x));
    x = max(x, dpp_or_zero<0x112, 0xF>(x));
    x = max(x, dpp_or_zero<0x114, 0xF>(x));
    x = max(x, dpp_or_zero<0x118, 0xF>(x));
    x = max(x, dpp_or_zero<0x142, 0xA>(x));
    x = max(x, dpp_or_zero<0x143, 0xC>(x));
    return x;
}

// Offset of the last newline before sample offset `end` (end > 0), or ~0 when there is none.
// Wave-uniform; walks back 1 KiB at a time (one step for ordinary read lengths).
__device__ uint64_t last_newline_before(const uint8_t* sbase, uint64_t end, int lane) {
    while (end > 0) {
        const long long off = static_cast<long long>(end) - 1024 + 16ll * lane;
        const uint4 v = load_granule_s(sbase, off, end);
        const uint32_t f[4] = {nl_flags(v.x), nl_flags(v.y), nl_flags(v.z), nl_flags(v.w)};
        const bool any = off >= 0 && (f[0] | f[1] | f[2] | f[3]) != 0u;
        const unsigned long long b = __ballot(any);
        if (b) {
            const int hl = 63 - __clzll(b);
            uint32_t pos = 0;  // byte of the last newline inside the granule
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (f[d]) pos = 4u * d + ((31u - __clz(f[d])) >> 3);
            const uint32_t p = lane_bcast(pos, hl);
            return end - 1024 + 16ull * hl + p;
        }
        end = end > 1024 ? end - 1024 : 0;
    }
    return ~0ull;
}

#ifdef VK_STAMPS
// Diagnostic build only (tools/stamps.sh): per-segment cycle sums of the piece loop, written to a
// debug buffer that nothing else reads.  Never quote this build's run time.
__device__ unsigned long long g_vk_stamps[8];
#define VK_STAMP(t)                                                         \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#else
#define VK_STAMP(t) \
    do {            \
    } while (0)
#endif

// A value every lane of the wavefront holds alike, moved to scalar registers (the compiler cannot
// prove uniformity of anything derived from threadIdx or from a select).
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(x >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

template <int K, bool SUB, int PF, typename Windows, typename PieceStart>
__device__ __forceinline__ void wave_stream(const uint8_t* __restrict__ sbase_, uint64_t len_, uint64_t w0_,
                                            uint64_t w1_, uint64_t* scratch, const uint4* below, const uint4* above,
                                            int lane, Windows windows, PieceStart piece_start, uint32_t& ph_start,
                                            uint32_t& ph_end, SubWave& sw) {
    // the range is the same for the whole wave: keep it (and everything derived from it: piece
    // addresses, loop counter, edge tests) in scalar registers
    const uint8_t* sbase = reinterpret_cast<const uint8_t*>(uniform64(reinterpret_cast<uint64_t>(sbase_)));
    const uint64_t len = uniform64(len_), w0 = uniform64(w0_), w1 = uniform64(w1_);
    const uint32_t ph0 = (w0 != 0) ? sync_phase(sbase, w0, len, scratch, lane) : 0u;
    ph_start = ph0;

    // Pieces start one 64-byte block BEFORE the range: lane 0 of piece 0 (the "pre-block")
    // only supplies the k-1 bases of context and its windows are not counted.  From then on
    // lane 0 takes its context from lane 63 of the previous piece.  A range that starts at
    // offset 0 has no bytes before it and no pre-block.
    const bool has_pre = w0 != 0;
    const uint64_t o0 = has_pre ? w0 - 64 : 0;                                // sample offset of piece 0
    const uint64_t span = w1 - o0;                                            // bytes from there to w1
    const uint32_t npieces = static_cast<uint32_t>((span + kPiece - 1) / kPiece);
    const uint32_t tail_bytes = static_cast<uint32_t>(span % kPiece);         // bytes of a last, partial piece
    const uint32_t lane64 = static_cast<uint32_t>(lane) * 64u;
    // Every lane loads ITS 64 contiguous bytes (four 16-byte loads): one instruction touches 64
    // cache lines, but the four together use them completely and the lines stay in the CU's L1
    // between them -- measured HBM traffic equals the file size, and the former LDS transpose
    // (coalesced rows in, lane blocks out: 8 LDS instructions and 64 KiB of LDS) bought nothing.
    // The loads go through a buffer descriptor over [o0, w1 rounded up to 16): granules at or beyond
    // that end come back as zeros from the hardware's range check, so every piece -- the last, partial
    // one included -- takes the same four instructions (scalar piece offset + the lane's 32-bit
    // offset, no 64-bit address registers, no edge branch).  w1 is a multiple of 64 unless it is the
    // end of the sample, whose last granule the ABI makes readable; its bytes at and beyond w1 are
    // cleared below (it == npieces - 1).  (vkimg.hip keeps a wave's range below 4 GiB.)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(sbase + o0), 0, static_cast<int>((span + 15) & ~15ull), 0x00020000);
    uint4 r0, r1, r2, r3;
    auto load_piece = [&](uint32_t piece) {
        const uint32_t soff = piece * static_cast<uint32_t>(kPiece);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64, soff, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64 + 16u, soff, 0);
        const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64 + 32u, soff, 0);
        const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane64 + 48u, soff, 0);
        r0 = make_uint4(a.x, a.y, a.z, a.w);
        r1 = make_uint4(b.x, b.y, b.z, b.w);
        r2 = make_uint4(c.x, c.y, c.z, c.w);
        r3 = make_uint4(d.x, d.y, d.z, d.w);
    };
    // the one granule that can hold bytes at or beyond w1 (see above): keep its first n bytes
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
    if (PF != 0) load_piece(0);
    uint32_t carry_c = 0u, carry_bad = 0x55555555u;
    uint32_t pph = 0;  // line phase at the start of the current piece
    uint32_t sub_carry = 0u;  // SUB: is the read that runs into the current piece taken?
    int32_t read_start_rel = 0;  // SUB: where that read's sequence line starts, relative to the current piece (<= 0)
    auto tbl_below = [&](uint32_t q) {
        uint4 v = below[q];
        vkl::Mask128 m;
        m.w[0] = v.x; m.w[1] = v.y; m.w[2] = v.z; m.w[3] = v.w;
        return m;
    };
    auto tbl_above = [&](uint32_t q) {
        uint4 v = above[q];
        vkl::Mask128 m;
        m.w[0] = v.x; m.w[1] = v.y; m.w[2] = v.z; m.w[3] = v.w;
        return m;
    };
#ifdef VK_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, acc[4] = {0, 0, 0, 0};
#endif
#ifndef VK_DIAG_K1_NO_PACE   // (the dense kernel's two priority levels, see there: the last quarter of a range at the lower one.  k = 9 pass A:
    // 12.70 / 12.43 -> 12.45 / 12.42 ms on uniform reads, 13.50 -> 13.40 GC-skewed, 14.18 -> 13.97 fastp-shaped, alternating runs of one call)
    const uint32_t pace_at = npieces - (npieces >> 2);
    __builtin_amdgcn_s_setprio(1);
#endif
    for (uint32_t it = 0; it < npieces; ++it) {
#ifndef VK_DIAG_K1_NO_PACE
        if (it == pace_at) __builtin_amdgcn_s_setprio(0);
#endif
        VK_STAMP(t0);
        if (PF == 0) load_piece(it);
        if (it + 1 == npieces && (tail_bytes & 15u) != 0u) {  // wave-uniform, once per range at most
            uint32_t tb = tail_bytes;
            asm volatile("" : "+s"(tb));  // keeps the masks of this cold path from being hoisted into registers held across the loop
            const int n = static_cast<int>(tb) - static_cast<int>(lane64);
            clip_granule(r0, n);
            clip_granule(r1, n - 16);
            clip_granule(r2, n - 32);
            clip_granule(r3, n - 48);
        }
        const uint4 q0 = r0, q1 = r1, q2 = r2, q3 = r3;  // this lane's 64 bytes
        // the consumer's vector-memory work of the PREVIOUS piece goes out here, behind the wait for this
        // piece's bytes: it then has a whole piece of arithmetic to retire before the next such wait
        // (loads, stores and atomics share one in-order counter, vmcnt)
        piece_start(q0);
        const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        VK_STAMP(t1);
        vkl::LaneBits lb;
        const uint32_t c = __any(vkl::has_non_ascii(d)) ? vkl::classify<false>(d, lb) : vkl::classify<true>(d, lb);

        // prefetch the next piece under the rest of the work; the bytes of this one are consumed, so
        // the loads land in the same registers
        if (PF == 1 && it + 1 < npieces) load_piece(it + 1);
        VK_STAMP(t2);
        // newline prefix over the wave -> line phase at the start of each lane's block
        const uint32_t incl = wave_inclusive_sum(c);
        const uint32_t total = lane_bcast(incl, 63);
        if (it == 0) pph = has_pre ? ph0 - lane_bcast(c, 0) : 0u;  // the pre-block's newlines precede w0
        const uint32_t lph = (pph + incl - c) & 3u;

        vkl::Mask128 seq;
        const bool degenerate = __any(c > 4u);  // wave-uniform tiers: <= 3 newlines per block, 4, more
        uint32_t s_raw = 0;
        const bool four = !degenerate && __any(c > 3u);
        if (degenerate) seq = vkl::seq_mask_general(lb.NL, lph);
        else if (four) seq = vkl::seq_mask_fast4(lb.NL, lph, tbl_below, tbl_above, s_raw);
        else if constexpr (SUB) seq = vkl::seq_mask_fast(lb.NL, lph, tbl_below, tbl_above, s_raw);  // needs s_raw
        else seq = vkl::seq_mask_count(lb.NL, lph);  // no LDS lookups in the steady state
        uint32_t bad[4], ok[4];
        vkl::bad_mask(lb, seq, bad);

        const uint32_t badh = wave_prev_lane(bad[3], carry_bad);
        const uint32_t ch = wave_prev_lane(lb.C[3], carry_c);
        carry_bad = lane_bcast(bad[3], 63);
        carry_c = lane_bcast(lb.C[3], 63);

        vkl::ok_mask<K>(badh, bad, ok);
        if (it == 0 && lane == 0 && has_pre) { ok[0] = 0u; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }

        if constexpr (SUB) {
            // Which read does each position belong to, and is that read taken?  A block either has
            // an anchor (the newline that ends a header line) and decides for what follows it, or
            // inherits the decision of the nearest anchor before it: a max-scan over
            // (lane + 1) << 1 | take, seeded with the decision carried in from the previous piece.
            const uint64_t base = o0 + static_cast<uint64_t>(it) * kPiece + 64ull * lane;
            if (it == 0 && w0 != 0 && (pph & 3u) == 1u) {
                // the range is entered inside a sequence line whose header ended before the pre-block
                const uint64_t a = last_newline_before(sbase, o0, lane);
                sub_carry = (a != ~0ull && vkl::sample_take(sw.seed, a, sw.threshold)) ? 1u : 0u;
                // (a read more than 2 GiB long would wrap this: such a sequence line is not FASTQ anyone subsamples)
                read_start_rel = a != ~0ull ? static_cast<int32_t>(static_cast<int64_t>(a + 1) - static_cast<int64_t>(o0)) : 0;
            }
            uint32_t first[4], inc[4], anchors, take, seq_at;  // seq_at: block position where the last anchor's read starts
            if (degenerate || four) {  // several reads may meet in one block: position by position
                uint32_t la;
                anchors = vkl::sample_strings_general(lb.NL, lph, base, sw.seed, sw.threshold, first, inc, take, la);
                seq_at = la + 1u;
            } else {
                anchors = (lph != 1u && s_raw <= 64u) ? 1u : 0u;
                take = (anchors && vkl::sample_take(sw.seed, base + s_raw - 1u, sw.threshold)) ? 1u : 0u;
                const uint32_t f = anchors ? 0u : 0xFFFFFFFFu, n = take ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g) { first[g] = f; inc[g] = n; }
                seq_at = s_raw;  // (a block with an anchor on this tier begins outside a sequence line: nothing of the read before)
            }
            const uint32_t v = anchors ? (((static_cast<uint32_t>(lane) + 1u) << 1) | take) : 0u;
            const uint32_t scan = wave_inclusive_max(v);
            const uint32_t pscan = wave_prev_lane(scan, 0u);
            const uint32_t inherited = max(pscan, sub_carry) & 1u;
            sub_carry = max(lane_bcast(scan, 63), sub_carry) & 1u;
            // breaklength (vkl::kBreakLength): where, relative to this piece, does the sequence line start that the
            // block begins in?  Behind the nearest anchor of an earlier lane, else where the piece before left it.
            const int32_t my_start = static_cast<int32_t>(lane64 + seq_at);
            const uint32_t alane = (pscan >> 1) - 1u;
            const int32_t from_lane = __builtin_amdgcn_ds_bpermute(static_cast<int>(alane << 2), my_start);
            const int32_t rstart = pscan != 0u ? from_lane : read_start_rel;
            {
                const uint32_t last = lane_bcast(scan, 63);
                const int32_t at_last = __builtin_amdgcn_readlane(my_start, static_cast<int>(((last >> 1) - 1u) & 63u));
                read_start_rel = (last != 0u ? at_last : read_start_rel) - static_cast<int32_t>(kPiece);
            }
            {
                const uint32_t rel0 = static_cast<uint32_t>(static_cast<int32_t>(lane64) - rstart);  // read position of the block's first byte
                uint32_t q1, lo2, hi2;
                vkl::break_stretches<K>(rel0 % vkl::kBreakLength, q1, lo2, hi2);
                const vkl::Mask128 m1 = tbl_below(q1), m2a = tbl_above(lo2), m2b = tbl_below(hi2);
#pragma unroll
                for (int g = 0; g < 4; ++g) ok[g] &= ~(((m2a.w[g] & m2b.w[g]) | m1.w[g]) & first[g]);
            }
            const uint32_t inh = 0u - inherited;
            // the pre-block belongs to the previous range; bytes at or beyond w1 are zero fill
            const bool mine = !(it == 0 && lane == 0 && has_pre) && base < w1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t takem = (first[g] & inh) | inc[g];
                ok[g] &= takem;
                if (mine) {  // (the position-by-position mask marks a line's own newline too)
                    const uint32_t sites = seq.w[g] & ~lb.NL[g] & 0x55555555u;
                    sw.sites += __popc(sites);
                    sw.sites_taken += __popc(sites & takem);
                }
            }
        }
        VK_STAMP(t3);
        if (PF == 2 && it + 1 < npieces) load_piece(it + 1);
        windows(ch, lb.C, ok);  // the consumer's window stage (LDS histogram or bucket queues)
        pph += total;
#ifdef VK_STAMPS
        VK_STAMP(t4);
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3;
#endif
    }
#ifdef VK_STAMPS
    if (lane == 0 && blockIdx.x == 300) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_vk_stamps[i], acc[i]);
        atomicAdd(&g_vk_stamps[4], npieces);
    }
#endif
    ph_end = pph & 3u;
}

__device__ __forceinline__ void fill_mask_tables(uint4* below, uint4* above, int tid) {
    if (tid < 66) {
        vkl::Mask128 m = vkl::ones_below(static_cast<uint32_t>(tid));
        below[tid] = make_uint4(m.w[0], m.w[1], m.w[2], m.w[3]);
        above[tid] = make_uint4(~m.w[0], ~m.w[1], ~m.w[2], ~m.w[3]);
    }
}

// K <= 7: the whole 4^K u32 histogram lives in LDS.
// Per-wave sums of a subsampling launch -> sites[sample][2].
__device__ __forceinline__ void flush_sites(const SubParams& sp, uint32_t s, const SubWave& sw, int lane) {
    if (!sp.sites) return;
    const uint32_t a = lane_bcast(wave_inclusive_sum(sw.sites), 63);
    const uint32_t b = lane_bcast(wave_inclusive_sum(sw.sites_taken), 63);
    if (lane == 0) {
        atomicAdd(&sp.sites[2ull * s], static_cast<unsigned long long>(a));
        atomicAdd(&sp.sites[2ull * s + 1], static_cast<unsigned long long>(b));
    }
}

template <int K, bool SUB>
__global__ __launch_bounds__(kCountThreads, SUB ? 4 : VK_K1_OCC) void vk_count_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
    const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ wavephase, int atomic_flush, SubParams sp) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    static_assert(NCODE <= kMaxBins, "LDS histogram too large");

    // The LDS histogram is indexed by the RAW packed field (first base least
    // significant, see vk_lane.h); the flush un-reverses to the ABI's code order.
    __shared__ uint32_t hist[NCODE];
    __shared__ uint64_t scratch[kWaves][8];  // sync_phase: newline positions after a range start
    __shared__ uint4 below[66];  // below[q] = bits [0, 2q) of a 128-bit string
    __shared__ uint4 above[66];  // above[q] = ~below[q]

    const uint32_t unit = blockIdx.x;
    const uint32_t s = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the range and the piece loop live on the SALU

    for (uint32_t i = tid; i < NCODE; i += kCountThreads) hist[i] = 0u;
    fill_mask_tables(below, above, tid);
    __syncthreads();

    const uint8_t* sbase = fastq + offs[s];
    const uint64_t len = lens[s];
    const WaveRange wr = wave_range(len, parts, part, wave);
    uint32_t ph_start = 0, ph_end = 0;
    if (!wr.empty) {
        const uint32_t hist_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) uint32_t*)hist));
        uint32_t hot = 0, hot_tick = 0;  // the last piece looked low-complexity (1 homopolymer, 2 tandem): see windows_lds_hot
        auto win = [&](uint32_t ch, const uint32_t* C, const uint32_t* ok) __attribute__((always_inline)) {
            uint32_t pa;
            unsigned long long pm;
            uint32_t still = 0;
            if (hot == 0u) windows_lds<K>(ch, C, ok, hist_base, pa, pm);
            else still = windows_lds_hot<K>(ch, C, ok, hist, hist_base, lane, hot != 2u || ((++hot_tick) & 7u) == 0u, pa, pm);
            hot = still != 0u ? still : (probe_is_hot(pa, pm) ? 1u : 0u);
        };
        SubWave sw = {0, 0, 0, 0};
        if constexpr (SUB) {
            sw.seed = sp.seeds[s];
            sw.threshold = sp.thresholds[s];
        }
        wave_stream<K, SUB, SUB ? 1 : VK_K1_PF>(sbase, len, wr.w0, wr.w1, &scratch[wave][0], below, above, lane, win,
                                                [](const uint4&) {}, ph_start, ph_end, sw);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
        if constexpr (SUB) flush_sites(sp, s, sw, lane);
    }
    if (lane == 0) wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));

    __syncthreads();
    uint32_t* out = hist_out + static_cast<uint64_t>(s) * NCODE;
    // ABI order: first base most significant.  The loop runs over the OUTPUT codes (consecutive threads store
    // consecutive words) and takes each one's counter from its reversed place in LDS -- the other way round every
    // wavefront's store touched 64 lines.
    for (uint32_t code = tid; code < NCODE; code += kCountThreads) {
        const uint32_t v = hist[pair_reverse(code, K)];
        if (atomic_flush) {
            if (v) atomicAdd(&out[code], v);
        } else {
            out[code] = v;
        }
    }
}

// This lane's number, recomputed where it is needed (two v_mbcnt): the dense kernel's piece loop holds no
// per-lane value across iterations besides the prefetched bytes -- what lives across the call of the
// general path would otherwise sit in scratch and come back with a vmcnt(0) wait at every use.
__device__ __forceinline__ uint32_t lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// ---- the read index (vk_ladder.h; vk_count_dense_kernel<K, true> fills it on the way) ---------------------
struct IndexParams {
    uint32_t* anchors;        // anchor lists of every sample of the launch
    const uint64_t* base;     // [nsamples] first entry of the sample's region
    uint32_t* count;          // [nsamples * parts * kWaves] anchors in the wave's segment
    unsigned long long* sites;  // [nsamples] bytes of all sequence lines
    uint32_t* overflow;       // [nsamples] != 0: a segment was full (more than one read per 32 bytes): no index for this sample
};

// anchors a wavefront's segment holds: one per 32 bytes of its range (a record of 150-base reads is 320), and a few
__device__ __host__ __forceinline__ uint64_t index_segment_cap(uint64_t range_bytes) { return range_bytes / 32 + 8; }

// first entry of the segment of wave `wave` of workgroup `part`; bw = bytes per wave range (wave_range's, a multiple of 64)
__device__ __forceinline__ uint64_t index_segment(const IndexParams& ip, uint32_t smp, uint64_t w0, uint32_t part, int wave) {
    return ip.base[smp] + w0 / 32 + 8ull * (static_cast<uint64_t>(part) * kWaves + static_cast<uint32_t>(wave));
}

// Bit p of {hi, lo} set <=> byte p of the block is '\n' -- for any bytes (vkl::newline_mask64 wants ASCII).
__device__ __forceinline__ void newline_mask64_any(const uint32_t d[16], uint32_t& lo, uint32_t& hi) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = vkl::udot4(nl_flags(d[2 * k + 1]), 0x80402010u, vkl::udot4(nl_flags(d[2 * k]), 0x08040201u, 0u));
    lo = (v[0] >> 7) | (v[1] << 1) | (v[2] << 9) | (v[3] << 17);
    hi = (v[4] >> 7) | (v[5] << 1) | (v[6] << 9) | (v[7] << 17);
}

// The anchors among the newlines of one lane's 64-byte block -- bit p of {mhi, mlo}: byte p is a newline; lph: phase of
// the line the block begins in; blk: the block's sample offset -- stored behind the `nanch` the wavefront's segment
// holds (cap: its room; `full` is raised instead of writing beyond it), and the block's share of the site count: the
// positions of the sequence lines' ends minus the positions behind their anchors.  All lanes call it (`mine`: this
// lane's block takes part); a round per newline of the fullest block.
__device__ __forceinline__ void index_block(uint32_t mlo, uint32_t mhi, uint32_t lph, uint64_t blk, bool mine, uint32_t* seg, uint32_t cap,
                                            uint32_t& nanch, bool& full, long long& acc) {
    unsigned long long m = mine ? ((static_cast<unsigned long long>(mhi) << 32) | mlo) : 0ull;
    const uint32_t ln = lane_now();
    uint32_t t = 0;
    while (__any(m != 0ull)) {   // newline by newline: three rounds for ordinary reads
        const bool have = m != 0ull;
        const uint32_t pos = have ? static_cast<uint32_t>(__builtin_ctzll(m)) : 0u;
        m &= m - 1ull;
        const uint32_t ph = (lph + t) & 3u;   // the line this newline ends
        const bool anchor = have && ph == 0u;
        if (have && ph == 1u) acc += static_cast<long long>(blk + pos);
        if (anchor) acc -= static_cast<long long>(blk + pos) + 1ll;
        const unsigned long long am = __ballot(anchor);
        if (am != 0ull) {
            const uint32_t na = static_cast<uint32_t>(__builtin_popcountll(am));
            if (nanch + na > cap) full = true;
            else if (anchor) seg[nanch + static_cast<uint32_t>(__builtin_popcountll(am & ((1ull << ln) - 1ull)))] = static_cast<uint32_t>(blk + pos);
            if (!full) nanch += na;
        }
        ++t;
    }
}

// ---- K <= 7, sequence-only heavy stage ("dense" kernel) ---------------------------------------
// Same ranges, pieces, loads and histogram as vk_count_kernel; what changes is what a piece costs.
// The LINE pass (vkl::newline_mask64 / seq_span, ~130 vector instructions per piece, no transposes)
// finds each lane's stretch of sequence line; the 16-byte granules that hold sequence bytes (52 % of
// all granules for 150-base reads) are handed from the registers of the lanes that loaded them to the
// lanes of the HEAVY stage through a 1 KiB exchange buffer per wave in LDS: 64 granules = one round,
// one granule per lane, in file order.  Transposes, classification, window masks and histogram updates
// then touch sequence bytes only, and a window block has ~60 live lanes instead of ~29.  Granules left
// over (< 64) stay in the buffer for the next piece.  Context (the K - 1 bases before a granule) comes
// from the previous lane of the round -- the previous granule of the stream.  Where that is not the
// granule's neighbour in the file, its last position is never a sequence base (a granule goes to the
// heavy stage as soon as ONE of its positions has line phase 1, the newline that ends the line
// included), so no window can reach across.  What a granule's sequence bytes are needs one bit of
// company: "the line starts inside" (bit 7 of the first byte; the bytes are ASCII) -- otherwise they are
// the bytes before its first newline.
// Pieces the line pass cannot describe in this form (a lane with more than three newlines or with a
// sequence line that starts after a '+' or quality line in the same 64 bytes: reads under ~45 bases;
// bytes >= 0x80; low-complexity runs; the first and last piece of a range) take the general path of
// vk_count_kernel on the registers they already hold, after the pending granules have been counted.
// LDS: 64 KiB histogram + 16 x 1 KiB exchange = exactly half of a CU's 160 KiB, two workgroups per CU
// (tools/occupancy_census.hip: they do co-reside); the range-start scratch lives in the wave's buffer.

// Timing diagnostics (results are wrong with either): -DVK_DIAG_NO_DSADD drops the histogram atomics of the dense
// kernel's heavy stage, -DVK_DIAG_NO_HEAVY the whole heavy stage (line pass, hand-over and loads remain).
#ifdef VK_DIAG_NO_DSADD
#define VK_DSADD(a) "s_nop 0\n\t"
#else
#define VK_DSADD(a) "ds_add_u32 " a ", %17\n\t"
#endif
template <int K>
__device__ __forceinline__ void windows_lds1(uint32_t ch, uint32_t C, uint32_t ok, uint32_t lds_base,
                                             uint32_t& probe_addr, unsigned long long& probe_mask) {
    static_assert(2 * K + 2 <= 16, "paired extraction needs the field << 2 to fit 16 bits");
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    const uint32_t one = 1u;
    uint32_t w = vkl::lshl_or(vkl::alignbit(ok, ok, 16u), 1u, ok);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t a[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 15 - 4 * half - j;
            const int plo = i - 8;
            const int o = 30 + 2 * (plo - K + 1);
            const int word = o >> 5, sh = o & 31;
            uint32_t x;
            if (word == 1) x = sh == 0 ? C : (C >> sh);  // the last fields end exactly at bit 64
            else x = vkl::alignbit(C, ch, static_cast<uint32_t>(sh));
            a[2 * j] = (x & kMask4) + lds_base;
            a[2 * j + 1] = ((x >> 16) & kMask4) + lds_base;
        }
        unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
        const unsigned long long exec_in = __builtin_amdgcn_read_exec();
        asm volatile(
            "v_add_co_u32_e64 %0, %1, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %2, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %3, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %4, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %5, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %6, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %7, %0, %0\n\t"
            "v_add_co_u32_e64 %0, %8, %0, %0\n\t"
            "s_mov_b64 exec, %1\n\t" VK_DSADD("%9") 
            "s_mov_b64 exec, %2\n\t" VK_DSADD("%10") 
            "s_mov_b64 exec, %3\n\t" VK_DSADD("%11") 
            "s_mov_b64 exec, %4\n\t" VK_DSADD("%12") 
            "s_mov_b64 exec, %5\n\t" VK_DSADD("%13") 
            "s_mov_b64 exec, %6\n\t" VK_DSADD("%14") 
            "s_mov_b64 exec, %7\n\t" VK_DSADD("%15") 
            "s_mov_b64 exec, %8\n\t" VK_DSADD("%16") 
            "s_mov_b64 exec, %18"
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(one), "s"(exec_in)
            : "memory");
        if (half == 1) {
            probe_addr = a[7];
            probe_mask = m7;
        }
    }
}

// The general path of one piece for the dense kernel: exactly vk_count_kernel's piece (all 64 bytes of
// every lane classified, any number of newlines, bytes >= 0x80, the low-complexity window loop); the
// mask tables of the rare tiers are worked out on the spot (no LDS left for them).
struct GeneralState {
    uint32_t ctx_c, ctx_bad, pph, hot, tick;   // in / out (wave-uniform)
    uint32_t nanch, full, sites;               // INDEX: anchors in the wave's segment (in / out), segment full (out), this lane's sequence bytes (out)
};



// (bytes and state travel in registers: through a struct in memory every call cost a round trip to scratch)
// `packed`: bit 0 first piece of the range, bit 1 it starts with a pre-block, bits 2-3 the line phase at the range's start,
// bits 4.. the bytes of the piece that lie inside the range (the last piece's tail is zero fill, not text).
// INDEX: iseg / icap = the wave's segment of the anchor list and its room, iblk = sample offset of the piece's first byte.
// (Scalars only, thirty-two dwords of them: everything travels in registers.  A small struct passed by value went
// through the caller's stack frame, and hipcc 7.2 let its slot share bytes with a value spilled around the same call --
// the site counts of the read index came back with whatever the slot held.)
template <int K, bool INDEX>
__device__ __attribute__((noinline)) GeneralState general_piece(uint4 q0, uint4 q1, uint4 q2, uint4 q3, GeneralState st,
                                                                uint32_t packed, uint32_t* hist, uint32_t hist_base,
                                                                uint32_t* iseg, uint32_t icap, uint32_t iblk) {
    const uint32_t flags = packed & 3u, ph0 = (packed >> 2) & 3u, ivalid = packed >> 4;
    const int lane = static_cast<int>(lane_now());
    const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const bool first = (flags & 1u) != 0u, has_pre = (flags & 2u) != 0u;  // first piece of the range; it starts with a pre-block
    uint32_t pph = st.pph;
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
    uint32_t bad[4], ok[4];
    vkl::bad_mask(lb, seq, bad);
    const uint32_t badh = wave_prev_lane(bad[3], st.ctx_bad);
    const uint32_t ch = wave_prev_lane(lb.C[3], st.ctx_c);
    st.ctx_bad = lane_bcast(bad[3], 63);
    st.ctx_c = lane_bcast(lb.C[3], 63);
    vkl::ok_mask<K>(badh, bad, ok);
    if (first && lane == 0 && has_pre) { ok[0] = 0u; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }
    if constexpr (INDEX) {   // the piece's anchors and sequence bytes (the pre-block is the wave before's)
        const bool mine = !(first && lane == 0 && has_pre);
        uint32_t mlo, mhi;
        {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = vkl::udot4(nl_flags(d[2 * j + 1]), 0x80402010u, vkl::udot4(nl_flags(d[2 * j]), 0x08040201u, 0u));
            mlo = (v[0] >> 7) | (v[1] << 1) | (v[2] << 9) | (v[3] << 17);
            mhi = (v[4] >> 7) | (v[5] << 1) | (v[6] << 9) | (v[7] << 17);
        }
        bool full = false;
        long long unused = 0;
        uint32_t nanch = st.nanch;
        index_block(mlo, mhi, lph, static_cast<uint64_t>(iblk) + 64u * static_cast<uint32_t>(lane), mine, iseg, icap, nanch, full, unused);
        st.nanch = nanch;
        st.full = full ? 1u : 0u;
        uint32_t sites = 0;
        const int vb = static_cast<int>(ivalid) - 64 * lane;     // bytes of this lane's block inside the range
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int vg = vb - 16 * g;
            const uint32_t inside = vg >= 16 ? 0xFFFFFFFFu : (vg <= 0 ? 0u : ((1u << (2 * vg)) - 1u));
            sites += vkl::popc(seq.w[g] & ~lb.NL[g] & inside & 0x55555555u);
        }
        st.sites = mine ? sites : 0u;
    }
    uint32_t pa;
    unsigned long long pm;
    uint32_t still = 0;
    uint32_t tick = st.tick & 0x7FFFFFFFu;   // bit 31 of the caller's tick: the last low-complexity piece was tandem repeats
    if (st.hot == 0u) windows_lds<K>(ch, lb.C, ok, hist_base, pa, pm);
    else {
        ++tick;
        // (the wave's exchange buffer sits behind the histogram in the dense kernel's LDS block and is empty here: the pending
        // granules were counted before the call)
        uint4* const xb = reinterpret_cast<uint4*>(hist + (1u << (2 * K))) + (threadIdx.x >> 6) * 64u;
        still = windows_lds_hot<K>(ch, lb.C, ok, hist, hist_base, lane, (st.tick >> 31) == 0u || (tick & 7u) == 0u, pa, pm, xb);
    }
    st.hot = (still != 0u || probe_low_complexity(pa, pm, tick)) ? 1u : 0u;
    st.tick = (tick & 0x7FFFFFFFu) | (still == 2u ? 0x80000000u : 0u);
    st.pph = pph + total;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
    return st;
}

// ---- lanes set aside --------------------------------------------------------------------------------
// A lane whose 64 bytes the line pass cannot describe as one tagged stretch of granules (vkl::seq_span: four or
// more newlines, reads under ~20-45 bases) does not send the piece down the general path: the lane puts ONE granule
// of newlines into the stream in place of its own -- a separator: no window of the stream reaches into or across the
// lane -- and leaves a word in the wave's list (kSetAside per piece at most, else the piece does take the general
// path).  Behind the count kernel vk_aside_kernel counts the listed lanes exactly, a wavefront per list, 21 entries at
// a time: three lanes per entry load the 64 bytes before the lane, the lane's and the 64 behind it, and go through the
// front end of vk_count_kernel (every byte classified, any number of newlines); what is counted is every window that touches the
// lane: all that end in it -- not those that reach back into a lane before it that was set aside itself, which that
// lane's entry has counted -- and those that end in the first K - 1 positions behind it.
// Entry: (block offset from the range's first piece) << 3 | (the lane before was set aside) << 2 | line phase at the lane.
constexpr uint32_t kSetAside = 6;         // lanes of one piece that may be set aside
constexpr uint32_t kSetAsideBatch = 21;   // entries a wavefront counts at a time (three lanes each)

// vk_aside_kernel<K><<<count launch's grid, 1024>>>, behind the count kernel in the stream (its histogram rows are stored
// by then): a workgroup per workgroup of the count launch, a wavefront per list.  k <= 7: the windows found go into a
// histogram in LDS (raw-field order, the count kernels' window blocks) that is added to the sample's row at the end --
// with one global atomic per window the kernel took 3.6 ms beside a 77 ms count of fastp-shaped reads (780 M atomics);
// k = 8, 9 (the packed route): one atomic per window.  The deferred count as a function of the count kernel itself,
// called or inlined behind its piece loop, made the K = 5 build lose the counts of that loop (gfx950, ROCm 7.2; not
// understood: the same source was exact for K = 6, 7 and for K = 5 as soon as the function's adds were compiled out).
// INDEX: the listed lanes' anchors and sequence bytes are added to the read index (vk_count_dense_kernel<K, true>).
template <int K, bool INDEX>
__global__ __launch_bounds__(kCountThreads, 8) void vk_aside_kernel(const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
                                                                     const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
                                                                     uint32_t* __restrict__ hist_out, const uint32_t* __restrict__ aside,
                                                                     uint32_t aside_cap, const uint32_t* __restrict__ aside_n, IndexParams ip,
                                                                     uint32_t upb) {   // upb: workgroups of the count launch (of ONE sample) that one workgroup here serves
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr bool LDSH = NCODE <= kMaxBins;
    __shared__ uint32_t lhist[LDSH ? NCODE : 1];
    __shared__ uint32_t any_list;
    // Round 6: the count launch is many small workgroups per sample now (vkimg.hip, choose_parts); one workgroup HERE zeroes
    // and flushes a 64 KB histogram whatever it finds to count, so it serves `upb` of them (all of one sample) in turn.
    const uint32_t ablocks = (parts + upb - 1u) / upb;
    const uint32_t smp = blockIdx.x / ablocks;
    const uint32_t part0 = (blockIdx.x % ablocks) * upb, part1 = part0 + upb < parts ? part0 + upb : parts;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6)));
    if (tid == 0) any_list = 0u;
    __syncthreads();
    {
        uint32_t any = 0u;
        for (uint32_t part = part0; part < part1; ++part) any |= aside_n[(smp * parts + part) * kWaves + wave];
        if (any != 0u && (tid & 63u) == 0u) any_list = 1u;
    }
    __syncthreads();
    if (any_list == 0u) return;                 // (uniform for the workgroup: most workgroups of a launch on ordinary reads)
    if constexpr (LDSH) {
        for (uint32_t i = tid; i < NCODE; i += kCountThreads) lhist[i] = 0u;
        __syncthreads();
    }
    const uint32_t lhist_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)lhist));
    const uint8_t* sbase = fastq + offs[smp];
    uint32_t* hist = hist_out + static_cast<uint64_t>(smp) * NCODE;
    const uint32_t lane = tid & 63u;
    const uint32_t ent = lane / 3u, j = lane - 3u * ent;
    for (uint32_t part = part0; part < part1; ++part) {
    const uint32_t unit = smp * parts + part;
    const uint32_t gw = unit * kWaves + wave;   // wave of the count launch
    const uint32_t naside = aside_n[gw];
    const WaveRange wr = wave_range(lens[smp], parts, part, static_cast<int>(wave));
    const uint64_t o0 = wr.w0 != 0 ? wr.w0 - 64 : 0;
    const uint64_t span = wr.w1 - o0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(sbase + o0), 0, static_cast<int>((span + 15) & ~15ull), 0x00020000);
    const uint32_t* list = aside + static_cast<uint64_t>(gw) * aside_cap;
    uint32_t* iseg = nullptr;
    uint32_t icap = 0, nanch = 0, isites = 0;
    bool ifull = false;
    if constexpr (INDEX) {
        iseg = ip.anchors + index_segment(ip, smp, wr.w0, part, static_cast<int>(wave));
        icap = static_cast<uint32_t>(index_segment_cap(wr.w1 - wr.w0));
        nanch = ip.count[gw];
    }
    for (uint32_t at = 0; at < naside; at += kSetAsideBatch) {
        const uint32_t n = naside - at < kSetAsideBatch ? naside - at : kSetAsideBatch;
        const bool live = ent < n;
        const uint32_t w = live ? list[at + ent] : 0u;
        // (the lane's block is never the range's first: pieces 1 .. are the line pass's)
        const uint32_t off = live ? ((w >> 3) - 1u + j) * 64u : 0xFFFFFF00u;   // beyond the range: zeros
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 16, 0);
        const u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 32, 0);
        const u32x4 e4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 48, 0);
        const uint32_t d[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x, c4.y, c4.z, c4.w, e4.x, e4.y, e4.z, e4.w};
        vkl::LaneBits lb;
        const uint32_t c = vkl::classify<false>(d, lb);
        const uint32_t cprev = wave_prev_lane(c, 0u);
        const uint32_t lph = ((w & 3u) + (j == 0u ? 0u - c : (j == 2u ? cprev : 0u))) & 3u;
        vkl::Mask128 seq;
        uint32_t s_raw = 0;
        if (__any(c > 4u)) seq = vkl::seq_mask_general(lb.NL, lph);   // (position by position: slow, and rarely needed)
        else seq = vkl::seq_mask_fast4(lb.NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw);
        uint32_t bad[4], ok[4];
        vkl::bad_mask(lb, seq, bad);
        const uint32_t badh = wave_prev_lane(bad[3], 0x55555555u);
        const uint32_t ch = wave_prev_lane(lb.C[3], 0u);
        vkl::ok_mask<K>(badh, bad, ok);
        constexpr uint32_t kBack = (1u << (2 * (K - 1))) - 1u;   // OK bits of the windows that reach into the block before
        if (!live || j == 0u) { ok[0] = 0u; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }
        if (j == 1u && (w & 4u) != 0u) ok[0] &= ~kBack;
        if (j == 2u) { ok[0] &= kBack; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }
        if constexpr (LDSH) {
            uint32_t pa;
            unsigned long long pm;
            windows_lds<K>(ch, lb.C, ok, lhist_base, pa, pm);
        } else {
            vkl::windows<K>(ch, lb.C, ok, [&](uint32_t field4) { atomicAdd(&hist[pair_reverse(field4 >> 2, K)], 1u); }, [] {});
        }
        if constexpr (INDEX) {   // the lane's own block (j = 1): its anchors, its sequence bytes
            const bool mine = live && j == 1u;
            uint32_t mlo, mhi;
            {
                uint32_t v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = vkl::udot4(nl_flags(d[2 * jj + 1]), 0x80402010u, vkl::udot4(nl_flags(d[2 * jj]), 0x08040201u, 0u));
                mlo = (v[0] >> 7) | (v[1] << 1) | (v[2] << 9) | (v[3] << 17);
                mhi = (v[4] >> 7) | (v[5] << 1) | (v[6] << 9) | (v[7] << 17);
            }
            long long unused = 0;
            index_block(mlo, mhi, lph, o0 + static_cast<uint64_t>(off), mine, iseg, icap, nanch, ifull, unused);
            if (mine) {
#pragma unroll
                for (int g = 0; g < 4; ++g) isites += vkl::popc(seq.w[g] & ~lb.NL[g] & 0x55555555u);
            }
        }
    }
    if constexpr (INDEX) {
        const uint32_t tot = lane_bcast(wave_inclusive_sum(isites), 63);
        if (lane == 0u && naside != 0u) {
            ip.count[gw] = nanch;
            if (tot != 0u) atomicAdd(&ip.sites[smp], static_cast<unsigned long long>(tot));
            if (ifull) atomicOr(&ip.overflow[smp], 1u);
        }
    }
    }   // (the units of this workgroup)
    if constexpr (LDSH) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
        __syncthreads();
        for (uint32_t code = tid; code < NCODE; code += kCountThreads) {
            const uint32_t v = lhist[pair_reverse(code, K)];
            if (v) atomicAdd(&hist[code], v);
        }
    }
}

// INDEX: the launch also fills the read index of its samples (vk_ladder.h) -- every read's anchor, the bytes of all
// sequence lines -- so that a ladder of subsamples needs no pass of its own for it: an anchor falls out of the line pass
// (seq_span's s_raw), the sequence bytes out of the heavy stage's masks; the general path and vk_aside_kernel add theirs.
template <int K, bool INDEX>
__global__ __launch_bounds__(kCountThreads, VK_K1_OCC) void vk_count_dense_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
    const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ wavephase, int atomic_flush,
    uint32_t* __restrict__ aside, uint32_t aside_cap, uint32_t* __restrict__ aside_n, IndexParams ip) {   // aside[grid * kWaves][aside_cap]: the waves' lists of lanes set aside; aside_n[grid * kWaves]: their lengths
    constexpr uint32_t NCODE = 1u << (2 * K);
    static_assert(NCODE <= kMaxBins, "LDS histogram too large");
    // (one block, so that general_piece finds the exchange buffers behind the histogram it is handed)
    struct DenseLds {
        uint32_t hist[NCODE];      // raw-field order, as in vk_count_kernel
        uint4 xbuf[kWaves][64];    // per wave: 64 granules on their way to the heavy stage
    };
    __shared__ __attribute__((aligned(16))) DenseLds dlds;
    uint32_t (&hist)[NCODE] = dlds.hist;
    uint4 (&xbuf)[kWaves][64] = dlds.xbuf;

    const uint32_t unit = blockIdx.x;
    const uint32_t smp = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (uint32_t i = tid; i < NCODE; i += kCountThreads) hist[i] = 0u;
    __syncthreads();

    const WaveRange wr = wave_range(lens[smp], parts, part, wave);
    uint32_t ph_start = 0, ph_end = 0, general_pieces = 0, aside_count = 0, anchors = 0;
    if (!wr.empty) {
        const uint32_t hist_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
            (__attribute__((address_space(3))) uint32_t*)hist));
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
        uint4 r0, r1, r2, r3;
        auto load_piece = [&](uint32_t piece) {
            const uint32_t soff = piece * static_cast<uint32_t>(kPiece);
            const uint32_t lane64 = lane_now() << 6;  // the granule offsets ride on the scalar offset
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

        // context of the next granule or block to be counted: codes and BAD of the 16 positions before it
        // (round 6: in vector registers, of which LANE 0 is what counts: the heavy stage shifts it in with the same DPP move
        // that fetches every other lane's neighbour, and a full round leaves its lane 63 there with one wave_ror -- a
        // v_readlane into a scalar and a v_mov back out of it per value and round before)
        uint32_t ctx_c = 0u, ctx_bad = 0x55555555u;
        uint32_t pph = 0;     // line phase at the start of the current piece
        uint32_t npend = 0;   // granules waiting in xb[0 .. npend), < 64 between pieces
        bool hot = false;
        uint32_t tick = 0;    // calls of the low-complexity probe
        uint32_t ngeneral = 0;  // pieces that took the general path (reported in the wave's phase word, bits 8..31)
        uint32_t* const alist = aside + (static_cast<uint64_t>(unit) * kWaves + static_cast<uint32_t>(wave)) * aside_cap;
        uint32_t naside = 0;    // entries in alist
        uint32_t aside63 = 0;   // lane 63 of the piece before was set aside (wave-uniform)
        // INDEX: the wave's segment of the sample's anchor list
        uint32_t* const iseg = INDEX ? ip.anchors + uniform64(index_segment(ip, smp, w0, part, wave)) : nullptr;
        const uint32_t icap = INDEX ? static_cast<uint32_t>(index_segment_cap(w1 - w0)) : 0u;
        uint32_t nanch = 0, nsite = 0;   // anchors stored; this lane's sequence bytes so far
        bool ifull = false;

        // The heavy stage on one granule per lane (the first n lanes; the others idle along on a granule
        // of newlines): q = xb[lane].  probe: also look whether the data has turned low-complexity.
        auto round_count = [&](uint32_t n, uint4 q, bool probe, bool full = true) __attribute__((always_inline)) {
#ifdef VK_DIAG_NO_HEAVY
            asm volatile("" :: "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
            return;
#endif
            uint32_t C, bad, SEQ;   // (bad: even bits; the odd ones hold garbage that ok_mask1 drops)
            vkl::classify_granule_note(q.x, q.y, q.z, q.w, C, bad, SEQ);
            if constexpr (INDEX) nsite += vkl::popc(SEQ & 0x55555555u);
            uint32_t badh = ctx_bad, ch = ctx_c;   // (lane 0 keeps these: wave_shr:1 has no source for it)
            if (full) {   // n = 64: the neighbours, and lane 63 into lane 0 of the context (wave_ror:1) -- in this order, by hand:
                          // hipcc put the rotation first and paid a copy of the old context for it
                uint32_t nb, nc;
                asm("s_nop 1\n\t"
                    "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_mov_b32_dpp %2, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                    "v_mov_b32_dpp %3, %5 wave_ror:1 row_mask:0xf bank_mask:0xf"
                    : "+v"(badh), "+v"(ch), "=&v"(nb), "=&v"(nc) : "v"(bad), "v"(C));
                ctx_bad = nb;
                ctx_c = nc;
            } else {
                badh = wave_prev_lane(bad, ctx_bad);
                ch = wave_prev_lane(C, ctx_c);
                ctx_bad = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bad), static_cast<int>(n - 1u)));
                ctx_c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(C), static_cast<int>(n - 1u)));
            }
            const uint32_t ok = vkl::ok_mask1<K>(badh, bad);
            uint32_t pa;
            unsigned long long pm;
            windows_lds1<K>(ch, C, ok, hist_base, pa, pm);
            if (probe) hot = probe_low_complexity(pa, pm, tick);
        };
        auto flush = [&]() __attribute__((always_inline)) {  // the pending granules, before a piece takes the general path
            uint4 q = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au | vkl::kGranuleEnd, 0x0A0A0A0Au, 0x0A0A0A0Au);   // (note: no sequence bytes)
            const uint32_t ln = lane_now();
            if (ln < npend) q = xb[ln];
            round_count(npend, q, true, false);
            npend = 0u;
        };

        load_piece(0);
#ifdef VK_STAMPS
        unsigned long long st_wait = 0, st_all = 0, st_t0 = 0, st_t1 = 0, st_prev = 0;
        VK_STAMP(st_prev);
#endif
#ifndef VK_DIAG_K1_NO_PACE
        const uint32_t pace_at = npieces - (npieces >> 2);   // (>= 1: the first piece of a range of under four runs at 1 too)
        __builtin_amdgcn_s_setprio(1);
#endif
        for (uint32_t it = 0; it < npieces; ++it) {
#ifndef VK_DIAG_K1_NO_PACE
            // A SIMD issues for its oldest ready wavefront first, and of a workgroup's four wavefronts on a SIMD the same one is
            // always the oldest: equal ranges ended apart, and a wave slot whose wavefront is done waits for the workgroup's last
            // (the inflate chunk decoder, vk_inflate.h, showed the effect at its largest).  The first three quarters of a range
            // run at priority 1, the last at 0: who is behind passes who is ahead.  -1 % (60.7 -> 60.0 ms; four levels the same,
            // the levels the other way round nothing: profiles/ab/r06_inflate_pacing.txt).
            if (it == pace_at) __builtin_amdgcn_s_setprio(0);
#endif
#ifdef VK_STAMPS
            VK_STAMP(st_t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            VK_STAMP(st_t1);
            st_wait += st_t1 - st_t0;
            st_all += st_t0 - st_prev;
            st_prev = st_t0;
#endif
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
            // ---- line pass ----
            bool fast = it != 0 && it + 1 != npieces && !hot && !__any(vkl::ascii_or(d) != 0u);
            uint32_t total = 0, s = 64, e = 64;
            if (fast) {
                uint32_t mlo, mhi;
                vkl::newline_mask64(d, mlo, mhi);
                const uint32_t c = vkl::popc(mlo) + vkl::popc(mhi);
                const uint32_t incl = wave_inclusive_sum(c);
                total = lane_bcast(incl, 63);
                const uint32_t dn = (c - incl + (1u - pph)) & 3u;   // line ends to pass before a sequence line starts: (1 - line phase) & 3
                uint32_t s_raw;
                // (one note per lane: a stretch that starts inside a granule AND ends in this lane -- a read under 64 bases --
                // is set aside like the shapes seq_span itself refuses)
                const bool plain = vkl::seq_span_note(mlo, mhi, c, dn, s, e, s_raw);
                const unsigned long long am = __builtin_amdgcn_ballot_w64(!plain);   // (the compare's own lane mask: __ballot() makes a 0 / 1 of it and compares again)
                if (am != 0ull) {   // rare: lanes set aside (see above), or too many of them
                    const uint32_t na = static_cast<uint32_t>(__builtin_popcountll(am));
#ifdef VK_DIAG_NO_ASIDE   // diagnostic build: every such piece down the general path, as before round 4
                    if (true) {
#else
                    if (na > kSetAside || naside + na > aside_cap) {
#endif
                        fast = false;
                    } else {
                        if (!plain) {
                            const uint32_t ln = lane_now();
                            const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(am >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(am), 0u));   // lanes set aside below this one
                            const bool before = ln == 0u ? aside63 != 0u : ((am >> (ln - 1u)) & 1ull) != 0ull;
                            // (a buffer store: with a 64-bit address in registers the branch spilled two register pairs and
                            // waited for them -- and for this store -- with vmcnt(0): 6 % of wave time on fastp-shaped reads)
                            const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(alist, 0, static_cast<int>(aside_cap * 4u), 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b32(((it * 64u + ln) << 3) | (before ? 4u : 0u) | ((1u - dn) & 3u), arsrc, (naside + below) * 4u, 0, 0);
                            // the separator: the lane's first granule as it is, under the note "the line ends at position 0" --
                            // no sequence bytes, whatever the bytes are (until round 6 the granule was overwritten with
                            // newlines: a write to the piece's registers in a branch, four moves on every piece's way)
                            s = 0u;
                            e = 0u;
                        }
                        naside = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(naside + na)));   // (kept in a scalar register)
                    }
                }
                aside63 = fast ? static_cast<uint32_t>(am >> 63) : 0u;   // (scalar: am is a ballot, fast wave-uniform)
                if constexpr (INDEX) {
                    // a plain lane holds one anchor at most: the newline its sequence line begins behind (a lane set aside
                    // leaves its anchors to vk_aside_kernel, a piece that goes the general way after all to general_piece)
                    const bool has = fast && plain && s_raw >= 1u && s_raw <= 64u;
                    const unsigned long long hm = __ballot(has);
                    if (hm != 0ull) {
                        const uint32_t nh = static_cast<uint32_t>(__builtin_popcountll(hm));
                        const uint32_t ln = lane_now();
                        if (nanch + nh > icap) ifull = true;
                        else if (has) {   // (a buffer store, as for the lanes set aside: no 64-bit address in vector registers)
                            const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(iseg, 0, static_cast<int>(icap * 4u), 0x00020000);
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hm), 0u));
                            __builtin_amdgcn_raw_buffer_store_b32(static_cast<uint32_t>(o0) + it * static_cast<uint32_t>(kPiece) + 64u * ln + s_raw - 1u,
                                                                  irsrc, (nanch + rank) * 4u, 0, 0);
                        }
                        if (!ifull) nanch += nh;
                    }
                }
            }
            if (!fast) aside63 = 0u;
            // The rounds of the piece that are counted behind the next piece's loads: none (mode 0), qb (1), qa and qb (2).
            // One load site for every way through the iteration: with one per branch the piece's sixteen registers met in
            // phi nodes and the back edge carried a dozen moves.
            uint32_t mode = 0u, tot = 0u;
            uint4 qa = make_uint4(0u, 0u, 0u, 0u), qb = qa;
            if (!fast) {
                // ---- general path (a function of its own: inlined, its register needs -- all 64 bytes
                // classified at once -- would spill the fast path's loop invariants) ----
                if (npend != 0u) flush();
                ++ngeneral;
                GeneralState gs;
                gs.ctx_c = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ctx_c)));
                gs.ctx_bad = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ctx_bad)));
                gs.pph = pph; gs.hot = hot ? 1u : 0u; gs.tick = tick;
                gs.nanch = nanch; gs.full = 0u; gs.sites = 0u;
                const uint32_t valid = (it + 1 == npieces && tail_bytes != 0u) ? tail_bytes : static_cast<uint32_t>(kPiece);
#ifdef VK_DIAG_NO_GENERAL   // timing only (results wrong): the loop without its one function call -- the piece's newlines still counted, so that the line phase stays right
                {
                    const uint32_t c = nl_count16(q0) + nl_count16(q1) + nl_count16(q2) + nl_count16(q3);
                    const uint32_t incl = wave_inclusive_sum(c);
                    if (it == 0) gs.pph = has_pre ? ph0 - lane_bcast(c, 0) : 0u;
                    gs.pph += lane_bcast(incl, 63);
                    gs.ctx_bad = 0x55555555u;
                }
#else
                gs = general_piece<K, INDEX>(q0, q1, q2, q3, gs, (it == 0 ? 1u : 0u) | (has_pre ? 2u : 0u) | (ph0 << 2) | (valid << 4),
                                             hist, hist_base, iseg, icap, static_cast<uint32_t>(o0) + it * static_cast<uint32_t>(kPiece));
#endif
                if constexpr (INDEX) {
                    nanch = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gs.nanch)));
                    if (__builtin_amdgcn_readfirstlane(static_cast<int>(gs.full)) != 0) ifull = true;
                    nsite += gs.sites;
                }
                ctx_c = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gs.ctx_c)));
                ctx_bad = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gs.ctx_bad)));
                pph = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gs.pph)));
                hot = __builtin_amdgcn_readfirstlane(static_cast<int>(gs.hot)) != 0;
                tick = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(gs.tick)));
                total = 0u;   // (pph is the general path's)
            } else {
                // ---- hand the granules with sequence bytes to the heavy stage, 64 at a time ----
                // (s = 64 comes with e = 64: gs = 4, one past the last granule = 4, n = 0)
                const uint32_t gs = vkl::span_first(s), n = ((vkl::umin(e, 63u) + 16u) >> 4) - gs;
                const uint32_t incl = wave_inclusive_sum(n);
                tot = npend + lane_bcast(incl, 63);
                const uint32_t mine0 = npend + incl - n;   // place in the stream (64 per round) of this lane's first granule
                const uint32_t first = mine0 - gs;         // + g = the place of its granule g
                // round and buffer address of every granule of this lane (round kNoRound: none of its rounds)
                constexpr uint32_t kNoRound = 0xFFFFFFFFu;   // (an inline constant; so is the "no note" place below: -1 >> 6 is no round either)
                const uint32_t xb_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint4*)xb));
                uint32_t rd[5], ad[5];
                // (granule g is this lane's iff bit g of ((1 << n) - 1) << gs is set: one v_bfm, then a sign-extended bit per
                // slot -- all ones on a slot that is not -- instead of a subtraction, a comparison and a select each)
                uint32_t minebits;
                asm("v_bfm_b32 %0, %1, %2" : "=v"(minebits) : "v"(n), "v"(gs));   // ((1 << n) - 1) << gs; n, gs <= 4
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g) {
                    const uint32_t place = first + g;
                    uint32_t own;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(own) : "v"(minebits), "n"(g));   // all ones: this lane's
                    rd[g] = (place >> 6) | ~own;    // kNoRound on a slot that is not this lane's
                    ad[g] = vkl::lshl_add_s(place & 63u, 4, xb_base);
                }
                // The note of the stretch's edge granule (vk_lane.h, classify_granule_note) is set on the granule's copy in the
                // buffer (the piece's registers stay as loaded: four aligned 128-bit tuples the stores can take as they are):
                // where the first base is (s & 15 != 0), else where the line ends (e < 64); never both (span_one_note), so
                // one of s & 15 and e & 15 is zero (a stretch that starts inside a granule runs on: e = 64).
                const bool has_s = (s & 15u) != 0u;        // (s = 64: no)
                const uint32_t pnote = has_s ? mine0 : (e < 64u ? mine0 + n - 1u : 0xFFFFFFFFu);
                rd[4] = pnote >> 6;
                ad[4] = vkl::lshl_add_s(pnote & 63u, 4, xb_base);
                const uint32_t note_lo = vkl::note_spread((s | e) & 15u), note_hi = has_s ? 0u : vkl::kGranuleEnd;
                const uint32_t rounds = tot >> 6;
                // put(r): the granules whose round is r go into the buffer, the note behind them.  Hand-written (the
                // compiler's version: a compare, a saved exec mask, a branch around the store and a restored mask per
                // slot -- ~46 scalar instructions and 15 branches per piece): five lane masks, then five {exec; LDS op}.
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                auto put = [&](uint32_t r) __attribute__((always_inline)) {
                    const u32x4 g0 = {r0.x, r0.y, r0.z, r0.w}, g1 = {r1.x, r1.y, r1.z, r1.w};
                    const u32x4 g2 = {r2.x, r2.y, r2.z, r2.w}, g3 = {r3.x, r3.y, r3.z, r3.w};
                    const u32x2 nt = {note_lo, note_hi};
                    unsigned long long m0, m1, m2, m3, m4;
                    const unsigned long long exec_in = __builtin_amdgcn_read_exec();
                    asm volatile(
                        "v_cmp_eq_u32_e64 %0, %5, %15\n\t"
                        "v_cmp_eq_u32_e64 %1, %6, %15\n\t"
                        "v_cmp_eq_u32_e64 %2, %7, %15\n\t"
                        "v_cmp_eq_u32_e64 %3, %8, %15\n\t"
                        "v_cmp_eq_u32_e64 %4, %9, %15\n\t"
                        "s_mov_b64 exec, %0\n\tds_write_b128 %10, %16\n\t"
                        "s_mov_b64 exec, %1\n\tds_write_b128 %11, %17\n\t"
                        "s_mov_b64 exec, %2\n\tds_write_b128 %12, %18\n\t"
                        "s_mov_b64 exec, %3\n\tds_write_b128 %13, %19\n\t"
                        "s_mov_b64 exec, %4\n\tds_or_b64 %14, %20\n\t"
                        "s_mov_b64 exec, %21"
                        : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4)
                        : "v"(rd[0]), "v"(rd[1]), "v"(rd[2]), "v"(rd[3]), "v"(rd[4]),
                          "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "s"(r),
                          "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(nt), "s"(exec_in)
                        : "memory");
                };
                // The last two rounds' granules are taken out of the buffer first, so that the piece's registers are
                // free -- and the next piece's loads in flight -- under the arithmetic of BOTH rounds (-0.5 % on 512
                // distinct samples; the 9 % such a launch loses against one on 64 samples, whose lines the
                // workgroups of an XCD share in L2, is not the loads' latency).
                uint32_t r = 0u;
                for (; r + 2u < rounds; ++r) {
                    put(r);
                    const uint4 q = xb[lane];
                    round_count(64u, q, false);
                }
                if (rounds >= 2u) {
                    put(r);
                    qa = xb[lane];
                    ++r;
                    mode = 2u;
                }
                if (rounds >= 1u) {
                    put(r);
                    qb = xb[lane];
                    ++r;
                    if (mode == 0u) mode = 1u;
                }
                put(r);               // what is left stays in the buffer for the next piece
            }
            if (it + 1 < npieces) load_piece(it + 1);
            if (mode == 2u) round_count(64u, qa, false);
            if (mode != 0u) round_count(64u, qb, true);
            npend = tot & 63u;
            pph += total;
        }
        ph_end = pph & 3u;
        general_pieces = ngeneral < 0xFFFFFFu ? ngeneral : 0xFFFFFFu;
        aside_count = naside;
#ifdef VK_DIAG_ASIDE_INLINE
        // DIAGNOSTIC: round 4's deferred count of the lanes set aside, inlined behind the piece loop instead of a
        // kernel of its own -- the arrangement that made the K = 5 build lose the loop's own counts.  Not shipped.
        if constexpr (!INDEX) {
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the list's stores
            const uint32_t* list = aside + static_cast<uint64_t>(unit * kWaves + static_cast<uint32_t>(wave)) * aside_cap;
            const uint32_t ulane = static_cast<uint32_t>(lane);
            const uint32_t ent = ulane / 3u, j = ulane - 3u * ent;
            for (uint32_t at = 0; at < naside; at += kSetAsideBatch) {
                const uint32_t n = naside - at < kSetAsideBatch ? naside - at : kSetAsideBatch;
                const bool live = ent < n;
                const uint32_t w = live ? list[at + ent] : 0u;
                const uint32_t off = live ? ((w >> 3) - 1u + j) * 64u : 0xFFFFFF00u;
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 16, 0);
                const u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 32, 0);
                const u32x4 e4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 48, 0);
                const uint32_t d[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x, c4.y, c4.z, c4.w, e4.x, e4.y, e4.z, e4.w};
                vkl::LaneBits lb;
                const uint32_t c = vkl::classify<false>(d, lb);
                const uint32_t cprev = wave_prev_lane(c, 0u);
                const uint32_t lph = ((w & 3u) + (j == 0u ? 0u - c : (j == 2u ? cprev : 0u))) & 3u;
                vkl::Mask128 seq;
                uint32_t s_raw = 0;
                if (__any(c > 4u)) seq = vkl::seq_mask_general(lb.NL, lph);
                else seq = vkl::seq_mask_fast4(lb.NL, lph, vkl::ones_below, vkl::ones_not_below, s_raw);
                uint32_t bad[4], ok[4];
                vkl::bad_mask(lb, seq, bad);
                const uint32_t badh = wave_prev_lane(bad[3], 0x55555555u);
                const uint32_t ch = wave_prev_lane(lb.C[3], 0u);
                vkl::ok_mask<K>(badh, bad, ok);
                constexpr uint32_t kBack = (1u << (2 * (K - 1))) - 1u;
                if (!live || j == 0u) { ok[0] = 0u; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }
                if (j == 1u && (w & 4u) != 0u) ok[0] &= ~kBack;
                if (j == 2u) { ok[0] &= kBack; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }
                uint32_t pa;
                unsigned long long pm;
                windows_lds<K>(ch, lb.C, ok, hist_base, pa, pm);
            }
        }
#endif
        if constexpr (INDEX) {
            anchors = nanch;
            const uint32_t tot = lane_bcast(wave_inclusive_sum(nsite), 63);
            if (lane == 0) {
                if (tot != 0u) atomicAdd(&ip.sites[smp], static_cast<unsigned long long>(tot));
                if (ifull || len >= (1ull << 32)) atomicOr(&ip.overflow[smp], 1u);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
#ifdef VK_STAMPS
        if (lane == 0 && (blockIdx.x & 63u) == 0u) {  // [5] wait for the piece's bytes, [6] whole iterations, [7] pieces
            atomicAdd(&g_vk_stamps[5], st_wait);
            atomicAdd(&g_vk_stamps[6], st_all);
            atomicAdd(&g_vk_stamps[7], static_cast<unsigned long long>(npieces));
        }
#endif
    }
    if (lane == 0) {
        wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2) | (general_pieces << 8));
        aside_n[unit * kWaves + wave] = aside_count;   // (vk_aside_kernel counts the listed lanes)
        if constexpr (INDEX) ip.count[unit * kWaves + wave] = anchors;
    }

    __syncthreads();
    uint32_t* out = hist_out + static_cast<uint64_t>(smp) * NCODE;
    for (uint32_t code = tid; code < NCODE; code += kCountThreads) {   // (over the output codes, as in vk_count_kernel)
        const uint32_t v = hist[pair_reverse(code, K)];
        if (atomic_flush) {
            if (v) atomicAdd(&out[code], v);
        } else {
            out[code] = v;
        }
    }
}

// ---- the packed sequence stream (written by vk_pack_kernel, vk_pack.h) --------------------------------
constexpr uint32_t kPackHasPre = 0x80000000u;   // count word: the segment opens with a record of context
constexpr uint32_t kPackBlock = 256;            // records a consumer wavefront takes at a time (four per lane)

struct PackParams {
    uint32_t* c;            // codes of every record of the launch
    uint32_t* m;            // masks: even bits BAD, odd bits SITE
    const uint64_t* base;   // [nsamples] first record of the sample's region (a multiple of 4)
    uint32_t* count;        // [nsamples * parts * kWaves] records of the wave's segment | kPackHasPre
};

// first record of the segment of wave `wave` of workgroup `part` (a multiple of 4: the consumers load four records per lane)
__device__ __forceinline__ uint64_t pack_segment(const PackParams& pk, uint32_t smp, uint32_t part, int wave, uint64_t w0) {
    return pk.base[smp] + (w0 >> 4) + 8ull * (static_cast<uint64_t>(part) * kWaves + static_cast<uint32_t>(wave));
}

// ---- K = 8, 9: the LDS-spill path ------------------------------------------------------
// 4^K u32 counters do not fit LDS.  Pass A (vk_bucket_kernel) streams the FASTQ exactly like
// vk_count_kernel, but instead of counting it PARTITIONS the windows into 16 bucket streams per
// sample; pass B (vk_bucket_count_kernel) gives every (sample, bucket) one workgroup that replays
// its stream into a 2 x 4^K/16-bin LDS histogram and adds that to the global one.
//
// Two windows that end at neighbouring positions p, p + 1 (p even) overlap in K - 1 bases; the
// bases p-1 and p, which both contain, are the bucket number q (4 bits) of the PAIR, and what is
// left of the K + 1 bases of the pair is one u16 entry:
//     bits [0, LB)      bases p-K+1 .. p-2   (LB = 2K - 4)
//     bits [LB, LB+2)   base  p+1
// i.e. ONE byte of bucket traffic per window (written once, read once).  Pairs of which only one
// window is countable (a read's first or last window, the neighbours of an N: ~1.3 per read) are
// "singles" and are counted with a global atomic on the spot, as is everything that does not fit a
// queue or the arena (low-complexity input) -- slower, still exact.
//
// Queues: 16 per wavefront in LDS, 128 entries each; a queue is drained in 64-byte blocks (32
// entries), all 16 queues at once (four lanes per queue), whenever one holds two blocks.  Blocks go
// to RUNS of 4 KiB that a wave takes from the sample's arena with one global atomic; a closed run's
// header word says which bucket it belongs to and how many of its blocks are filled, so no stream
// has a fixed capacity: however skewed the base composition, the arena holds all pairs of the
// sample (at most len / 4 of them) plus one open run per (wave, queue).
constexpr uint32_t kQueues = 16;          // queues per wave = bucket streams per sample
constexpr uint32_t kQueueBytes = 256;     // 128 u16 entries
constexpr uint32_t kQueueShift = 6;       // log2(kQueueBytes / 4): data offset = counter offset << 6
constexpr uint32_t kBlockBytes = 64;      // drain unit
constexpr uint32_t kQueueBlocks = kQueueBytes / kBlockBytes;
constexpr uint32_t kRunBytes = 4096;      // arena allocation unit
constexpr uint32_t kRunBlocks = kRunBytes / kBlockBytes;
// Inside a piece the queues are drained when one holds this many bytes: three blocks while the wave has not seen a queue
// overflow, two from its first overflow on.  Three for good is -1.3 % on uniform bases and -5 % on fastp-shaped reads but
// +20 % on GC-skewed ones, whose queues overflow into global atomics (profiles/ab/r04_k9_drain_threshold.txt).
constexpr uint32_t kDrainAt = 2 * kBlockBytes;
constexpr uint32_t kDrainAtCalm = 3 * kBlockBytes;
constexpr uint32_t kNoRun = 0x100;        // "blocks used" of a queue that has no run open

struct BucketParams {
    uint32_t* cursors;   // [nsamples] next free run of the sample's arena
    uint32_t* hdrs;      // [nsamples][runs_cap] 0x80000000 | filled blocks << 8 | bucket, 0 = never closed
    uint8_t* arena;      // [nsamples][runs_cap][kRunBytes]
    uint32_t* bucket_hist;  // [nsamples][16][2 * 4^K / 16] pass B's counters, merged into the histogram by pass C
    uint32_t* bsize;     // [nsamples][16] blocks of each bucket stream in the arena (added up as runs are closed)
    uint32_t* order;     // [nsamples * 16] pass B's (sample, bucket) jobs, the large ones first
    uint32_t* wide;      // [nsamples * 16] != 0: the job's u16 pair counters overflowed, bucket_hist holds u32 window counters instead
    uint32_t runs_cap;
    uint32_t force_wide; // tests: every job through the u32 replay
    // the quad route (MODE 3 of vk_bucket_kernel; bsize = runs per bucket, [nsamples][kQuadBuckets]; wide is not used by it):
    uint32_t* qfirst;    // [nsamples][kQuadBuckets + 1] first entry of every bucket's runs in qlist (vk_quad_list_kernel)
    uint32_t* qlist;     // [nsamples][runs_cap] closed runs sorted by bucket: run | filled blocks << 24
    uint32_t* preg;      // [grid][kQuadBuckets][1 << preg_shift] quads of which only some windows count, by pass A workgroup and bucket: K + 3 bases | OK bits << 24
    uint32_t* preg_n;    // [grid][kQuadBuckets] entries in each region
    uint32_t preg_shift; // log2 of a region's capacity
    uint32_t parts;      // workgroups of pass A per sample
};

// LDS of the bucket kernel, one array with fixed offsets (the hand-written queue appends address it
// with immediate offsets, so it must be the kernel's only LDS object and start at 0: checked at run time)
constexpr uint32_t kLdsCnt = 0;                                   // u32 [kWaves][16] queue fill in bytes
constexpr uint32_t kLdsRun = kLdsCnt + kWaves * kQueues * 4;      // u32 [kWaves][16] current run of the queue
constexpr uint32_t kLdsUsed = kLdsRun + kWaves * kQueues * 4;     // u32 [kWaves][16] blocks used in it (kNoRun = none open)
constexpr uint32_t kLdsScratch = kLdsUsed + kWaves * kQueues * 4; // u64 [kWaves][8]
constexpr uint32_t kLdsBelow = kLdsScratch + kWaves * 64;         // uint4 [66]
constexpr uint32_t kLdsAbove = kLdsBelow + 66 * 16;               // uint4 [66]
constexpr uint32_t kLdsQueues = 8192;                             // u16 [kWaves][16][128]
constexpr uint32_t kLdsXchg = kLdsQueues + kWaves * kQueues * kQueueBytes;  // uint2 [kWaves][64]: (codes, OK) of lane groups on their way to the append stage
constexpr uint32_t kLdsBucketBytes = kLdsXchg + kWaves * 64 * 8;
constexpr uint32_t kLdsPool = kLdsAbove + 66 * 16;                // u32 [kWaves][2]: the wave's reserve of arena runs, [next, end)
constexpr uint32_t kPoolRuns = 16;        // runs a wave takes from the arena with one returning atomic
constexpr uint32_t kLdsSync = kLdsPool + kWaves * 8;              // u32 [2] (room for 6): the quad route's (rounds | tight << 16) of a step, agreed by the workgroup
constexpr uint32_t kLdsCntP = kLdsSync + 24;                      // u32 [256]: the quad route's listed quads per bucket (this workgroup's)
static_assert(kLdsCntP + 256 * 4 <= kLdsQueues, "LDS layout");
constexpr uint32_t kQuadBuckets = 256;    // the quad route's bucket streams per sample = kWaves * kQueues queues per WORKGROUP
static_assert(kQuadBuckets == kWaves * kQueues, "a wave drains sixteen of its workgroup's queues");
static_assert(2 * kLdsBucketBytes <= 160 * 1024, "two bucket workgroups per CU");  // = exactly 160 KiB

// Raw window field (first base least significant) of the window of type t (0: ends at p, 1: ends at
// p + 1) of a pair entry e of bucket q:  t = 0: rest | q << LB;  t = 1: rest without its first base,
// then q, then the base p+1.  `idx` = the 14-bit (K = 9) index pass B counts it under: rest for
// type 0, (rest >> 2) | last << (LB - 2) for type 1.
template <int K>
__device__ __forceinline__ uint32_t entry_raw(uint32_t q, uint32_t idx_and_type) {
    constexpr uint32_t LB = 2 * K - 4;
    const uint32_t idx = idx_and_type & ((1u << LB) - 1u);
    if ((idx_and_type >> LB) == 0u) return (q << LB) | idx;
    return (idx & ((1u << (LB - 2)) - 1u)) | (q << (LB - 2)) | ((idx >> (LB - 2)) << (2 * K - 2));
}

// both windows of the pair entry e (u16, upper garbage bits allowed for K = 8) of bucket q, counted directly
template <int K>
__device__ __forceinline__ void count_entry_direct(uint32_t* hist_s, uint32_t q, uint32_t e) {
    constexpr uint32_t LB = 2 * K - 4;
    const uint32_t rest = e & ((1u << LB) - 1u), last = (e >> LB) & 3u;
    atomicAdd(&hist_s[pair_reverse(entry_raw<K>(q, rest), K)], 1u);
    atomicAdd(&hist_s[pair_reverse(entry_raw<K>(q, (1u << LB) | (rest >> 2) | (last << (LB - 2))), K)], 1u);
}

// ---- the quad route (vk_bucket_kernel<K, 3>): ONE u16 entry per FOUR windows ----------------------------
// The windows ending at p .. p + 3 (p a multiple of 4 in the file) span K + 3 bases b0 .. b(K+2) and all hold the four
// bases b(K-4) .. b(K-1): those are the bucket number (8 bits, 256 streams per sample), and what is left is the entry:
//     bits [0, 2K - 8)        head: b0 .. b(K-5)
//     bits [2K - 8, 2K - 2)   tail: b(K), b(K+1), b(K+2)
// -- half a byte of bucket traffic per window.  256 queues do not fit a wavefront's share of LDS, so the queues belong
// to the WORKGROUP (256 x 128 entries, the same 64 KiB): every wave appends to all of them with returning LDS atomics
// and drains sixteen of them (wave w: queues 16 w .. 16 w + 15) at points the workgroup agrees on, between two
// barriers each -- measured free (profiles/ab/r05_k9_append_diagnostics.txt) -- one before every round of appends.
// Pass B (vk_quad_count_kernel) counts an entry as two PAIRS (windows 0, 1: the entry's low 2K - 6 bits; windows 2, 3:
// its high 2K - 6 bits) into two u32 tables of 4^(K-3) counters per bucket; vk_quad_merge_kernel sums per k-mer code.
// Quads of which only some windows count (a read's first and last, the neighbours of an N: ~1.5 per read) travel as
// u32 words (K + 3 bases | OK bits << 24) through a list per wavefront in HBM and are counted with global atomics by
// spare workgroups of pass B's launch.
// x = the K + 3 bases of a quad (2 bits each, first base lowest; garbage above allowed where noted)
template <int K>
__device__ __forceinline__ uint32_t quad_bases(uint32_t q, uint32_t e) {   // bucket q, entry e (garbage above bit 2K - 2 allowed)
    constexpr uint32_t HB = 2 * K - 8;
    return (e & ((1u << HB) - 1u)) | (q << HB) | (((e >> HB) & 63u) << (2 * K));
}
// the windows of a quad named by the OK bits `okb` (bit 2t: window t), counted directly (n times)
template <int K>
__device__ __forceinline__ void count_quad_direct(uint32_t* hist_s, uint32_t x, uint32_t okb, uint32_t n = 1u) {
    constexpr uint32_t FMASK = (1u << (2 * K)) - 1u;
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t)
        if ((okb >> (2u * t)) & 1u) atomicAdd(&hist_s[pair_reverse((x >> (2u * t)) & FMASK, K)], n);
}
// The same for the lanes of a wavefront that hold a whole quad (`have`), equal quads counted ONCE with their number:
// what a full queue sends here is low-complexity input, where 64 lanes hold a handful of different quads -- and 64
// global atomics on one counter are carried out one after the other at the memory side.  Up to eight classes; what
// is left after that counts alone.  All lanes call it.
// (a function of its own: inlined at the append sites it cost the piece loop 18 vector instructions per piece in moves)
template <int K>
__device__ __attribute__((noinline)) void count_quads_aggregated(uint32_t* hist_s, bool have, uint32_t x) {
    const uint32_t key = x & ((1u << (2 * K + 6)) - 1u);
    unsigned long long left = __ballot(have);
    for (int t = 0; t < 8 && left != 0ull; ++t) {
        const int src = __builtin_ctzll(left);
        const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), src));
        const unsigned long long eq = __ballot(have && key == first) & left;
        if (lane_now() == static_cast<uint32_t>(src)) count_quad_direct<K>(hist_s, first, 0x55u, static_cast<uint32_t>(__builtin_popcountll(eq)));
        left &= ~eq;
    }
    if ((left >> lane_now()) & 1ull) count_quad_direct<K>(hist_s, key, 0x55u);
}

// A lane group (codes [lo | hi], OK string okg) of a low-complexity stretch, before its quads are appended: quads that
// SEVERAL lanes hold alike -- whole or not: the read ends the repeat shortcut leaves over are the same few quads in a
// fifth of the lanes -- are counted here, once per class with their number, and taken out of okg; what one lane holds
// alone goes on to the queues.  (Sent there, 16 waves x 64 lanes fill the handful of queues of those quads within a
// round: returning LDS atomics on one counter, a drain point per round, and the overflow counted anyway.)  All lanes call it.
template <int K>
__device__ __attribute__((noinline)) uint32_t count_hot_quads(uint32_t* hist_s, uint32_t lo, uint32_t hi, uint32_t okg) {
    const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
#pragma nounroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t okb = (okg >> (8u * j)) & 0x55u;
        const uint32_t o = (32u - 2u * (K - 1) + 8u * j) & 63u;   // bit offset of the quad's first base in [lo | hi]
        const uint32_t key = (static_cast<uint32_t>(v >> o) & ((1u << (2 * K + 6)) - 1u)) | (okb << 24);
        unsigned long long left = __ballot(okb != 0u);
        uint32_t alone = 0;
        for (int t = 0; t < 6 && left != 0ull && alone < 2u; ++t) {
            const int src = __builtin_ctzll(left);
            const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), src));
            const unsigned long long eq = __ballot(okb != 0u && key == first) & left;
            const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(eq));
            left &= ~eq;
            if (n == 1u) {   // ordinary reads among the repeats: two of those and the rest is left alone
                ++alone;
                continue;
            }
            if (lane_now() == static_cast<uint32_t>(src)) count_quad_direct<K>(hist_s, first & 0xFFFFFFu, first >> 24, n);
            if ((eq >> lane_now()) & 1ull) okg &= ~(0x55u << (8u * j));
        }
    }
    return okg;
}

__device__ __forceinline__ uint32_t quad_bcast0(uint32_t x) {  // value of lane (lane & ~3)
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(x), 0x00, 0xF, 0xF, true));
}

// MODE 0: the FASTQ text through vk_count_kernel's front end (every byte classified); 1: the same with read
// subsampling; 2: the packed stream of vk_pack_kernel (no line logic, no classification here); 3: MODE 0's front end
// with quad entries in workgroup-shared queues (see above: the shipped pass A).
template <int K, int MODE>
__global__ __launch_bounds__(kCountThreads, MODE == 1 ? 4 : VK_K1_OCC) void vk_bucket_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
    const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ wavephase, BucketParams bp, SubParams sp, PackParams pk) {
    constexpr bool SUB = MODE == 1, STREAM = MODE == 2, QUAD = MODE == 3;
#ifndef VK_QUAD_CAP_BYTES
#define VK_QUAD_CAP_BYTES kQueueBytes   // (timing experiments: a smaller capacity in the same 256-byte slots)
#endif
    constexpr uint32_t kCap = QUAD ? static_cast<uint32_t>(VK_QUAD_CAP_BYTES) : kQueueBytes;
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr uint32_t NQ = QUAD ? kQuadBuckets : kQueues;   // bucket streams per sample
    constexpr uint32_t LB = 2 * K - 4;               // bits of an entry that come from the shared prefix
    constexpr uint32_t LMASK = (1u << LB) - 1u;
    constexpr uint32_t FMASK = (1u << (2 * K)) - 1u;
    static_assert(LB + 2 <= 16, "entries are u16");

    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsBucketBytes / 4];
    auto lds_addr = [](const void* p) {
        return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p));
    };
    if (lds_addr(lds) != 0u) __builtin_trap();  // see kLdsCnt: never taken, the array is the kernel's only LDS object
    uint8_t* const ldsb = reinterpret_cast<uint8_t*>(lds);
    uint32_t* const qcnt_all = reinterpret_cast<uint32_t*>(ldsb + kLdsCnt);
    uint32_t* const qrun_all = reinterpret_cast<uint32_t*>(ldsb + kLdsRun);
    uint32_t* const qused_all = reinterpret_cast<uint32_t*>(ldsb + kLdsUsed);
    uint64_t* const scratch_all = reinterpret_cast<uint64_t*>(ldsb + kLdsScratch);
    uint4* const below = reinterpret_cast<uint4*>(ldsb + kLdsBelow);
    uint4* const above = reinterpret_cast<uint4*>(ldsb + kLdsAbove);

    const uint32_t unit = blockIdx.x;
    const uint32_t s = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    fill_mask_tables(below, above, tid);
    if (lane < static_cast<int>(kQueues)) {
        qcnt_all[wave * kQueues + lane] = 0u;
        qrun_all[wave * kQueues + lane] = 0u;
        qused_all[wave * kQueues + lane] = kNoRun;
    }
    if (lane < 2) reinterpret_cast<uint32_t*>(ldsb + kLdsPool)[wave * 2 + lane] = 0u;   // an empty reserve
    if (tid < 6) reinterpret_cast<uint32_t*>(ldsb + kLdsSync)[tid] = 0u;
    if (QUAD && tid < static_cast<int>(kQuadBuckets)) reinterpret_cast<uint32_t*>(ldsb + kLdsCntP)[tid] = 0u;
    __syncthreads();

    const uint8_t* sbase = fastq + offs[s];
    const uint64_t len = lens[s];
    const WaveRange wr = wave_range(len, parts, part, wave);
    uint32_t* hist_s = hist_out + static_cast<uint64_t>(s) * NCODE;

    // Four lanes per queue: q = lane / 4, every lane moves 16 B of a 64-byte block.
    const uint32_t q = static_cast<uint32_t>(lane) >> 2, sub = static_cast<uint32_t>(lane) & 3u;
    uint32_t* const qcnt = qcnt_all + wave * kQueues;
    uint32_t* const qrun = qrun_all + wave * kQueues;
    uint32_t* const qused = qused_all + wave * kQueues;
    uint8_t* const qdata = ldsb + kLdsQueues + (wave * kQueues + q) * kQueueBytes;  // this lane's queue
    uint32_t* const cursor = bp.cursors + s;
    uint32_t* const hdrs = bp.hdrs + static_cast<uint64_t>(s) * bp.runs_cap;
    uint8_t* const arena = bp.arena + static_cast<uint64_t>(s) * bp.runs_cap * kRunBytes;
    const uint32_t runs_cap = bp.runs_cap;

    // Move the full blocks of every queue (nb = 0..4 of them, n = its fill in bytes) to the arena.
    // All lanes call it; nb and n are the same in the four lanes of a queue.
    auto drain_all = [&](uint32_t n, uint32_t nb_) __attribute__((always_inline)) {
        // (the queue's addresses are worked out again here from an opaque copy of the lane number: kept in
        // registers across the piece loop they were spilled, and every reload from scratch came with a wait for
        // all outstanding stores -- the blocks of one drain then went out one store round trip at a time)
        uint32_t lz = static_cast<uint32_t>(lane), nb = nb_;
        asm volatile("" : "+v"(lz));
        const uint32_t q = lz >> 2, sub = lz & 3u;
        const uint32_t bq = QUAD ? static_cast<uint32_t>(wave) * kQueues + q : q;   // the queue's bucket
        uint8_t* const qdata = ldsb + kLdsQueues + (static_cast<uint32_t>(wave) * kQueues + q) * kQueueBytes;
        uint32_t run = qrun[q], used = qused[q];
        const bool need = nb != 0u && used + nb > kRunBlocks;  // the blocks of one drain stay in one run
        if (need && sub == 0 && used <= kRunBlocks) {  // close the old run
            hdrs[run] = 0x80000000u | (used << 8) | bq;
            if constexpr (!QUAD) atomicAdd(&bp.bsize[s * kQueues + q], used);
        }
        // A queue that needs a new run takes it from the wave's reserve (kPoolRuns runs per returning atomic on the
        // sample's cursor): one run per atomic made four drains in ten wait out a round trip to the memory-side atomic
        // unit -- and, vmcnt being one counter, every store still in flight.  What is left of a reserve that cannot
        // serve a drain's requests is abandoned: a run that is never closed has no header and pass B skips it.
        uint32_t nrun = 0;
        const unsigned long long needm = __ballot(need && sub == 0);
        if (needm != 0ull) {
            uint32_t* const pool = reinterpret_cast<uint32_t*>(ldsb + kLdsPool) + static_cast<uint32_t>(wave) * 2u;
            const uint32_t want = static_cast<uint32_t>(__builtin_popcountll(needm));   // <= 16 = kPoolRuns
            uint32_t next = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pool[0])));
            uint32_t end = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pool[1])));
            if (next + want > end) {
                uint32_t got = 0;
                if (lz == 0u) got = atomicAdd(cursor, kPoolRuns);
                next = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(got)));
                end = next + kPoolRuns;
            }
            if (need && sub == 0)
                nrun = next + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(needm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(needm), 0u));
            if (lz == 0u) {
                pool[0] = next + want;
                pool[1] = end;
            }
        }
        nrun = quad_bcast0(nrun);
        if (need) {
            run = nrun;
            used = nrun < runs_cap ? 0u : kNoRun;  // arena exhausted: no run open, the blocks are counted directly
        }
        const bool store = used + nb <= kRunBlocks;
        const uint4* src = reinterpret_cast<const uint4*>(qdata) + sub;
        uint4* dst = reinterpret_cast<uint4*>(arena + static_cast<uint64_t>(run) * kRunBytes + used * kBlockBytes) + sub;
        // (the queue's first two blocks are read together, whether full or not -- a drain seldom finds more: one LDS
        // round trip, not one per block; all four at once cost a spilled quad at 64 registers)
        const uint4 blk0 = src[0], blk1 = src[4];
#pragma unroll
        for (uint32_t b = 0; b < kQueueBlocks; ++b) {
            if (b < nb) {
                const uint4 v = b == 0 ? blk0 : (b == 1 ? blk1 : src[b * 4u]);
                if (store) {
#ifdef VK_DIAG_K9_NO_STORE   // timing only: the drain with everything but its stores to the arena
                    asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(dst));
#else
                    dst[b * 4u] = v;
#endif
                } else {  // no room in the arena: count these entries directly (exact, slow; kept small)
                    const uint16_t* e16 = reinterpret_cast<const uint16_t*>(qdata) + b * (kBlockBytes / 2) + sub * 8u;
#pragma nounroll
                    for (int j = 0; j < 8; ++j) {
                        if constexpr (QUAD) count_quad_direct<K>(hist_s, quad_bases<K>(bq, e16[j]), 0x55u);
                        else count_entry_direct<K>(hist_s, q, e16[j]);
                    }
                }
            }
        }
        // move the remainder (< one block) to the front
        const bool tail = nb != 0u && nb < kQueueBlocks;
        uint4 keep = make_uint4(0, 0, 0, 0);
        if (tail) keep = src[nb * 4u];
        wave_lds_fence();
        if (tail) reinterpret_cast<uint4*>(qdata)[sub] = keep;
        if (nb != 0u && sub == 0) {
            qcnt[q] = n - nb * kBlockBytes;
            qrun[q] = run;
            qused[q] = store ? used + nb : used;
        }
        wave_lds_fence();
    };
    auto maybe_drain = [&](uint32_t at_least) __attribute__((always_inline)) {
        wave_lds_fence();
        uint32_t n = qcnt[q];
        if (n > kCap) n = kCap;  // appends beyond the capacity were counted directly
        if (__any(n >= at_least)) drain_all(n, n / kBlockBytes);
    };

    uint32_t ph_start = 0, ph_end = 0;
#ifdef VK_DIAG_K9_BARRIER
    uint32_t diag_nbar = 0;
    {
        uint32_t* const slot = reinterpret_cast<uint32_t*>(ldsb + kLdsPool) + kWaves * 2;   // (a free word behind the pools)
        if (tid == 0) *slot = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t np = wr.empty ? 0u : static_cast<uint32_t>((wr.w1 - (wr.w0 ? wr.w0 - 64 : 0)) / kPiece);
        if (lane == 0) atomicMin(slot, np);
        __syncthreads();
        diag_nbar = *slot;
    }
#endif
    // (quad route: the waves of a workgroup meet at barriers, so a wave without a range of its own goes through the block too)
    if (!wr.empty || QUAD) {
        const uint32_t wbase = QUAD ? 0u : static_cast<uint32_t>(wave) * (kQueues * 4u);  // this wave's counters, relative to kLdsCnt (quad route: the workgroup's)
        const uint32_t two = 2u;
        // Append four pairs.  x[j] = the K + 1 bases of pair j (2 bits each, first base lowest, garbage
        // above), f[j] != 0 <=> both windows of the pair are countable.  Straight-line code: the lane
        // predicates are parked in SGPR pairs, the four returning atomics on the queue counters go out
        // back to back, each append then waits only for its own counter value.  Returns, per lane, a mask
        // of the pairs that found their queue full (they are counted directly by the caller).
        auto append4 = [&](const uint32_t (&x)[4], const uint32_t (&f)[4]) __attribute__((always_inline)) -> uint32_t {
            uint32_t c[4], e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // counter offset: q * 4 + this wave's base; entry: shared-prefix bits | base p+1
#ifdef VK_DIAG_K9_NOCONFLICT    // timing only: the queue by the lane's number, two lanes per counter in every half-wave
                c[j] = ((static_cast<uint32_t>(lane) & 15u) << 2) | wbase;
#else
                if constexpr (QUAD) c[j] = (x[j] >> (2 * K - 10)) & 0x3FCu;   // bucket = bases K-4 .. K-1 of the quad
                else c[j] = ((x[j] >> (2 * K - 6)) & 0x3Cu) | wbase;
#endif
                if constexpr (QUAD) {
                    constexpr uint32_t HM = (1u << (2 * K - 8)) - 1u;
                    e[j] = (x[j] & HM) | ((x[j] >> 8) & ~HM);   // one v_bfi: the tail moves down over the eight bucket bits
                } else {
                    e[j] = (x[j] & LMASK) | ((x[j] >> 4) & ~LMASK);  // one v_bfi: base p+1 moves down over the four bucket bits (garbage above it)
                }
            }
            uint32_t a0, a1, a2, a3;
            unsigned long long m0, m1, m2, m3, k0, k1, k2, k3;
            const uint32_t cap = kCap;
            const unsigned long long exec_in = __builtin_amdgcn_read_exec();  // put back behind the block, whatever it was
            asm volatile(
                "v_cmp_ne_u32_e64 %[m0], 0, %[f0]\n\t"
                "v_cmp_ne_u32_e64 %[m1], 0, %[f1]\n\t"
                "v_cmp_ne_u32_e64 %[m2], 0, %[f2]\n\t"
                "v_cmp_ne_u32_e64 %[m3], 0, %[f3]\n\t"
                "s_mov_b64 exec, %[m0]\n\tds_add_rtn_u32 %[a0], %[c0], %[two]\n\t"
                "s_mov_b64 exec, %[m1]\n\tds_add_rtn_u32 %[a1], %[c1], %[two]\n\t"
                "s_mov_b64 exec, %[m2]\n\tds_add_rtn_u32 %[a2], %[c2], %[two]\n\t"
                "s_mov_b64 exec, %[m3]\n\tds_add_rtn_u32 %[a3], %[c3], %[two]\n\t"
                "s_mov_b64 exec, %[m0]\n\ts_waitcnt lgkmcnt(3)\n\t"
                "v_cmp_gt_u32_e64 %[k0], %[cap], %[a0]\n\tv_lshl_add_u32 %[a0], %[c0], 6, %[a0]\n\t"
                "s_mov_b64 exec, %[k0]\n\tds_write_b16 %[a0], %[e0] offset:%[qb]\n\t"
                "s_mov_b64 exec, %[m1]\n\ts_waitcnt lgkmcnt(3)\n\t"
                "v_cmp_gt_u32_e64 %[k1], %[cap], %[a1]\n\tv_lshl_add_u32 %[a1], %[c1], 6, %[a1]\n\t"
                "s_mov_b64 exec, %[k1]\n\tds_write_b16 %[a1], %[e1] offset:%[qb]\n\t"
                "s_mov_b64 exec, %[m2]\n\ts_waitcnt lgkmcnt(3)\n\t"
                "v_cmp_gt_u32_e64 %[k2], %[cap], %[a2]\n\tv_lshl_add_u32 %[a2], %[c2], 6, %[a2]\n\t"
                "s_mov_b64 exec, %[k2]\n\tds_write_b16 %[a2], %[e2] offset:%[qb]\n\t"
                "s_mov_b64 exec, %[m3]\n\ts_waitcnt lgkmcnt(3)\n\t"
                "v_cmp_gt_u32_e64 %[k3], %[cap], %[a3]\n\tv_lshl_add_u32 %[a3], %[c3], 6, %[a3]\n\t"
                "s_mov_b64 exec, %[k3]\n\tds_write_b16 %[a3], %[e3] offset:%[qb]\n\t"
                "s_mov_b64 exec, %[ex]\n\t"
                "s_andn2_b64 %[m0], %[m0], %[k0]\n\t"
                "s_andn2_b64 %[m1], %[m1], %[k1]\n\t"
                "s_andn2_b64 %[m2], %[m2], %[k2]\n\t"
                "s_andn2_b64 %[m3], %[m3], %[k3]"
                : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [m0] "=&s"(m0), [m1] "=&s"(m1),
                  [m2] "=&s"(m2), [m3] "=&s"(m3), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
                : [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [c0] "v"(c[0]), [c1] "v"(c[1]),
                  [c2] "v"(c[2]), [c3] "v"(c[3]), [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]),
                  [two] "v"(two), [cap] "s"(cap), [qb] "i"(kLdsQueues), [ex] "s"(exec_in)
                : "memory");
            // m_j now holds the lanes whose pair j found its queue full
            if ((m0 | m1 | m2 | m3) == 0ull) return 0u;  // wave-uniform; the rule
            const unsigned long long me = 1ull << lane;
            return ((m0 & me) ? 1u : 0u) | ((m1 & me) ? 2u : 0u) | ((m2 & me) ? 4u : 0u) | ((m3 & me) ? 8u : 0u);
        };
        // one countable window (raw field in the low 2K bits of x), straight to the histogram
        auto count_direct = [&](uint32_t x) __attribute__((always_inline)) {
            atomicAdd(&hist_s[pair_reverse(x & FMASK, K)], 1u);
        };
        // Singles wait in two registers per lane until the start of the next piece (piece_start): a
        // global atomic issued in the window stage would still be in flight at the wait for the next
        // piece's bytes, and every wave would sit out its round trip to the memory-side atomic unit.
        uint32_t pend0 = 0, pend1 = 0, npend = 0;
        auto single = [&](uint32_t x) __attribute__((always_inline)) {
            const uint32_t code = pair_reverse(x & FMASK, K);
            if (npend == 0u) pend0 = code;
            else if (npend == 1u) pend1 = code;
            else atomicAdd(&hist_s[code], 1u);  // a third single of this lane in one piece (reads full of N)
            ++npend;
        };
        // ---- quad route: the workgroup's drain points ----
        // LDS operations of this wave complete (lgkmcnt) before it arrives; vector memory is not waited for.
        auto wg_sync = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // One drain point: every append issued before it is in the queues (first barrier), every wave moves the full
        // blocks of its sixteen queues to the arena, and nobody appends again before the queues are set up (second).
        auto mid_sync = [&]() __attribute__((always_inline)) {
            wg_sync();
#ifndef VK_QUAD_MID_DRAIN_AT
#define VK_QUAD_MID_DRAIN_AT (2 * kBlockBytes)   // inside a step only queues that are filling up fast are drained (skewed bases)
#endif
            maybe_drain(VK_QUAD_MID_DRAIN_AT);
            wg_sync();
        };
        // The first drain point of a step (= one piece of every wave that still has pieces): the waves also agree on the
        // number of drain points of the step.  While the queues stay calm that is ONE: a step appends ~27 entries to a
        // queue of 128 that starts it with at most 31 -- every further drain point costs the workgroup two barriers with a
        // chain of LDS round trips between them (4 points in 10 wave cycles measured).  A wave that finds a queue three
        // quarters full (skewed bases) raises `tight` for good: from the next step on, a drain point before every round
        // of appends, as many as the busiest wave has.  (slots[step & 1]: the largest (rounds | tight << 16) any wave published
        // -- an LDS atomicMax before the step's first barrier, read between its barriers: the same in every wave; a wave
        // publishes the `tight` it knows, so a new one reaches everybody one step later.)
        // A step in which NO wave appends (stretches of homopolymer reads, which the shortcut in `win` counts) makes the
        // next kIdleSteps steps go without a drain point: the queues hold less than a block each then and take a few
        // steps' appends; more than that overflows into the exact direct count.  (An even number: the slots alternate.)
        constexpr uint32_t kIdleSteps = 6;
        uint32_t qstep = 0, idle_left = 0, tight_known = 0;
        auto step_sync = [&](uint32_t rounds) __attribute__((always_inline)) -> uint32_t {
#ifndef VK_DIAG_QUAD_NO_IDLE
            if (idle_left != 0u) {   // (wave-uniform and the same in every wave: set from the slot word below)
                --idle_left;
                ++qstep;
                return 1u;
            }
#endif
            uint32_t* const slots = reinterpret_cast<uint32_t*>(ldsb + kLdsSync);
            const uint32_t mine = rounds | (tight_known << 16);
            if (lane == 0 && mine != 0u) atomicMax(&slots[qstep & 1u], mine);   // (the rounds field stays 0 when no wave appends in this step)
            wg_sync();
            wave_lds_fence();
            uint32_t n = qcnt[q];
            const uint32_t word = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(slots[qstep & 1u])));
            const bool tight = (word >> 16) != 0u;
            const uint32_t most = word & 0xFFFFu;
            if (tight || __any(n >= 3u * kBlockBytes)) tight_known = 1u;
            if (n > kCap) n = kCap;  // appends beyond the capacity were counted directly
            if (__any(n >= kBlockBytes)) drain_all(n, n / kBlockBytes);
            if (tid == 0) slots[(qstep + 1u) & 1u] = 0u;   // (the next step's slot: last read before this step's first barrier)
            wg_sync();
            ++qstep;
#ifndef VK_DIAG_QUAD_NO_IDLE
            if (most == 0u) idle_left = kIdleSteps;
#endif
            return tight && most > 1u ? most : 1u;
        };
        // Quads of which only some windows count (~1.5 per read) go to HBM as they are found, into the workgroup's region
        // of their bucket (pass B's job of the bucket reads the regions of the sample's workgroups): the place comes from a
        // returning LDS atomic on the workgroup's counter of the bucket; a full region: exact, slow.  (The words of a
        // region are written a few at a time, pieces apart, and meet in L2: 256 open lines per workgroup.)
        uint32_t* const cntp = reinterpret_cast<uint32_t*>(ldsb + kLdsCntP);
        uint32_t* const pregion = QUAD ? bp.preg + ((static_cast<uint64_t>(unit) * kQuadBuckets) << bp.preg_shift) : nullptr;
        // (in two steps, so that the atomic's round trip to LDS passes under the append block; preg_cap = 1 << preg_shift)
        auto partial_reserve = [&](bool have, uint32_t entry) __attribute__((always_inline)) -> uint32_t {   // entry: K + 3 bases | OK bits << 24
            uint32_t a = 0xFFFFFFFFu;
            if (have) a = atomicAdd(&cntp[(entry >> (2 * K - 8)) & 0xFFu], 1u);
            return a;
        };
        auto partial_store = [&](bool have, uint32_t entry, uint32_t a) __attribute__((always_inline)) {
            const uint32_t cap = 1u << bp.preg_shift;
            if (a < cap) {
                const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(pregion, 0, static_cast<int>((kQuadBuckets * 4u) << bp.preg_shift), 0x00020000);
                const uint32_t bq = (entry >> (2 * K - 8)) & 0xFFu;
                __builtin_amdgcn_raw_buffer_store_b32(entry, prsrc, ((bq << bp.preg_shift) | a) << 2, 0, 0);
            }
            if (__any(have && a >= cap)) {
                if (have && a >= cap) count_quad_direct<K>(hist_s, entry & 0xFFFFFFu, entry >> 24);
            }
        };
#ifdef VK_DIAG_K9_BARRIER   // timing only: what two workgroup barriers per piece cost
        uint32_t diag_piece = 0;
#endif
        auto piece_start = [&](const uint4&) __attribute__((always_inline)) {
            if constexpr (QUAD) return;   // (the queues are drained at the step's first drain point, in `win`)
#ifdef VK_DIAG_K9_BARRIER
            if (diag_piece < diag_nbar) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
            }
            ++diag_piece;
#endif
            if (npend > 0u) atomicAdd(&hist_s[pend0], 1u);
            if (npend > 1u) atomicAdd(&hist_s[pend1], 1u);
            npend = 0u;
            maybe_drain(kBlockBytes);  // the queues as the previous piece left them: every full block goes out now
        };
        // Low-complexity input sends every pair to ONE queue, which overflows into per-pair global atomics.
        // A piece that saw an overflow makes the next ones try the homopolymer shortcut first: one global
        // atomic per wavefront and piece (the lanes' window counts summed) instead of thousands.
        // The append stage of one lane group (16 positions): lo = codes of the 16 positions before it, hi = its own,
        // okg = its OK string.
        bool hot = false;
        uint32_t drain_at = kDrainAtCalm;   // (wave-uniform; lowered for good by the first overflow)
        uint32_t xpend = 0, xctx = 0;  // dense stage: groups waiting in the exchange buffer, codes of the last group appended
        auto append_group = [&](uint32_t lo, uint32_t hi, uint32_t okg) __attribute__((always_inline)) {
            // bit 4j of `both` / `one`: both / exactly one of the windows ending at 2j, 2j + 1 count
            const uint32_t both = okg & (okg >> 2) & 0x11111111u;
            const uint32_t one = (okg ^ (okg >> 2)) & 0x11111111u;
#ifdef VK_DIAG_K9_HALF_APPEND   // timing only: half of the pairs are dropped
            constexpr int kHalves = 1;
#else
            constexpr int kHalves = 2;
#endif
#pragma unroll
            for (int h = 0; h < kHalves; ++h) {
                uint32_t x[4], f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = 8 * h + 2 * j;             // windows ending at p and p + 1
                    const int o = 32 + 2 * (p - K + 1);      // bit offset of base p-K+1 in [lo | hi]
                    x[j] = o >= 32 ? (hi >> (o - 32)) : vkl::alignbit(hi, lo, static_cast<uint32_t>(o));
                    f[j] = both & (1u << (16 * h + 4 * j));
                }
                const uint32_t full = append4(x, f);
                if (__any(full != 0u)) {
                    hot = true;
                    drain_at = kDrainAt;
                }
                if (full) {  // rare: the queue was full, count the pair directly
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (full & (1u << j)) {
                            count_direct(x[j]);
                            count_direct(x[j] >> 2);
                        }
                }
            }
            // singles: at most a few per lane and piece
            // (timing diagnostics of pass A, results wrong: -DVK_DIAG_K9_NO_SINGLES, _NO_APPEND, _NO_DRAIN; tools/k9_ab.py)
#ifdef VK_DIAG_K9_NO_SINGLES
            uint32_t rem = 0u & one;
#else
            uint32_t rem = one;
#endif
            while (__any(rem != 0u)) {
                if (rem != 0u) {
                    const uint32_t b = vkl::ffbl(rem);                 // 4j
                    rem &= rem - 1u;
                    const uint32_t second = ((okg >> b) & 1u) ^ 1u;    // 0: the window ending at p counts, 1: at p + 1
                    const uint32_t o = 32u + b + 2u * second - 2u * (K - 1);   // bit offset of the window's first base in [lo | hi]
                    single(o < 32u ? vkl::alignbit(hi, lo, o) : (hi >> (o - 32u)));
                }
            }
        };
        // The quad route's append stage of one lane group: its four quads (windows ending at 4j .. 4j + 3).
        uint32_t hotq_rest = 0;   // rounds of a low-complexity stretch that go without count_hot_quads (wave-uniform)
        auto append_quads = [&](uint32_t lo, uint32_t hi, uint32_t okg) __attribute__((always_inline)) {
#ifndef VK_DIAG_HOT_NO_CLASSES
            if (hot) {   // (wave-uniform, rare)
                // a call that takes out fewer than half of the groups it is shown (repeats with ordinary reads among
                // them: every group of those comes by here) buys the next 32 rounds a rest
                if (hotq_rest != 0u) --hotq_rest;
                else {
                    const uint32_t was = okg;
                    okg = count_hot_quads<K>(hist_s, lo, hi, okg);
                    const uint32_t shown = static_cast<uint32_t>(__builtin_popcountll(__ballot(was != 0u)));
                    const uint32_t taken = static_cast<uint32_t>(__builtin_popcountll(__ballot(okg != was)));
                    if (2u * taken < shown) hotq_rest = 32u;
                }
            }
#endif
            // bit 8j of `all` / `some`: all four / some but not all of the windows ending at 4j .. 4j + 3 count
            const uint32_t p2 = okg & (okg >> 2), o2 = okg | (okg >> 2);
            const uint32_t all = p2 & (p2 >> 4) & 0x01010101u;
            const uint32_t some = ((o2 | (o2 >> 4)) & 0x01010101u) ^ all;
#ifdef VK_DIAG_QUAD_NO_PARTIAL   // timing only
            uint32_t rem = 0u & some;
#else
            uint32_t rem = some;
#endif
            // the lane's first quad of which only some windows count, in straight-line code (most calls find one in some
            // lane, few lanes have two): its place is asked for here, it is stored behind the append block
            auto entry_at = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t {   // b = 8j
                const uint32_t o = (32u - 2u * (K - 1) + b) & 63u;   // bit offset of the quad's first base in [lo | hi]
                const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
                const uint32_t xq = static_cast<uint32_t>(v >> o) & ((1u << (2 * K + 6)) - 1u);
                return xq | (((okg >> (b & 31u)) & 0x55u) << 24);
            };
            const bool have0 = rem != 0u;
            const uint32_t entry0 = entry_at(vkl::ffbl(rem) & 24u);
            const uint32_t place0 = partial_reserve(have0, entry0);
            rem &= rem - 1u;
            uint32_t x[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 32 + 2 * (4 * j - K + 1);      // bit offset of the quad's first base in [lo | hi]
                x[j] = o >= 32 ? (hi >> (o - 32)) : vkl::alignbit(hi, lo, static_cast<uint32_t>(o));
                f[j] = all & (1u << (8 * j));
            }
            const uint32_t full = append4(x, f);
            if (__any(full != 0u)) hot = true;
#ifdef VK_DIAG_QUAD_NO_AGG
            if (full) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (full & (1u << j)) count_quad_direct<K>(hist_s, x[j], 0x55u);
            }
#else
            if (__any(full != 0u)) {  // rare: a queue was full, count those quads directly
#pragma unroll
                for (int j = 0; j < 4; ++j) count_quads_aggregated<K>(hist_s, (full & (1u << j)) != 0u, x[j]);
            }
#endif
            partial_store(have0, entry0, place0);
            while (__any(rem != 0u)) {   // reads cut up by N, read ends that meet in one group
                const bool have = rem != 0u;
                const uint32_t entry = entry_at(vkl::ffbl(rem) & 24u);
                partial_store(have, entry, partial_reserve(have, entry));
                rem &= rem - 1u;
            }
        };
        auto win = [&](uint32_t ch, const uint32_t* C, const uint32_t* ok_) __attribute__((always_inline)) {
            const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
            uint32_t okl[4] = {ok_[0], ok_[1], ok_[2], ok_[3]};
            if (hot) {
                uint32_t n, b;
                if (piece_is_homopolymer<K>(ch, C, ok_, n, b)) {
#pragma unroll
                    for (uint32_t bb = 0; bb < 4; ++bb) {  // (a piece can hold poly-A and poly-T reads: one sum per base)
                        const uint32_t tot = lane_bcast(wave_inclusive_sum(b == bb ? n : 0u), 63);
                        if (tot != 0u && lane == 0) atomicAdd(&hist_s[pair_reverse(bb * (FMASK / 3u), K)], tot);
                    }
                    okl[0] = 0u; okl[1] = 0u; okl[2] = 0u; okl[3] = 0u;   // nothing left to append (the dense stage still passes the context on)
                } else {
                    // tandem repeats of a short period: the lane groups that repeat are counted with a handful of global
                    // atomics per wave (count_repeats), the rest goes through the queues as usual
                    const uint32_t handled = count_repeats<K>(v, okl, lane, [&](uint32_t f, uint32_t cnt) {
                        atomicAdd(&hist_s[pair_reverse(f, K)], cnt);
                    });
                    hot = repeats_dominate(handled, ok_);
                }
            }
            if constexpr (SUB || STREAM) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (__any(okl[g] != 0u)) {  // wave-uniform: a group of sixteen positions without any window is common (MODE 1)
                        append_group(v[g], v[g + 1], okl[g]);
                        if (g == 1) maybe_drain(drain_at);  // inside a piece only a queue that is filling up fast is drained (skewed bases)
                    }
                }
            } else {
                // The append stage runs on lane groups that HAVE windows (45 % of them in a FASTQ of 150-base reads):
                // (codes, OK) of those go through a 64-slot exchange buffer in file order, one group per lane and
                // round.  A group's K - 1 bases of context are the codes of the group before it in the stream:
                // a group without windows of its own is sent along when the next one's windows reach back into it
                // (and the piece's very last group always: the next piece cannot be asked yet).
                constexpr uint32_t kBack = (1u << (2 * (K - 1))) - 1u;  // OK bits of the windows that need bases of the group before
                const uint32_t nb0 = (okl[0] & kBack) != 0u, nb1 = (okl[1] & kBack) != 0u, nb2 = (okl[2] & kBack) != 0u,
                               nb3 = (okl[3] & kBack) != 0u;
                const uint32_t nbn = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(1, static_cast<int>(nb0), 0x130, 0xF, 0xF, false));  // wave_shl:1: lane + 1's, lane 63 gets 1
                const uint32_t lv[4] = {(okl[0] != 0u) | nb1, (okl[1] != 0u) | nb2, (okl[2] != 0u) | nb3, (okl[3] != 0u) | nbn};
                const uint32_t n = lv[0] + lv[1] + lv[2] + lv[3];
                const uint32_t incl = wave_inclusive_sum(n);
                const uint32_t tot = xpend + lane_bcast(incl, 63);
                uint32_t at = xpend + incl - n;
                uint32_t wp[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    wp[g] = lv[g] ? at : 0xFFFFFFC0u;
                    at += lv[g];
                }
                uint2* const xg = reinterpret_cast<uint2*>(ldsb + kLdsXchg) + static_cast<uint32_t>(wave) * 64u;
                const uint32_t rounds = tot >> 6;
                // quad route: the step's first drain point; `synced` of its `m` drain points are behind this wave
                uint32_t m = 0, synced = 1;
                if constexpr (QUAD) m = step_sync(rounds);
                for (uint32_t r = 0; r <= rounds; ++r) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if ((wp[g] >> 6) == r) xg[wp[g] & 63u] = make_uint2(C[g], okl[g]);
                    if (r == rounds) break;  // what is left (< 64 groups) waits in the buffer for the next piece
                    uint32_t lz = static_cast<uint32_t>(lane);
                    asm volatile("" : "+v"(lz));
                    const uint2 e = xg[lz];
                    const uint32_t lo = wave_prev_lane(e.x, xctx);
                    xctx = lane_bcast(e.x, 63);
                    if constexpr (QUAD) {
#ifdef VK_DIAG_K9_NO_APPEND
                        asm volatile("" :: "v"(lo), "v"(e.x), "v"(e.y));
#else
                        append_quads(lo, e.x, e.y);
#endif
                        if (synced < m) {   // a drain point before every further round of the step
                            mid_sync();
                            ++synced;
                        }
                        continue;
                    }
#ifdef VK_DIAG_K9_NO_APPEND
                    asm volatile("" :: "v"(lo), "v"(e.x), "v"(e.y));
#else
                    append_group(lo, e.x, e.y);
#endif
#ifdef VK_DIAG_K9_NO_DRAIN
                    wave_lds_fence();
                    if (lane < static_cast<int>(kQueues)) qcnt[lane] = 0u;   // timing only: the queues are thrown away
                    wave_lds_fence();
#else
                    maybe_drain(drain_at);
#endif
                }
                if constexpr (QUAD) {
                    while (synced < m) {   // the drain points of rounds other waves have and this one has not
                        mid_sync();
                        ++synced;
                    }
                }
                xpend = tot & 63u;
            }
        };
        SubWave sw = {0, 0, 0, 0};
        if constexpr (SUB) {
            sw.seed = sp.seeds[s];
            sw.threshold = sp.thresholds[s];
        }
        if constexpr (STREAM) {
            // The wave's segment of the packed stream, kPackBlock records at a time, four consecutive records per lane:
            // exactly the (codes of the 16 positions before, four code words, four OK words) a piece of the text front
            // end hands to `win`, with every group holding sequence.  Records beyond the segment's end read as BAD.
            const uint32_t cw = pk.count[unit * kWaves + static_cast<uint32_t>(wave)];
            const uint32_t nrec = cw & ~kPackHasPre;
            const uint64_t seg = pack_segment(pk, s, part, wave, wr.w0);
            const uint4* const pc4 = reinterpret_cast<const uint4*>(pk.c + seg);
            const uint4* const pm4 = reinterpret_cast<const uint4*>(pk.m + seg);
            const uint32_t nblocks = (nrec + kPackBlock - 1u) / kPackBlock;
            uint32_t carry_c = 0u, carry_bad = 0x55555555u;
            uint4 c4 = make_uint4(0u, 0u, 0u, 0u), m4 = c4;
            if (nblocks != 0u) {
                c4 = pc4[lane];
                m4 = pm4[lane];
            }
            for (uint32_t b = 0; b < nblocks; ++b) {
                const uint4 cc = c4, mm = m4;
                piece_start(cc);
                if (b + 1u < nblocks) {   // the next block's loads fly under this one's appends
                    c4 = pc4[(b + 1u) * 64u + static_cast<uint32_t>(lane)];
                    m4 = pm4[(b + 1u) * 64u + static_cast<uint32_t>(lane)];
                }
                const uint32_t at = b * kPackBlock + 4u * static_cast<uint32_t>(lane);
                const uint32_t cv[4] = {cc.x, cc.y, cc.z, cc.w}, mv[4] = {mm.x, mm.y, mm.z, mm.w};
                uint32_t C[4], bad[4], ok[4];
#pragma unroll
                for (uint32_t g = 0; g < 4; ++g) {
                    const bool in = at + g < nrec;
                    C[g] = in ? cv[g] : 0u;
                    bad[g] = in ? (mv[g] & 0x55555555u) : 0x55555555u;
                }
                const uint32_t badh = wave_prev_lane(bad[3], carry_bad);
                const uint32_t ch = wave_prev_lane(C[3], carry_c);
                carry_bad = lane_bcast(bad[3], 63);
                carry_c = lane_bcast(C[3], 63);
                vkl::ok_mask<K>(badh, bad, ok);
                if (b == 0u && lane == 0 && (cw & kPackHasPre) != 0u) ok[0] = 0u;   // the record of context: its windows are the wave before's
                win(ch, C, ok);
            }
        } else {
#ifndef VK_QUAD_PF
#define VK_QUAD_PF 0   // (1, 2: the next piece prefetched into registers -- measured: 32 bytes of scratch per lane, 12.9 against 11.2 ms)
#endif
            if (!QUAD || !wr.empty)
                wave_stream<K, SUB, SUB ? 1 : (QUAD ? VK_QUAD_PF : 0)>(sbase, len, wr.w0, wr.w1, scratch_all + wave * 8, below, above, lane, win,
                                                        piece_start, ph_start, ph_end, sw);
        }
        if constexpr (SUB) flush_sites(sp, s, sw, lane);
        if constexpr (MODE == 0 || QUAD) {
            if (xpend != 0u) {  // the groups still waiting in the exchange buffer
                const uint2* const xg = reinterpret_cast<const uint2*>(ldsb + kLdsXchg) + static_cast<uint32_t>(wave) * 64u;
                uint2 e = make_uint2(0u, 0u);
                if (static_cast<uint32_t>(lane) < xpend) e = xg[lane];
                const uint32_t lo = wave_prev_lane(e.x, xctx);
                if constexpr (QUAD) append_quads(lo, e.x, e.y);
                else append_group(lo, e.x, e.y);
                xpend = 0u;
            }
        }
        if constexpr (QUAD) {
            // the steps of the waves that have more pieces than this one (wave_stream's piece count: one `win` per piece),
            // then one more barrier: the appends behind the last step's drain points are in the queues
            uint32_t nmax = 0;
            for (int w = 0; w < kWaves; ++w) {
                const WaveRange r = wave_range(len, parts, part, w);
                if (!r.empty) {
                    const uint64_t o0 = r.w0 != 0 ? r.w0 - 64 : 0;
                    const uint32_t np = static_cast<uint32_t>((r.w1 - o0 + kPiece - 1) / kPiece);
                    nmax = np > nmax ? np : nmax;
                }
            }
            while (qstep < nmax) {
                const uint32_t m = step_sync(0u);
                for (uint32_t r = 1; r < m; ++r) mid_sync();
            }
            wg_sync();
            if (tid < static_cast<int>(kQuadBuckets)) {
                const uint32_t c = cntp[tid];
                bp.preg_n[static_cast<uint64_t>(unit) * kQuadBuckets + static_cast<uint32_t>(tid)] = c < (1u << bp.preg_shift) ? c : (1u << bp.preg_shift);
            }
        }
        if (npend > 0u) atomicAdd(&hist_s[pend0], 1u);
        if (npend > 1u) atomicAdd(&hist_s[pend1], 1u);
        // the end of the range: full blocks to the arena, the rest of every queue counted directly, runs closed
        wave_lds_fence();
        uint32_t n = qcnt[q];
        if (n > kCap) n = kCap;
        drain_all(n, n / kBlockBytes);
        n = qcnt[q];
        const uint16_t* q16 = reinterpret_cast<const uint16_t*>(qdata);
        const uint32_t bq = QUAD ? static_cast<uint32_t>(wave) * kQueues + q : q;
        for (uint32_t i = sub; i < n / 2u; i += 4u) {
            if constexpr (QUAD) count_quad_direct<K>(hist_s, quad_bases<K>(bq, q16[i]), 0x55u);
            else count_entry_direct<K>(hist_s, q, q16[i]);
        }
        const uint32_t used = qused[q];
        if (sub == 0 && used <= kRunBlocks && used != 0u) {
            hdrs[qrun[q]] = 0x80000000u | (used << 8) | bq;
            if constexpr (!QUAD) atomicAdd(&bp.bsize[s * kQueues + q], used);
        }
    }
    if constexpr (!STREAM) {   // (MODE 2: vk_pack_kernel has written the waves' phase words)
        if (lane == 0) wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));
    }
}

// Between pass A and pass B: the (sample, bucket) jobs of pass B ordered by size class (the power of two of
// their block count), largest first.  One workgroup of pass B fills a CU (128 KB histogram), a launch is six
// rounds of them, and with a skewed base composition some bucket streams are four times the average: in
// (sample, bucket) order the launch ended with one of those running alone (3.9 -> 6.3 ms on GC-rich data).
// A counting sort over 33 classes by one workgroup; order within a class is whatever the atomics make it.
__global__ __launch_bounds__(1024) void vk_bucket_order_kernel(BucketParams bp, uint32_t njobs) {
    __shared__ uint32_t cnt[33], first[33];
    const uint32_t tid = threadIdx.x;
    if (tid < 33) cnt[tid] = 0u;
    __syncthreads();
    auto cls = [](uint32_t blocks) { return blocks ? 32u - static_cast<uint32_t>(__builtin_clz(blocks)) : 0u; };  // 0 .. 32
    // (one LDS atomic per wavefront and class, not per job: the quad route has 256 jobs per sample, nearly all of two or
    // three classes -- 25600 atomics on three counters were the kernel's 45 us)
    auto grouped = [&](uint32_t c, bool have, uint32_t* counters) -> uint32_t {   // the job's place among its class: counters[c] moves on by the class's lanes
        unsigned long long left = __ballot(have);
        const uint32_t ln = tid & 63u;
        uint32_t at = 0;
        while (left != 0ull) {
            const uint32_t lead = static_cast<uint32_t>(__builtin_ctzll(left));
            const uint32_t cc = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(c), static_cast<int>(lead)));
            const unsigned long long eq = __ballot(have && c == cc) & left;
            uint32_t base = 0;
            if (ln == lead) base = atomicAdd(&counters[cc], static_cast<uint32_t>(__builtin_popcountll(eq)));
            base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(base), static_cast<int>(lead)));
            if (have && c == cc) at = base + static_cast<uint32_t>(__builtin_popcountll(eq & ((1ull << ln) - 1ull)));
            left &= ~eq;
        }
        return at;
    };
    const uint32_t njobs_up = (njobs + 1023u) / 1024u * 1024u;   // (whole wavefronts take part in the ballots)
    // (the loops count in a scalar: `j = tid; j < njobs_up; j += 1024` is a loop the lanes may leave one by one as far as
    // hipcc can tell, and the ballots and v_readlane of `grouped` do not belong in one -- tools/asm_lint.py, convergence)
    for (uint32_t j0 = 0; j0 < njobs_up; j0 += 1024) {
        const uint32_t j = j0 + tid;
        grouped(j < njobs ? cls(bp.bsize[j]) : 0u, j < njobs, cnt);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t at = 0;
        for (int c = 32; c >= 0; --c) {
            first[c] = at;
            at += cnt[c];
        }
    }
    __syncthreads();
    for (uint32_t j0 = 0; j0 < njobs_up; j0 += 1024) {
        const uint32_t j = j0 + tid;
        const uint32_t at = grouped(j < njobs ? cls(bp.bsize[j]) : 0u, j < njobs, first);
        if (j < njobs) bp.order[at] = j;
    }
}

// Pass B, the shipped replay (vk_bucket_count_kernel): ONE LDS add per PAIR.  A pair entry e (LB + 2 bits: the K - 2
// bases before the bucket's two, and the base after them) names both of its windows, so the entry itself is counted:
// 4^(K+1) / 16 pair counters per bucket, u16 each, two to a word -- word = the low LB + 1 bits of e, half = its top
// bit -- which is the 128 KiB (K = 9) the two u32 window tables took.  Pass C sums, per k-mer code, the four pair
// counters that hold it as their first window and the four that hold it as their second.  Half the LDS atomics of
// counting windows (the replay is bound by their bank conflicts).
// A u16 counter wraps after 65535 equal pairs of one bucket stream (poly-A tails of a whole sample do that): the
// wrap carries into the neighbouring counter or out of the word, either way the sum over all counters no longer
// equals the number of entries replayed (each wrap loses 65535 or 65536, and there are fewer than 2^30 entries), so
// the job compares the two, and a job that does not add up is flagged (bp.wide) and replayed by
// vk_bucket_count_wide_kernel into u32 WINDOW counters (the replay of rounds 1-3, two adds per pair); pass C reads
// whichever table the flag names.  Exact either way.
template <int K>
__global__ __launch_bounds__(kCountThreads) void vk_bucket_count_kernel(BucketParams bp) {
    constexpr uint32_t LB = 2 * K - 4;
    constexpr uint32_t WORDS = 2u << LB;  // u32 words = pair counters / 2
    constexpr uint32_t kList = 4096;      // runs listed per round (a uniform sample has ~2400 per bucket)
    __shared__ uint32_t hist[WORDS];
    __shared__ uint32_t list[kList];
    __shared__ uint32_t nlist, sum_all, sum_tallied;
    const uint32_t job = bp.order[blockIdx.x];  // large streams first (vk_bucket_order_kernel)
    const uint32_t s = job / kQueues, q = job % kQueues;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < WORDS; i += kCountThreads) hist[i] = 0u;
    if (tid == 0) { sum_all = 0u; sum_tallied = 0u; }
    uint32_t nruns = bp.cursors[s];
    if (nruns > bp.runs_cap) nruns = bp.runs_cap;
    const uint32_t* hdrs = bp.hdrs + static_cast<uint64_t>(s) * bp.runs_cap;
    const uint8_t* arena = bp.arena + static_cast<uint64_t>(s) * bp.runs_cap * kRunBytes;
    // entry e (u16; garbage above bit LB + 1 for K = 8): byte address of its word (e << 2) & M, addend 1 or 0x10000 by bit LB + 1
    constexpr uint32_t M = ((1u << (LB + 1)) - 1u) << 2;
    uint8_t* const h0 = reinterpret_cast<uint8_t*>(hist);
    auto tally2 = [&](uint32_t w) __attribute__((always_inline)) {  // two entries: the halves of a dword
#ifdef VK_DIAG_QB_NOATOMIC       // timing only
        asm volatile("" :: "v"(w));
        return;
#endif
#ifdef VK_DIAG_QB_NOCONFLICT     // timing only: every lane its own bank
        w = (w & ~0x03FF03FFu) | (((tid & 63u) | ((tid & 15u) << 6)) * 0x00010001u);   // bits 0..5 and bits 4..9 both name the lane
#endif
        const uint32_t t0 = (w >> (LB + 1)) & 1u, t1 = (w >> (LB + 17)) & 1u;
        atomicAdd(reinterpret_cast<uint32_t*>(h0 + ((w << 2) & M)), __umul24(t0, 0xFFFFu) + 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(h0 + ((w >> 14) & M)), __umul24(t1, 0xFFFFu) + 1u);
    };
    const uint32_t g = tid & 255u;   // 256 threads per run: the 16-byte granule this thread reads (four per block)
    const uint32_t grp = tid >> 8;   // four runs at a time
    auto fetch = [&](uint32_t i, uint32_t n, uint4& v) __attribute__((always_inline)) -> bool {
        const uint32_t item = i < n ? list[i] : 0u;   // (behind the list's end: run 0, nothing taken)
        const bool have = i < n && (g >> 2) < (item >> 24);
        // (unconditional, a thread without a granule re-reads the run's first: in a branch every load waited out the one before -- vk_quad_count_kernel)
        v = *(reinterpret_cast<const uint4*>(arena + static_cast<uint64_t>(item & 0xFFFFFFu) * kRunBytes) + (have ? g : 0u));
        return have;
    };
    uint32_t tallied = 0;            // entries this thread replayed
    auto tally4 = [&](const uint4& v) __attribute__((always_inline)) {
        tally2(v.x);
        tally2(v.y);
        tally2(v.z);
        tally2(v.w);
        tallied += 8u;
    };
    uint32_t r0 = 0;
    while (r0 < nruns) {
        if (tid == 0) nlist = 0u;
        __syncthreads();
        uint32_t listed = 0;
        for (; r0 < nruns; r0 += kCountThreads) {
            if (listed + kCountThreads > kList) break;
            const uint32_t r = r0 + tid;
            const uint32_t h = r < nruns ? hdrs[r] : 0u;
            const bool mine = (h >> 31) != 0u && (h & 0xFFu) == q;
            const unsigned long long bal = __ballot(mine);
            uint32_t base = 0;
            if ((tid & 63u) == 0u && bal != 0ull) base = atomicAdd(&nlist, static_cast<uint32_t>(__popcll(bal)));
            base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
            if (mine) list[base + __popcll(bal & ((1ull << (tid & 63u)) - 1ull))] = r | ((h >> 8) & 0xFFu) << 24;
            __syncthreads();
            listed = nlist;
            __syncthreads();
        }
        const uint32_t n = listed;
        uint4 a0, a1, a2, a3, b0, b1, b2, b3;
        bool ha0, ha1, ha2, ha3, hb0, hb1, hb2, hb3;
        uint32_t i = grp;
        ha0 = fetch(i, n, a0); ha1 = fetch(i + 4, n, a1); ha2 = fetch(i + 8, n, a2); ha3 = fetch(i + 12, n, a3);
        while (i < n) {
            hb0 = fetch(i + 16, n, b0); hb1 = fetch(i + 20, n, b1); hb2 = fetch(i + 24, n, b2); hb3 = fetch(i + 28, n, b3);
            if (ha0) tally4(a0);
            if (ha1) tally4(a1);
            if (ha2) tally4(a2);
            if (ha3) tally4(a3);
            i += 16;
            if (i >= n) break;
            ha0 = fetch(i + 16, n, a0); ha1 = fetch(i + 20, n, a1); ha2 = fetch(i + 24, n, a2); ha3 = fetch(i + 28, n, a3);
            if (hb0) tally4(b0);
            if (hb1) tally4(b1);
            if (hb2) tally4(b2);
            if (hb3) tally4(b3);
            i += 16;
        }
        __syncthreads();
    }
    __syncthreads();
    uint32_t* out = bp.bucket_hist + (static_cast<uint64_t>(s) * kQueues + q) * WORDS;
    uint32_t total = 0;
    for (uint32_t i = tid; i < WORDS; i += kCountThreads) {
        const uint32_t w = hist[i];
        out[i] = w;
        total += (w & 0xFFFFu) + (w >> 16);
    }
    // do the counters add up to the entries replayed?  (wave sums first: two LDS atomics per wavefront)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        total += __shfl_xor(total, d);
        tallied += __shfl_xor(tallied, d);
    }
    if ((tid & 63u) == 0u) {
        atomicAdd(&sum_all, total);
        atomicAdd(&sum_tallied, tallied);
    }
    __syncthreads();
    if (tid == 0) bp.wide[job] = (sum_all != sum_tallied || bp.force_wide != 0u) ? 1u : 0u;
}

// The u32 replay of a job whose pair counters wrapped (see above; rounds 1-3 ran every job through it):
// one workgroup per (sample, bucket) replays the runs of its bucket into a 2 x 4^K/16-bin
// LDS histogram and stores it, as it stands, to bucket_hist (plain coalesced stores).
// Pass C (vk_bucket_merge_kernel): one thread per k-mer code adds the two counters that can name it
// -- the code seen as the first and as the second window of a pair -- to the histogram, which already
// holds pass A's direct counts.  (Adding from pass B with atomics in code order scattered every
// lane to its own cache line: the bucket number is the END of the first window, the least
// significant digits of its code; that flush alone took as long as the replay.)
template <int K>
__global__ __launch_bounds__(kCountThreads) void vk_bucket_count_wide_kernel(BucketParams bp) {
    constexpr uint32_t LB = 2 * K - 4;
    constexpr uint32_t BINS = 2u << LB;  // type bit | index
    constexpr uint32_t kList = 4096;     // runs listed per round (a uniform sample has ~2400 per bucket)
    __shared__ uint32_t hist[BINS];
    __shared__ uint32_t list[kList];
    __shared__ uint32_t nlist;
    const uint32_t job = bp.order[blockIdx.x];  // large streams first (vk_bucket_order_kernel)
    if (bp.wide[job] == 0u) return;             // the u16 pair counters of vk_bucket_count_kernel held
    const uint32_t s = job / kQueues, q = job % kQueues;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < BINS; i += kCountThreads) hist[i] = 0u;
    uint32_t nruns = bp.cursors[s];
    if (nruns > bp.runs_cap) nruns = bp.runs_cap;
    const uint32_t* hdrs = bp.hdrs + static_cast<uint64_t>(s) * bp.runs_cap;
    const uint8_t* arena = bp.arena + static_cast<uint64_t>(s) * bp.runs_cap * kRunBytes;
    // Entry e = rest | last << LB (LB + 2 bits, garbage above for K = 8).  Type 0 counts under rest, type 1
    // under (rest >> 2) | last << (LB - 2) = e >> 2: as BYTE offsets into the two halves of hist,
    // (e << 2) & M and e & M with M = the LB + 2 bit mask without its two lowest bits.
    constexpr uint32_t M = (1u << (LB + 2)) - 4u;
    uint8_t* const h0 = reinterpret_cast<uint8_t*>(hist);
    uint8_t* const h1 = reinterpret_cast<uint8_t*>(hist + (1u << LB));
    auto tally2 = [&](uint32_t w) __attribute__((always_inline)) {  // two entries: the halves of a dword
        atomicAdd(reinterpret_cast<uint32_t*>(h0 + ((w << 2) & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(h1 + (w & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(h0 + ((w >> 14) & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(h1 + ((w >> 16) & M)), 1u);
    };
    const uint32_t g = tid & 255u;   // 256 threads per run: the 16-byte granule this thread reads (four per block)
    const uint32_t grp = tid >> 8;   // four runs at a time
    auto fetch = [&](uint32_t i, uint32_t n, uint4& v) __attribute__((always_inline)) -> bool {
        const uint32_t item = i < n ? list[i] : 0u;   // (behind the list's end: run 0, nothing taken)
        const bool have = i < n && (g >> 2) < (item >> 24);
        // (unconditional, a thread without a granule re-reads the run's first: in a branch every load waited out the one before -- vk_quad_count_kernel)
        v = *(reinterpret_cast<const uint4*>(arena + static_cast<uint64_t>(item & 0xFFFFFFu) * kRunBytes) + (have ? g : 0u));
        return have;
    };
    auto tally4 = [&](const uint4& v) __attribute__((always_inline)) {
        tally2(v.x);
        tally2(v.y);
        tally2(v.z);
        tally2(v.w);
    };
    uint32_t r0 = 0;
    while (r0 < nruns) {
        // round: list this bucket's runs among the headers from r0 on, until the list is full
        if (tid == 0) nlist = 0u;
        __syncthreads();
        uint32_t listed = 0;  // nlist as of the last barrier: the same in every thread, so the exit is uniform
        for (; r0 < nruns; r0 += kCountThreads) {
            if (listed + kCountThreads > kList) break;
            const uint32_t r = r0 + tid;
            const uint32_t h = r < nruns ? hdrs[r] : 0u;
            // compacted per wave: one LDS atomic per wavefront, not one per run
            const bool mine = (h >> 31) != 0u && (h & 0xFFu) == q;
            const unsigned long long bal = __ballot(mine);
            uint32_t base = 0;
            if ((tid & 63u) == 0u && bal != 0ull) base = atomicAdd(&nlist, static_cast<uint32_t>(__popcll(bal)));
            base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
            if (mine) list[base + __popcll(bal & ((1ull << (tid & 63u)) - 1ull))] = r | ((h >> 8) & 0xFFu) << 24;
            __syncthreads();
            listed = nlist;
            __syncthreads();  // nobody adds to nlist before everybody has read it
        }
        const uint32_t n = listed;
        // the stream is read once: keep eight 16-byte loads in flight per thread (two batches of four
        // runs, the next batch's loads issued before the current one is tallied)
        uint4 a0, a1, a2, a3, b0, b1, b2, b3;
        bool ha0, ha1, ha2, ha3, hb0, hb1, hb2, hb3;
        uint32_t i = grp;
        ha0 = fetch(i, n, a0); ha1 = fetch(i + 4, n, a1); ha2 = fetch(i + 8, n, a2); ha3 = fetch(i + 12, n, a3);
        while (i < n) {
            hb0 = fetch(i + 16, n, b0); hb1 = fetch(i + 20, n, b1); hb2 = fetch(i + 24, n, b2); hb3 = fetch(i + 28, n, b3);
            if (ha0) tally4(a0);
            if (ha1) tally4(a1);
            if (ha2) tally4(a2);
            if (ha3) tally4(a3);
            i += 16;
            if (i >= n) break;
            ha0 = fetch(i + 16, n, a0); ha1 = fetch(i + 20, n, a1); ha2 = fetch(i + 24, n, a2); ha3 = fetch(i + 28, n, a3);
            if (hb0) tally4(b0);
            if (hb1) tally4(b1);
            if (hb2) tally4(b2);
            if (hb3) tally4(b3);
            i += 16;
        }
        __syncthreads();
    }
    __syncthreads();
    uint32_t* out = bp.bucket_hist + (static_cast<uint64_t>(s) * kQueues + q) * BINS;
    for (uint32_t i = tid; i < BINS; i += kCountThreads) out[i] = hist[i];
}

template <int K>
__global__ __launch_bounds__(256) void vk_bucket_merge_kernel(BucketParams bp, uint32_t* __restrict__ hist_out) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr uint32_t LB = 2 * K - 4;
    constexpr uint32_t BINS = 2u << LB;
    const uint32_t s = blockIdx.x / (NCODE / 256), code = (blockIdx.x % (NCODE / 256)) * 256 + threadIdx.x;
    const uint32_t raw = pair_reverse(code, K);  // first base least significant, as the pair entries are built
    // as the FIRST window of a pair: bucket = its last two bases, the entry's low LB bits = its first K - 2 bases
    // (the base after the pair's bucket bases is any of four)
    const uint32_t q0 = raw >> LB, i0 = raw & ((1u << LB) - 1u);
    // as the SECOND window: bucket = bases K-3, K-2; the entry's low LB bits = (any base | its first K - 3 bases),
    // the entry's top two bits its last base
    const uint32_t q1 = (raw >> (LB - 2)) & 15u;
    const uint32_t head = raw & ((1u << (LB - 2)) - 1u), lastb = raw >> (2 * K - 2);
    const uint32_t* bh = bp.bucket_hist + static_cast<uint64_t>(s) * kQueues * BINS;
    const uint32_t* wide = bp.wide + s * kQueues;
    uint32_t add;
    auto halves = [](uint32_t w) { return (w & 0xFFFFu) + (w >> 16); };
    if (wide[q0] != 0u) {
        add = bh[q0 * BINS + i0];                                   // u32 window counters: [type][index]
    } else {
        // pair counters: word = low LB + 1 bits of the entry, half = its top bit; last base 0..3 = two words, both halves
        add = halves(bh[q0 * BINS + i0]) + halves(bh[q0 * BINS + i0 + (1u << LB)]);
    }
    if (wide[q1] != 0u) {
        add += bh[q1 * BINS + (1u << LB) + (head | (lastb << (LB - 2)))];
    } else {
        // the four entries (any first base) are neighbours: one 16-byte load, the half by the last base's upper bit
        const uint4 v = *reinterpret_cast<const uint4*>(bh + q1 * BINS + ((head << 2) | ((lastb & 1u) << LB)));
        const uint32_t sh = (lastb >> 1) * 16u;
        add += ((v.x >> sh) & 0xFFFFu) + ((v.y >> sh) & 0xFFFFu) + ((v.z >> sh) & 0xFFFFu) + ((v.w >> sh) & 0xFFFFu);
    }
    uint32_t* out = hist_out + static_cast<uint64_t>(s) * NCODE;
    out[code] += add;
}

// ---- the quad route behind pass A ------------------------------------------------------------------------
// vk_quad_list_kernel: one workgroup per sample sorts the sample's closed runs by bucket (a counting sort over
// the run headers): qfirst[s][b] .. qfirst[s][b + 1] are bucket b's entries of qlist[s], each run | filled blocks << 24.
__global__ __launch_bounds__(1024) void vk_quad_list_kernel(BucketParams bp) {
    __shared__ uint32_t cnt[kQuadBuckets], pos[kQuadBuckets];
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    if (tid < kQuadBuckets) cnt[tid] = 0u;
    __syncthreads();
    uint32_t nruns = bp.cursors[s];
    if (nruns > bp.runs_cap) nruns = bp.runs_cap;
    const uint32_t* hdrs = bp.hdrs + static_cast<uint64_t>(s) * bp.runs_cap;
    for (uint32_t r = tid; r < nruns; r += 1024) {
        const uint32_t h = hdrs[r];
        if (h >> 31) atomicAdd(&cnt[h & 0xFFu], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // exclusive prefix over the 256 buckets: four per lane of one wavefront
        const uint32_t c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
        const uint32_t sum = c0 + c1 + c2 + c3;
        const uint32_t before = wave_inclusive_sum(sum) - sum;
        pos[4 * tid] = before;
        pos[4 * tid + 1] = before + c0;
        pos[4 * tid + 2] = before + c0 + c1;
        pos[4 * tid + 3] = before + c0 + c1 + c2;
        uint32_t* first = bp.qfirst + static_cast<uint64_t>(s) * (kQuadBuckets + 1);
        first[4 * tid] = before;
        first[4 * tid + 1] = before + c0;
        first[4 * tid + 2] = before + c0 + c1;
        first[4 * tid + 3] = before + c0 + c1 + c2;
        if (tid == 63) first[kQuadBuckets] = before + sum;
    }
    if (tid < kQuadBuckets) bp.bsize[s * kQuadBuckets + tid] = cnt[tid];   // the job's size for vk_bucket_order_kernel: its runs
    __syncthreads();
    uint32_t* list = bp.qlist + static_cast<uint64_t>(s) * bp.runs_cap;
    for (uint32_t r = tid; r < nruns; r += 1024) {
        const uint32_t h = hdrs[r];
        if (h >> 31) list[atomicAdd(&pos[h & 0xFFu], 1u)] = r | (((h >> 8) & 0x7Fu) << 24);
    }
}

// Pass B of the quad route: one 512-thread workgroup per (sample, bucket) replays the bucket's runs into two u32
// tables of 4^(K-3) counters in LDS -- an entry's low 2K - 6 bits name its windows 0 and 1 (head, first tail base: T0),
// its bits [4, 2K - 2) its windows 2 and 3 (head without its first two bases, all three tail bases: T1).  The
// bucket's listed quads follow: a whole pair of one goes into the same tables, a lone window t into R[t], the four
// arrays of 4^(K-4) counters per window type that the job stores in the end --
//     R[0][j] += sum over a of T0[j | a << HB]      R[1][j] += sum over a of T0[j << 2 | a]
//     R[2][j] += sum over a of T1[j | a << HB]      R[3][j] += sum over a of T1[j << 2 | a]          (HB = 2K - 8)
// (j = the window's K - 4 bases outside the bucket, first base lowest) -- to bucket_hist[sample][bucket][t][j'],
// j' = j with its bases in reverse order (the merge reads them by k-mer code, last base lowest).
// u32 counters: nothing can wrap, no second replay.
template <int K>
__global__ __launch_bounds__(512) void vk_quad_count_kernel(BucketParams bp) {
    constexpr uint32_t TB = 1u << (2 * K - 6);            // counters per table
    constexpr uint32_t HB = 2 * K - 8;
    constexpr uint32_t RB = 1u << HB;                     // counters per window array
    constexpr uint32_t M = (TB - 1u) << 2;                // byte offset mask
    __shared__ __attribute__((aligned(16))) uint32_t tab[2 * TB];
    __shared__ uint32_t rr[4 * RB];
    const uint32_t tid = threadIdx.x;
    const uint32_t job = bp.order[blockIdx.x];  // large streams first (vk_bucket_order_kernel): skewed bases make some buckets several times the average
    const uint32_t s = job / kQuadBuckets, q = job % kQuadBuckets;
    const uint32_t* first = bp.qfirst + static_cast<uint64_t>(s) * (kQuadBuckets + 1);
    const uint32_t beg = first[q], end = first[q + 1];
    {   // a bucket without entries and without listed quads (low-complexity samples: most of their 256) stores zeros and leaves
        uint32_t any = end - beg;
        if (any == 0u)   // (only then: a population per part is a chain of loads a job with runs should not wait for)
            for (uint32_t p = 0; p < bp.parts; ++p) any |= bp.preg_n[(static_cast<uint64_t>(s) * bp.parts + p) * kQuadBuckets + q];
        if (any == 0u) {
            uint32_t* out0 = bp.bucket_hist + (static_cast<uint64_t>(s) * kQuadBuckets + q) * (4 * RB);
            for (uint32_t j = tid; j < 4 * RB; j += 512) out0[j] = 0u;
            return;
        }
    }
    for (uint32_t i = tid; i < 2 * TB; i += 512) tab[i] = 0u;
    for (uint32_t i = tid; i < 4 * RB; i += 512) rr[i] = 0u;
    const uint32_t* list = bp.qlist + static_cast<uint64_t>(s) * bp.runs_cap;
    const uint8_t* arena = bp.arena + static_cast<uint64_t>(s) * bp.runs_cap * kRunBytes;
    // The job's run list goes through LDS, kListChunk items at a time: read from memory where it is needed, an item came
    // back behind every load issued before it (one counter, in order) -- each batch of run loads waited out the batch
    // before it, and a memory round trip with nothing but four list words in flight followed.
    constexpr uint32_t kListChunk = 1024;
    __shared__ uint32_t litems[kListChunk];
    for (uint32_t i = tid; i < kListChunk && beg + i < end; i += 512) litems[i] = list[beg + i];
    __syncthreads();
    uint8_t* const t0 = reinterpret_cast<uint8_t*>(tab);
    uint8_t* const t1 = reinterpret_cast<uint8_t*>(tab + TB);
    auto tally2 = [&](uint32_t w) __attribute__((always_inline)) {  // two entries: the halves of a dword
        atomicAdd(reinterpret_cast<uint32_t*>(t0 + ((w << 2) & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(t1 + ((w >> 2) & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(t0 + ((w >> 14) & M)), 1u);
        atomicAdd(reinterpret_cast<uint32_t*>(t1 + ((w >> 18) & M)), 1u);
    };
    auto tally4 = [&](const uint4& v) __attribute__((always_inline)) {
        tally2(v.x);
        tally2(v.y);
        tally2(v.z);
        tally2(v.w);
    };
    // a run = 256 granules of 16 bytes: half of the workgroup per run (g = the thread's granule), two runs at a time,
    // eight loads in flight per thread
    const uint32_t g = tid & 255u, half = tid >> 8;
    for (uint32_t cbeg = beg; cbeg < end; cbeg += kListChunk) {
        const uint32_t cn = end - cbeg < kListChunk ? end - cbeg : kListChunk;   // items of this chunk: litems[0 .. cn)
        if (cbeg != beg) {   // (a job of more than kListChunk runs: skewed or low-complexity samples)
            __syncthreads();
            for (uint32_t i = tid; i < cn; i += 512) litems[i] = list[cbeg + i];
            __syncthreads();
        }
        // (the load is unconditional -- a thread without a granule re-reads the first one of the run, or of run 0 behind the
        // list's end: inside a branch every load came out with an `s_waitcnt vmcnt(0)` in front of it, ONE load in flight per
        // wave, and the kernel ran at the latency of its stream: 1.93 ms whether or not it counted anything)
        auto fetch = [&](uint32_t i, uint4& v) __attribute__((always_inline)) -> bool {
            const uint32_t item = litems[i < cn ? i : 0u];
            const bool in = i < cn;
            const bool have = in && (g >> 2) < (item >> 24);
            const uint32_t run = in ? (item & 0xFFFFFFu) : 0u;
            v = *(reinterpret_cast<const uint4*>(arena + static_cast<uint64_t>(run) * kRunBytes) + (have ? g : 0u));
            return have;
        };
        uint4 a0, a1, a2, a3, b0, b1, b2, b3;
        bool ha0, ha1, ha2, ha3, hb0, hb1, hb2, hb3;
        uint32_t i = half;
        ha0 = fetch(i, a0); ha1 = fetch(i + 2, a1); ha2 = fetch(i + 4, a2); ha3 = fetch(i + 6, a3);
        while (i < cn) {
            hb0 = fetch(i + 8, b0); hb1 = fetch(i + 10, b1); hb2 = fetch(i + 12, b2); hb3 = fetch(i + 14, b3);
            if (ha0) tally4(a0);
            if (ha1) tally4(a1);
            if (ha2) tally4(a2);
            if (ha3) tally4(a3);
            i += 8;
            if (i >= cn) break;
            ha0 = fetch(i + 8, a0); ha1 = fetch(i + 10, a1); ha2 = fetch(i + 12, a2); ha3 = fetch(i + 14, a3);
            if (hb0) tally4(b0);
            if (hb1) tally4(b1);
            if (hb2) tally4(b2);
            if (hb3) tally4(b3);
            i += 8;
        }
    }
    // the bucket's listed quads, a region per workgroup of pass A: x = K + 3 bases | OK bits << 24 (bit 24 + 2t: window t counts)
    for (uint32_t p = 0; p < bp.parts; ++p) {
        const uint64_t reg = (static_cast<uint64_t>(s) * bp.parts + p) * kQuadBuckets + q;
        const uint32_t pe = bp.preg_n[reg];
        const uint32_t* ps = bp.preg + (reg << bp.preg_shift);
        for (uint32_t i = tid; i < pe; i += 512) {
            const uint32_t x = ps[i];
            const uint32_t e = (x & (RB - 1u)) | (((x >> (2 * K)) & 63u) << HB);   // the quad's entry
            const uint32_t i1 = e & (TB - 1u), i2 = (e >> 4) & (TB - 1u);
            const uint32_t w0 = (x >> 24) & 1u, w1 = (x >> 26) & 1u, w2 = (x >> 28) & 1u, w3 = (x >> 30) & 1u;
            if (w0 & w1) atomicAdd(&tab[i1], 1u);
            else {
                if (w0) atomicAdd(&rr[i1 & (RB - 1u)], 1u);
                if (w1) atomicAdd(&rr[RB + (i1 >> 2)], 1u);
            }
            if (w2 & w3) atomicAdd(&tab[TB + i2], 1u);
            else {
                if (w2) atomicAdd(&rr[2 * RB + (i2 & (RB - 1u))], 1u);
                if (w3) atomicAdd(&rr[3 * RB + (i2 >> 2)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* out = bp.bucket_hist + (static_cast<uint64_t>(s) * kQuadBuckets + q) * (4 * RB);
    for (uint32_t j = tid; j < RB; j += 512) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(tab + 4 * j), v1 = *reinterpret_cast<const uint4*>(tab + TB + 4 * j);
        const uint32_t r0 = rr[j] + tab[j] + tab[j + RB] + tab[j + 2 * RB] + tab[j + 3 * RB];
        const uint32_t r1 = rr[RB + j] + v0.x + v0.y + v0.z + v0.w;
        const uint32_t r2 = rr[2 * RB + j] + tab[TB + j] + tab[TB + j + RB] + tab[TB + j + 2 * RB] + tab[TB + j + 3 * RB];
        const uint32_t r3 = rr[3 * RB + j] + v1.x + v1.y + v1.z + v1.w;
        const uint32_t jr = pair_reverse(j, K - 4);
        out[jr] = r0;
        out[RB + jr] = r1;
        out[2 * RB + jr] = r2;
        out[3 * RB + jr] = r3;
    }
}

// Pass C of the quad route: every k-mer code adds the four counters that can name it -- the code as window
// t = 0 .. 3 of a quad: bucket = its bases K-4-t .. K-1-t, index = its other bases -- to the histogram, which already
// holds pass A's direct counts.  (Indices with the LAST base lowest, as pass B stored them.)
// A workgroup takes a tile of 4096 consecutive codes (their last six bases) and goes over it once per window type in
// the order pass B's arrays lie in memory: for t < 3 the tile's codes of one bucket are 16 neighbouring words of its
// array t, for t = 3 64 -- with a thread per code of a run of 256, as until round 5, the 256 codes of t = 0 sat in 256
// different buckets, one word from each line: 1.37 ms per 512 samples, 2.2 TB/s.  The sums meet in an LDS copy of the
// tile (XOR-swizzled: the codes of one bucket's words lie 256, 1024 or 16 apart), which is added to the histogram in code order.
constexpr uint32_t kMergeTile = 4096;
template <int K>
__global__ __launch_bounds__(256) void vk_quad_merge_kernel(BucketParams bp, uint32_t* __restrict__ hist_out) {
    static_assert(K >= 8, "a tile fixes the code's first K - 6 bases, and window type 3 needs one of them");
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr uint32_t RB = 1u << (2 * K - 8);
    constexpr uint32_t TILES = NCODE / kMergeTile;
    __shared__ uint32_t tile[kMergeTile];
    const uint32_t s = blockIdx.x / TILES, base = (blockIdx.x % TILES) * kMergeTile, tid = threadIdx.x;
    const uint32_t* bh = bp.bucket_hist + static_cast<uint64_t>(s) * kQuadBuckets * (4 * RB);
    auto phys = [](uint32_t local) { return local ^ (local >> 8); };   // (bits 8 .. 11 onto bits 0 .. 3)
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) {
        uint32_t v[16], at[16];
#pragma unroll
        for (uint32_t m = 0; m < 16; ++m) {
            const uint32_t e = m * 256u + tid;
            // the e-th word of the tile in array t's order -> its code's place in the tile
            uint32_t local;
            if (t == 3) local = e;
            else {
                const uint32_t w = e & 15u, qc = pair_reverse(e >> 4, 4);
                local = t == 0 ? (w << 8) | qc : (t == 1 ? ((w >> 2) << 10) | (qc << 2) | (w & 3u) : (qc << 4) | w);
            }
            const uint32_t code = base | local;
            // code = [first K-4-t bases | four bucket bases | last t bases], first base most significant
            const uint32_t lastb = code & ((1u << (2 * t)) - 1u);
            const uint32_t qc = (code >> (2 * t)) & 0xFFu;                 // the bucket's bases, first one most significant
            const uint32_t firstb = code >> (2 * t + 8);
            const uint32_t q = pair_reverse(qc, 4);                          // pass A's bucket number: first base lowest
            v[m] = bh[(static_cast<uint64_t>(q) * 4 + t) * RB + ((firstb << (2 * t)) | lastb)];
            at[m] = phys(local);
        }
        if (t != 0) __syncthreads();
#pragma unroll
        for (uint32_t m = 0; m < 16; ++m) {
            if (t == 0) tile[at[m]] = v[m];
            else tile[at[m]] += v[m];
        }
    }
    __syncthreads();
    uint32_t* out = hist_out + static_cast<uint64_t>(s) * NCODE + base;
#pragma unroll
    for (uint32_t m = 0; m < 16; ++m) {
        const uint32_t local = m * 256u + tid;
        out[local] += tile[phys(local)];
    }
}

// One thread per sample: the line phase each wave ended with must be the phase
// the next wave recovered for itself, and the file must end after a quality line.
__global__ void vk_check_kernel(const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
                                const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
                                const uint32_t* __restrict__ wavephase, uint32_t* __restrict__ status) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    uint32_t st = 0;
    const uint64_t len = lens[s];
    if (len) {
        const uint8_t* sbase = fastq + offs[s];
        if (sbase[0] != '@') st |= VK_ST_BAD_START;
        // the third line of the first record must be the '+' line: catches FASTA and wrapped
        // (multi-line) FASTQ, whose line counts could otherwise look consistent by accident
        // (sixteen bytes per load: byte by byte this one thread's ~170 dependent loads were the whole 0.06-0.09 ms of the kernel)
        uint32_t seen = 0;
        const uint64_t lim = len < 65536 ? len : 65536;
        bool found = false;
        for (uint64_t g = 0; g < lim && !found; g += 16) {
            const uint4 v = load_granule(sbase, g, len);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t j = 0; j < 16 && !found; ++j) {
                const uint64_t p = g + j;
                if (p < lim && ((w[j >> 2] >> (8u * (j & 3u))) & 0xFFu) == '\n' && ++seen == 2) {
                    if (p + 1 < len && sbase[p + 1] != '+') st |= VK_ST_BAD_START;
                    found = true;
                }
            }
        }
        uint32_t prev = 0;  // phase at byte 0
        const uint32_t* wp = wavephase + static_cast<uint64_t>(s) * parts * kWaves;
        for (uint32_t i = 0; i < parts * kWaves; ++i) {
            uint32_t v = wp[i];
            if (v & 0x80u) continue;
            if ((v & 3u) != prev) st |= VK_ST_BAD_PHASE;
            prev = (v >> 2) & 3u;
        }
        uint32_t want = (sbase[len - 1] == '\n') ? 0u : 3u;
        if (prev != want) st |= VK_ST_BAD_PHASE;
    }
    status[s] = st;
}

}  // namespace

#endif  // VK_COUNT_H
