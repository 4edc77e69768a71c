// vkimg.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of include/vkimg.h.
//
// Hot path of varKoder's `image` command (reference: varKoder/commands/image.py
// count_kmers :727-806 -> dsk, make_image :808-936 -> dsk2ascii + pandas/NumPy):
//
//   K1  vk_count_kernel   FASTQ text in HBM -> forward-strand k-mer histogram u32[4^k]
//   K1c vk_check_kernel   line-phase consistency of the byte ranges -> status word
//   K2  vk_image_kernel   strand merge + pixel scatter (count+1) + sort + 256-quantile
//                         rank binning -> uint8 image
//   vk_synth_kernel       synthetic FASTQ generator of BASELINE.md section 4
//
// Design notes live in DESIGN.md; the short version for K1:
//   * one 1024-thread workgroup per (sample, byte-range part);
//     its 16 wavefronts run WITHOUT workgroup barriers in steady state: every
//     wave streams its own contiguous byte range in 4 KiB pieces (4 coalesced
//     16-B loads per lane, prefetched one piece ahead), transposes the piece
//     through a private 4 KiB LDS slot so that each lane owns 64 contiguous
//     bytes, and classifies them with 32-bit SWAR arithmetic (vk_lane.h);
//   * FASTQ line phase (header/sequence/plus/quality) comes from a wave-level
//     prefix sum of newline counts; the phase at a range start is recovered
//     locally from the '@' / '+' framing, so byte ranges are independent;
//   * k-mer windows are counted forward-strand only into an LDS histogram
//     (ds_add_u32); the strand merge happens once per sample in K2;
//   * 4^k u32 > LDS for k = 8, 9: windows are bucketed through wave-private LDS queues into
//     16 streams per sample in HBM and replayed into 4^k/16-bin LDS histograms (vk_bucket_kernel,
//     vk_bucket_count_kernel).
//   vk_remap_kernel / vk_preprocess_kernel: `convert`'s remap and the input side of `query`.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "vkimg.h"
#include "vk_lane.h"

namespace {

constexpr int kWaves = 16;             // wavefronts per count workgroup
constexpr int kCountThreads = kWaves * 64;
constexpr int kPiece = 4096;           // bytes per wave iteration (64 lanes x 64 B)
constexpr uint32_t kMaxBins = 16384;   // u32 LDS histogram bins per workgroup (64 KiB)

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
    uint32_t x = c;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32 - 2 * k);
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
// significant, vk_lane.h).  `st` is the wave's private 4 KiB LDS slot,
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
template <int K>
__device__ __forceinline__ void windows_lds(uint32_t ch, const uint32_t C[4], const uint32_t ok[4],
                                            uint32_t lds_base) {
    static_assert(2 * K + 2 <= 16, "paired extraction needs the field << 2 to fit 16 bits");
    const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
    constexpr uint32_t kMask4 = ((1u << (2 * K)) - 1u) << 2;
    const uint32_t one = 1u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // even bits: OK of positions 0..15 of this dword; odd bits: the same rotated by 8
        // positions -> shifting out the top bit twice yields positions (i - 8, i), i = 15..8
        uint32_t w = ok[g] | (vkl::alignbit(ok[g], ok[g], 16u) << 1);
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
                "s_mov_b64 exec, -1"
                : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(one)
                : "memory");
        }
    }
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

template <int K, bool SUB, typename Windows>
__device__ __forceinline__ void wave_stream(const uint8_t* __restrict__ sbase, uint64_t len, uint64_t w0,
                                            uint64_t w1, uint4* st, const uint4* below, const uint4* above,
                                            int lane, Windows windows, uint32_t& ph_start, uint32_t& ph_end,
                                            SubWave& sw) {
    const uint32_t ph0 = (w0 != 0) ? sync_phase(sbase, w0, len, reinterpret_cast<uint64_t*>(st), lane) : 0u;
    ph_start = ph0;

    // Pieces start one 64-byte block BEFORE the range: lane 0 of piece 0 (the "pre-block")
    // only supplies the k-1 bases of context and its windows are not counted.  From then on
    // lane 0 takes its context from lane 63 of the previous piece.
    const long long o0 = static_cast<long long>(w0) - 64;
    const uint64_t npieces = (w1 - w0 + 64 + kPiece - 1) / kPiece;
    uint4 r0, r1, r2, r3;
    // A piece that lies wholly inside [0, w1) (all but the first and last of a range) is
    // loaded with four unguarded 16-byte loads off one address; edge pieces zero-fill.
    auto load_piece = [&](uint64_t piece) {
        const long long pb = o0 + static_cast<long long>(piece) * kPiece;
        if (pb >= 0 && static_cast<uint64_t>(pb) + kPiece <= w1) {  // wave-uniform
            const uint4* g = reinterpret_cast<const uint4*>(sbase + pb) + lane;
            r0 = g[0];
            r1 = g[64];
            r2 = g[128];
            r3 = g[192];
        } else {
            const long long p = pb + static_cast<long long>(lane) * 16;
            r0 = load_granule_s(sbase, p, w1);
            r1 = load_granule_s(sbase, p + 1024, w1);
            r2 = load_granule_s(sbase, p + 2048, w1);
            r3 = load_granule_s(sbase, p + 3072, w1);
        }
    };
    load_piece(0);
    uint32_t carry_c = 0u, carry_bad = 0x55555555u;
    uint32_t pph = 0;  // line phase at the start of the current piece
    uint32_t sub_carry = 0u;  // SUB: is the read that runs into the current piece taken?
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
    for (uint64_t it = 0; it < npieces; ++it) {
        VK_STAMP(t0);
        // transpose through LDS: coalesced rows in, 64 contiguous bytes per lane out
        wave_lds_fence();
        st[lane] = r0;
        st[64 + lane] = r1;
        st[128 + lane] = r2;
        st[192 + lane] = r3;
        wave_lds_fence();
        uint4 q0 = st[lane * 4 + 0], q1 = st[lane * 4 + 1], q2 = st[lane * 4 + 2], q3 = st[lane * 4 + 3];
        wave_lds_fence();
        if (it + 1 < npieces) load_piece(it + 1);  // prefetch the next piece under the SWAR work
        const uint32_t d[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
        VK_STAMP(t1);
        vkl::LaneBits lb;
        const uint32_t c = __any(vkl::has_non_ascii(d)) ? vkl::classify<false>(d, lb) : vkl::classify<true>(d, lb);

        VK_STAMP(t2);
        // newline prefix over the wave -> line phase at the start of each lane's block
        const uint32_t incl = wave_inclusive_sum(c);
        const uint32_t total = lane_bcast(incl, 63);
        if (it == 0) pph = ph0 - lane_bcast(c, 0);  // the pre-block's newlines precede w0
        const uint32_t lph = (pph + incl - c) & 3u;

        vkl::Mask128 seq;
        const bool degenerate = __any(c > 3u);
        uint32_t s_raw = 0;
        if (degenerate) seq = vkl::seq_mask_general(lb.NL, lph);
        else seq = vkl::seq_mask_fast(lb.NL, lph, tbl_below, tbl_above, s_raw);
        uint32_t bad[4], ok[4];
        vkl::bad_mask(lb, seq, bad);

        const uint32_t badh = wave_prev_lane(bad[3], carry_bad);
        const uint32_t ch = wave_prev_lane(lb.C[3], carry_c);
        carry_bad = lane_bcast(bad[3], 63);
        carry_c = lane_bcast(lb.C[3], 63);

        vkl::ok_mask<K>(badh, bad, ok);
        if (it == 0 && lane == 0) { ok[0] = 0u; ok[1] = 0u; ok[2] = 0u; ok[3] = 0u; }

        if constexpr (SUB) {
            // Which read does each position belong to, and is that read taken?  A block either has
            // an anchor (the newline that ends a header line) and decides for what follows it, or
            // inherits the decision of the nearest anchor before it: a max-scan over
            // (lane + 1) << 1 | take, seeded with the decision carried in from the previous piece.
            const uint64_t base = static_cast<uint64_t>(o0 + static_cast<long long>(it) * kPiece) + 64ull * lane;
            if (it == 0 && w0 != 0 && (pph & 3u) == 1u) {
                // the range is entered inside a sequence line whose header ended before the pre-block
                const uint64_t a = last_newline_before(sbase, static_cast<uint64_t>(o0), lane);
                sub_carry = (a != ~0ull && vkl::sample_take(sw.seed, a, sw.threshold)) ? 1u : 0u;
            }
            uint32_t first[4], inc[4], anchors, take;
            if (degenerate) {
                anchors = vkl::sample_strings_general(lb.NL, lph, base, sw.seed, sw.threshold, first, inc, take);
            } else {
                anchors = (lph != 1u && s_raw <= 64u) ? 1u : 0u;
                take = (anchors && vkl::sample_take(sw.seed, base + s_raw - 1u, sw.threshold)) ? 1u : 0u;
                const uint32_t f = anchors ? 0u : 0xFFFFFFFFu, n = take ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g) { first[g] = f; inc[g] = n; }
            }
            const uint32_t v = anchors ? (((static_cast<uint32_t>(lane) + 1u) << 1) | take) : 0u;
            const uint32_t scan = wave_inclusive_max(v);
            const uint32_t inherited = max(wave_prev_lane(scan, 0u), sub_carry) & 1u;
            sub_carry = max(lane_bcast(scan, 63), sub_carry) & 1u;
            const uint32_t inh = 0u - inherited;
            // the pre-block belongs to the previous range; bytes at or beyond w1 are zero fill
            const bool mine = !(it == 0 && lane == 0) && base < w1;
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
__global__ __launch_bounds__(kCountThreads) void vk_count_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
    const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ wavephase, int atomic_flush, SubParams sp) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    static_assert(NCODE <= kMaxBins, "LDS histogram too large");

    // The LDS histogram is indexed by the RAW packed field (first base least
    // significant, see vk_lane.h); the flush un-reverses to the ABI's code order.
    __shared__ uint32_t hist[NCODE];
    __shared__ uint4 stage[kWaves][kPiece / 16];
    __shared__ uint4 below[66];  // below[q] = bits [0, 2q) of a 128-bit string
    __shared__ uint4 above[66];  // above[q] = ~below[q]

    const uint32_t unit = blockIdx.x;
    const uint32_t s = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

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
        auto win = [&](uint32_t ch, const uint32_t* C, const uint32_t* ok) __attribute__((always_inline)) {
            windows_lds<K>(ch, C, ok, hist_base);
        };
        SubWave sw = {0, 0, 0, 0};
        if constexpr (SUB) {
            sw.seed = sp.seeds[s];
            sw.threshold = sp.thresholds[s];
        }
        wave_stream<K, SUB>(sbase, len, wr.w0, wr.w1, &stage[wave][0], below, above, lane, win, ph_start, ph_end, sw);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the hand-written ds_add are invisible to hipcc
        if constexpr (SUB) flush_sites(sp, s, sw, lane);
    }
    if (lane == 0) wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));

    __syncthreads();
    uint32_t* out = hist_out + static_cast<uint64_t>(s) * NCODE;
    for (uint32_t i = tid; i < NCODE; i += kCountThreads) {
        const uint32_t v = hist[i];
        const uint32_t code = pair_reverse(i, K);  // ABI order: first base most significant
        if (atomic_flush) {
            if (v) atomicAdd(&out[code], v);
        } else {
            out[code] = v;
        }
    }
}

// ---- K = 8, 9: the LDS-spill path ------------------------------------------------------
// 4^K u32 counters do not fit LDS.  Pass A streams the FASTQ exactly like vk_count_kernel but
// appends the windows, two at a time (see entry_raw), to one of 16 wave-private LDS queues chosen
// by the two bases both windows of a pair share, and drains full 64-entry blocks (128 B) into
// per-(sample, queue) bucket streams in HBM.  The drain handles all 16 queues at once, four lanes
// per queue; block runs are reserved 32 at a time with one global atomic, unused run tails are
// padded with 0xFFFF.  Pass B gives every (sample, queue) one workgroup that replays its stream into
// a 2 x 4^K/16-bin LDS histogram (one half per entry type) and adds it to the global histogram.
// Pairs that cannot be queued or whose bucket is full are counted with global atomics on the spot:
// slower, still exact.
constexpr uint32_t kQueues = 16;         // queues per wave = bucket streams per sample
constexpr uint32_t kQueueCap = 128;      // u16 entries per queue (two blocks)
constexpr uint32_t kBlockEntries = 64;   // u16 entries per 128-byte bucket block
constexpr uint32_t kRunBlocks = 32;      // blocks reserved per global atomic

struct BucketParams {
    uint32_t* cursors;   // [nsamples][16] next free block of each bucket stream
    uint32_t* buckets;   // [nsamples][16][cap_blocks * 32] dwords
    uint32_t cap_blocks; // multiple of kRunBlocks
};

// Bucket entries (u16, 0xFFFF = padding).  Two windows that end at neighbouring positions p, p + 1
// (p even) share the bases p-1 and p; those four bits are the queue number q of BOTH, so one
// returning LDS atomic and one 32-bit store queue the pair.  LB = 2K - 4 bits remain per window:
//   low half,  type 0 (window ending at p):     bases p-K+1 .. p-2
//   high half, type 1 (window ending at p + 1): bases p-K+2 .. p-2, then base p+1
// The type is the half of the dword the entry sits in (blocks move as whole 128-byte units, so the
// halves never mix).  entry_raw rebuilds the raw window field (first base least significant) from
// rest | type << LB.
template <int K>
__device__ __forceinline__ uint32_t entry_raw(uint32_t q, uint32_t e) {
    constexpr uint32_t LB = 2 * K - 4;
    const uint32_t rest = e & ((1u << LB) - 1u);
    if ((e >> LB) == 0u) return (q << LB) | rest;
    return (rest & ((1u << (LB - 2)) - 1u)) | (q << (LB - 2)) | ((rest >> (LB - 2)) << (2 * K - 2));
}

__device__ __forceinline__ uint32_t quad_bcast0(uint32_t x) {  // value of lane (lane & ~3)
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(x), 0x00, 0xF, 0xF, true));
}

template <int K, bool SUB>
__global__ __launch_bounds__(kCountThreads) void vk_bucket_kernel(
    const uint8_t* __restrict__ fastq, const uint64_t* __restrict__ offs,
    const uint64_t* __restrict__ lens, uint32_t nsamples, uint32_t parts,
    uint32_t* __restrict__ hist_out, uint32_t* __restrict__ wavephase, BucketParams bp, SubParams sp) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr uint32_t LB = 2 * K - 4;               // local bits of an entry
    constexpr uint32_t LMASK = (1u << LB) - 1u;
    static_assert(LB <= 15, "entries are u16 with 0xFFFF as padding");

    __shared__ uint4 stage[kWaves][kPiece / 16];
    __shared__ uint4 below[66];
    __shared__ uint4 above[66];
    __shared__ uint4 qbuf[kWaves][kQueues * kQueueCap / 8];  // u16 entries, eight per uint4
    __shared__ uint32_t qcnt[kWaves][kQueues];
    __shared__ uint32_t runbase[kWaves][kQueues];
    __shared__ uint32_t runleft[kWaves][kQueues];

    const uint32_t unit = blockIdx.x;
    const uint32_t s = unit / parts;
    const uint32_t part = unit % parts;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    fill_mask_tables(below, above, tid);
    if (lane < static_cast<int>(kQueues)) {
        qcnt[wave][lane] = 0u;
        runbase[wave][lane] = 0u;
        runleft[wave][lane] = 0u;
    }
    __syncthreads();

    const uint8_t* sbase = fastq + offs[s];
    const uint64_t len = lens[s];
    const WaveRange wr = wave_range(len, parts, part, wave);
    uint32_t* hist_s = hist_out + static_cast<uint64_t>(s) * NCODE;
    uint16_t* q16 = reinterpret_cast<uint16_t*>(&qbuf[wave][0]);

    // Four lanes per queue: q = lane / 4, every lane moves 32 B of a 128-byte block.
    const uint32_t q = static_cast<uint32_t>(lane) >> 2, sub = static_cast<uint32_t>(lane) & 3u;
    const uint32_t cap_blocks = bp.cap_blocks;
    uint32_t* const cursor = bp.cursors + (s * kQueues + q);
    uint4* const gq = reinterpret_cast<uint4*>(bp.buckets + (static_cast<uint64_t>(s) * kQueues + q) * cap_blocks * 32u);

    // Drain `nb` (0..2, per queue) blocks from the front of every queue.  All lanes call it.
    auto drain_all = [&](uint32_t n, uint32_t nb) __attribute__((always_inline)) {
        uint32_t base = runbase[wave][q], left = runleft[wave][q];
        const bool need = nb > left;
        if (need && left) {  // the rest of the old run (fewer than nb <= 2 blocks) stays padding
            const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            gq[static_cast<uint64_t>(base) * 8u + sub * 2u] = ff;
            gq[static_cast<uint64_t>(base) * 8u + sub * 2u + 1u] = ff;
        }
        uint32_t nbase = 0;
        if (need && sub == 0) nbase = atomicAdd(cursor, kRunBlocks);
        nbase = quad_bcast0(nbase);
        if (need) {
            base = nbase;
            left = (nbase + kRunBlocks <= cap_blocks) ? kRunBlocks : 0u;
        }
        const bool store = left >= nb;  // false only when the bucket is full
        const uint4* src = &qbuf[wave][q * (kQueueCap / 8)];
#pragma unroll
        for (uint32_t b = 0; b < 2; ++b) {
            if (b < nb) {
                const uint4 v0 = src[b * 8u + sub * 2u], v1 = src[b * 8u + sub * 2u + 1u];
                if (store) {
                    gq[static_cast<uint64_t>(base + b) * 8u + sub * 2u] = v0;
                    gq[static_cast<uint64_t>(base + b) * 8u + sub * 2u + 1u] = v1;
                } else {  // bucket full: count these entries directly (exact, slow)
                    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t lo = w[j] & 0xFFFFu, hi = w[j] >> 16;
                        if (lo != 0xFFFFu) atomicAdd(&hist_s[pair_reverse(entry_raw<K>(q, lo), K)], 1u);
                        if (hi != 0xFFFFu) atomicAdd(&hist_s[pair_reverse(entry_raw<K>(q, hi | (1u << LB)), K)], 1u);
                    }
                }
            }
        }
        // move the remainder (< one block) to the front: block nb -> block 0 (nb = 1 only;
        // after two blocks nothing is left because a queue holds two)
        uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
        if (nb == 1) {
            k0 = src[8u + sub * 2u];
            k1 = src[8u + sub * 2u + 1u];
        }
        wave_lds_fence();
        if (nb == 1) {
            qbuf[wave][q * (kQueueCap / 8) + sub * 2u] = k0;
            qbuf[wave][q * (kQueueCap / 8) + sub * 2u + 1u] = k1;
        }
        if (nb && sub == 0) {
            qcnt[wave][q] = 2u * (n - nb * kBlockEntries);
            runbase[wave][q] = store ? base + nb : base;
            runleft[wave][q] = store ? left - nb : 0u;
        }
        wave_lds_fence();
    };

    uint32_t ph_start = 0, ph_end = 0;
    if (!wr.empty) {
        // LDS byte addresses: the wave's 16 counters (which count BYTES, 4 per pair) and its queues
        auto lds_addr = [](const void* p) {
            return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p));
        };
        const uint32_t cnt_base = lds_addr(&qcnt[wave][0]);
        // data address of queue qq = (counter address << 6) + data_skew, counters being 4 B apart
        const uint32_t data_skew = lds_addr(&qbuf[wave][0]) - (cnt_base << 6);
        // x = the K + 1 bases p-K+1 .. p+1 (2 bits each, first base lowest); okw bits `bit` and
        // `bit + 2` = the window ending at p / p + 1 is countable.  A missing partner becomes padding.
        // x = the K + 1 bases p-K+1 .. p+1 (2 bits each, first base lowest); okw bits `bit` and
        // `bit + 2` = the window ending at p / p + 1 is countable.  A missing partner becomes padding.
        // (Issuing the atomics of four pairs back to back behind one wait was measured: 3 % slower,
        // the loop is bound by VALU + SALU issue, not by the LDS round trip.)
        auto emit_pair = [&](uint32_t x, uint32_t okw, int bit) __attribute__((always_inline)) {
            constexpr uint32_t FMASK = (1u << (2 * K)) - 1u;
            const uint32_t caddr = cnt_base + 4u * __builtin_amdgcn_ubfe(x, LB, 4);
            uint32_t at;  // returning LDS atomic on the queue's byte counter
            asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(at) : "v"(caddr), "v"(4u) : "memory");
            const uint32_t rest_a = x & LMASK;
            // bit-field extracts spelled out: hipcc turns the builtins back into shift pairs here
            uint32_t low_b, top_b, keep0, keep1;
            asm("v_bfe_u32 %0, %1, 2, %2" : "=v"(low_b) : "v"(x), "n"(LB - 2));
            asm("v_bfe_u32 %0, %1, %2, 2" : "=v"(top_b) : "v"(x), "n"(2 * K));
            const uint32_t rest_b = (top_b << (LB - 2)) | low_b;
            // all ones where the window counts: 0xFFFF in the half of a window that does not
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep0) : "v"(okw), "n"(bit));
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(keep1) : "v"(okw), "n"(bit + 2));
            const uint32_t w = (rest_a | (rest_b << 16)) | ~__builtin_amdgcn_perm(keep1, keep0, 0x05040100u);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(at) : : "memory");
            if (at < 2u * kQueueCap) {
                const uint32_t daddr = (caddr << 6) + data_skew + at;
                asm volatile("ds_write_b32 %0, %1" : : "v"(daddr), "v"(w) : "memory");
            } else {  // queue full: exact slow path
                if (keep0) atomicAdd(&hist_s[pair_reverse(x & FMASK, K)], 1u);
                if (keep1) atomicAdd(&hist_s[pair_reverse((x >> 2) & FMASK, K)], 1u);
            }
        };
        auto after_group = [&]() __attribute__((always_inline)) {
            wave_lds_fence();
            uint32_t n = qcnt[wave][q] >> 1;  // bytes -> entries
            if (n > kQueueCap) n = kQueueCap;
            const uint32_t nb = n / kBlockEntries;
            if (__any(nb != 0u)) drain_all(n, nb);
        };
        auto win = [&](uint32_t ch, const uint32_t* C, const uint32_t* ok) __attribute__((always_inline)) {
            const uint32_t v[5] = {ch, C[0], C[1], C[2], C[3]};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int p = 16 * g + 2 * j;  // windows ending at p and p + 1
                    if (ok[g] & (5u << (4 * j))) {
                        const int o = 32 + 2 * (p - K + 1);  // bit offset of base p-K+1 in [ch | C]
                        const int word = o >> 5, sh = o & 31;
                        uint32_t x;
                        if (sh == 0) x = v[word];
                        else if (word == 4) x = v[4] >> sh;  // the last pair ends exactly at bit 160
                        else x = vkl::alignbit(v[word + 1], v[word], static_cast<uint32_t>(sh));
                        emit_pair(x, ok[g], 4 * j);
                    }
                }
                after_group();  // 16 positions of every lane done: drain the queues that hold a block
            }
        };
        SubWave sw = {0, 0, 0, 0};
        if constexpr (SUB) {
            sw.seed = sp.seeds[s];
            sw.threshold = sp.thresholds[s];
        }
        wave_stream<K, SUB>(sbase, len, wr.w0, wr.w1, &stage[wave][0], below, above, lane, win, ph_start, ph_end, sw);
        if constexpr (SUB) flush_sites(sp, s, sw, lane);
        // final drain: pad the last partial block of every queue, write it, then the rest of every run
        wave_lds_fence();
        uint32_t n = qcnt[wave][q] >> 1;
        if (n > kQueueCap) n = kQueueCap;
        const uint32_t nb = (n + kBlockEntries - 1) / kBlockEntries;
        for (uint32_t e = n + sub; e < nb * kBlockEntries; e += 4) q16[q * kQueueCap + e] = 0xFFFFu;
        wave_lds_fence();
        drain_all(nb * kBlockEntries, nb);
        const uint32_t left = runleft[wave][q], base = runbase[wave][q];
        const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        for (uint32_t b = 0; b < left; ++b) {
            gq[static_cast<uint64_t>(base + b) * 8u + sub * 2u] = ff;
            gq[static_cast<uint64_t>(base + b) * 8u + sub * 2u + 1u] = ff;
        }
    }
    if (lane == 0) wavephase[unit * kWaves + wave] = wr.empty ? 0x80u : (0x40u | ph_start | (ph_end << 2));
}

// Pass B: one workgroup per (sample, queue) replays the bucket stream into LDS and adds the
// 4^K/16 counters to the histogram (which already holds pass A's direct counts).
template <int K>
__global__ __launch_bounds__(kCountThreads) void vk_bucket_count_kernel(BucketParams bp,
                                                                         uint32_t* __restrict__ hist_out) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr uint32_t LB = 2 * K - 4;
    constexpr uint32_t BINS = 2u << LB;  // indexed by the entry: type bit | rest
    __shared__ uint32_t hist[BINS];
    const uint32_t s = blockIdx.x / kQueues, q = blockIdx.x % kQueues;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < BINS; i += kCountThreads) hist[i] = 0u;
    __syncthreads();
    uint32_t nblk = bp.cursors[s * kQueues + q];
    if (nblk > bp.cap_blocks) nblk = bp.cap_blocks;
    const uint4* src = reinterpret_cast<const uint4*>(bp.buckets + (static_cast<uint64_t>(s) * kQueues + q) *
                                                                       bp.cap_blocks * 32u);
    const uint64_t n16 = static_cast<uint64_t>(nblk) * 8u;  // 16-byte groups
    auto tally = [&](const uint4& v) __attribute__((always_inline)) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = w[j] & 0xFFFFu, hi = w[j] >> 16;
            if (lo != 0xFFFFu) atomicAdd(&hist[lo], 1u);
            if (hi != 0xFFFFu) atomicAdd(&hist[hi + (1u << LB)], 1u);  // the high half holds the type-1 entries
        }
    };
    // four 16-byte loads in flight per thread: the stream is read once, latency is all there is to hide
    uint64_t i = tid;
    for (; i + 3ull * kCountThreads < n16; i += 4ull * kCountThreads) {
        const uint4 v0 = src[i], v1 = src[i + kCountThreads], v2 = src[i + 2ull * kCountThreads],
                    v3 = src[i + 3ull * kCountThreads];
        tally(v0);
        tally(v1);
        tally(v2);
        tally(v3);
    }
    for (; i < n16; i += kCountThreads) tally(src[i]);
    __syncthreads();
    uint32_t* out = hist_out + static_cast<uint64_t>(s) * NCODE;
    for (uint32_t i = tid; i < BINS; i += kCountThreads) {
        const uint32_t v = hist[i];
        // type-1 entries of this queue and type-0 entries of another can name the same code
        if (v) atomicAdd(&out[pair_reverse(entry_raw<K>(q, i), K)], v);
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
        uint32_t seen = 0;
        const uint64_t lim = len < 65536 ? len : 65536;
        for (uint64_t p = 0; p < lim; ++p) {
            if (sbase[p] == '\n' && ++seen == 2) {
                if (p + 1 < len && sbase[p + 1] != '+') st |= VK_ST_BAD_START;
                break;
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

__global__ __launch_bounds__(kImgThreads) void vk_image_count_kernel(
    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ pix, int k, uint32_t npix,
    uint32_t npad, uint32_t* __restrict__ scratch, uint8_t* __restrict__ img, uint32_t* __restrict__ flags) {
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

    for (uint32_t i = tid; i < npad; i += kImgThreads) val[i] = 0u;
    if (tid == 0) { novf = 0u; any_hi = 0u; }
    __syncthreads();
    for (uint32_t c = tid; c < ncode; c += kImgThreads) {  // strand merge + scatter, as in vk_image_kernel
        uint32_t r = revcomp_code(c, k);
        uint32_t tot = (r == c) ? h[c] : h[c] + h[r];
        val[pix[c]] = tot + 1u;
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
        if (pass == 1 && any_hi == 0u) break;  // uniform: nothing in [32768, 65536)
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

// ---------------------------------------------------------------- C ABI ------

struct vk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipError_t last = hipSuccess;
    uint32_t* d_pix[10] = {};
    uint32_t npix[10] = {};
    // workspaces (grown on demand, never inside a timed launch after warm-up)
    uint64_t* d_desc = nullptr;   // offsets | lengths
    size_t desc_cap = 0;
    uint64_t* h_desc = nullptr;   // pinned mirror of d_desc (descriptor cache)
    size_t h_desc_cap = 0;
    uint32_t desc_n = 0;
    uint32_t* d_wavephase = nullptr;
    size_t wavephase_cap = 0;
    uint32_t* d_scratch = nullptr;
    size_t scratch_cap = 0;
    uint32_t* d_spill = nullptr;  // k >= 8: bucket cursors + bucket streams
    size_t spill_cap = 0;
    size_t spill_budget = 96ull << 30;  // bytes of HBM the spill path may use at a time
    // host-call staging
    uint64_t* d_sub = nullptr;    // subsampling launches: seeds | thresholds
    size_t sub_cap = 0;
    uint8_t* d_stage = nullptr;
    size_t stage_cap = 0;
    uint32_t* d_hist1 = nullptr;
    uint32_t* d_status1 = nullptr;
    uint8_t* d_img1 = nullptr;
    uint32_t last_grid = 0, last_block = 0, last_lds = 0;
    bool image_sort_only = false;  // VKIMG_IMAGE_SORT_ONLY=1: always take the sort kernel (tests, A/B timing)
};

#define VK_HIP(ctx, call)                     \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            (ctx)->last = e_;                 \
            return VK_EHIP;                   \
        }                                     \
    } while (0)

namespace {

int ensure(vk_ctx* ctx, void** p, size_t* cap, size_t need) {
    if (*cap >= need) return VK_OK;
    if (*p) VK_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = need + need / 4 + 256;
    VK_HIP(ctx, hipMalloc(p, n));
    *cap = n;
    return VK_OK;
}

// Sample descriptors go through a pinned host mirror; an unchanged batch (the
// steady state of a pipeline that recycles its buffers) is not uploaded again.
int upload_desc(vk_ctx* ctx, const uint64_t* offsets, const uint64_t* lengths, uint32_t n) {
    const size_t bytes = static_cast<size_t>(n) * sizeof(uint64_t);
    if (ctx->desc_n == n && ctx->h_desc && memcmp(ctx->h_desc, offsets, bytes) == 0 &&
        memcmp(ctx->h_desc + n, lengths, bytes) == 0)
        return VK_OK;
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // a previous upload may still read h_desc
    if (ctx->h_desc_cap < 2 * bytes) {
        if (ctx->h_desc) VK_HIP(ctx, hipHostFree(ctx->h_desc));
        ctx->h_desc = nullptr;
        ctx->h_desc_cap = 0;
        VK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_desc), 2 * bytes + 4096, hipHostMallocDefault));
        ctx->h_desc_cap = 2 * bytes + 4096;
    }
    memcpy(ctx->h_desc, offsets, bytes);
    memcpy(ctx->h_desc + n, lengths, bytes);
    ctx->desc_n = n;
    VK_HIP(ctx, hipMemcpyAsync(ctx->d_desc, ctx->h_desc, 2 * bytes, hipMemcpyHostToDevice, ctx->stream));
    return VK_OK;
}

uint32_t npad_of(uint32_t npix) {
    uint32_t p = 1;
    while (p < npix) p <<= 1;
    return p;
}

template <int K>
int launch_count(vk_ctx* ctx, const uint8_t* d_fastq, const uint64_t* d_offs, const uint64_t* d_lens,
                 uint32_t nsamples, uint32_t parts, uint64_t /*maxlen*/, uint32_t* d_hist, const SubParams* sub) {
    const uint32_t grid = nsamples * parts;
    const int atomic_flush = parts > 1 ? 1 : 0;
    if (atomic_flush)
        VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * (1u << (2 * K)) * sizeof(uint32_t),
                                   ctx->stream));
    ctx->last_grid = grid;
    ctx->last_block = kCountThreads;
    ctx->last_lds = (1u << (2 * K)) * 4u + kWaves * kPiece + 2 * 66 * 16;
    if (sub)
        hipLaunchKernelGGL((vk_count_kernel<K, true>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                           d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, *sub);
    else
        hipLaunchKernelGGL((vk_count_kernel<K, false>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                           d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, SubParams{});
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

// k = 8, 9: bucket pass + replay pass, in sub-batches that fit the spill budget.
template <int K>
int launch_spill(vk_ctx* ctx, const uint8_t* d_fastq, const uint64_t* d_offs, const uint64_t* d_lens,
                 uint32_t nsamples, uint32_t parts, uint64_t maxlen, uint32_t* d_hist, const SubParams* sub) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    // windows <= bytes/2; a uniform sample sends ~0.45*bytes/16 entries to each queue.  Room for
    // bytes/16 entries (2.2x) plus the run each wave may leave unfinished.
    uint64_t cap = maxlen / kQueues / kBlockEntries + 1 + static_cast<uint64_t>(parts) * kWaves * kRunBlocks;
    cap = (cap + kRunBlocks - 1) / kRunBlocks * kRunBlocks;
    if (cap > 0xFFFFFFFFull - kRunBlocks) return VK_EINVAL;
    const size_t per_sample = static_cast<size_t>(kQueues) * cap * 128u;
    // never plan for more than three quarters of what is free (plus what this context already holds)
    size_t free_b = 0, total_b = 0;
    VK_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    size_t budget = ctx->spill_budget;
    const size_t avail = free_b / 4 * 3 + ctx->spill_cap;
    if (budget > avail) budget = avail;
    uint32_t batch = static_cast<uint32_t>(budget / per_sample);
    if (batch == 0) batch = 1;
    if (batch > nsamples) batch = nsamples;
    const size_t cursor_bytes = (static_cast<size_t>(batch) * kQueues * sizeof(uint32_t) + 255) / 256 * 256;
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_spill), &ctx->spill_cap, cursor_bytes + batch * per_sample);
    if (rc) return rc;
    BucketParams bp;
    bp.cursors = ctx->d_spill;
    bp.buckets = ctx->d_spill + cursor_bytes / sizeof(uint32_t);
    bp.cap_blocks = static_cast<uint32_t>(cap);
    VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * NCODE * sizeof(uint32_t), ctx->stream));
    ctx->last_block = kCountThreads;
    ctx->last_lds = kWaves * kPiece + 2 * 66 * 16 + kWaves * kQueues * kQueueCap * 2 + 3 * kWaves * kQueues * 4;
    for (uint32_t s0 = 0; s0 < nsamples; s0 += batch) {
        const uint32_t n = nsamples - s0 < batch ? nsamples - s0 : batch;
        VK_HIP(ctx, hipMemsetAsync(bp.cursors, 0, static_cast<size_t>(n) * kQueues * sizeof(uint32_t), ctx->stream));
        ctx->last_grid = n * parts;
        if (sub) {
            SubParams sp = *sub;  // this sub-batch's slice of the per-sample arrays
            sp.seeds += s0;
            sp.thresholds += s0;
            if (sp.sites) sp.sites += 2ull * s0;
            hipLaunchKernelGGL((vk_bucket_kernel<K, true>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                               d_fastq, d_offs + s0, d_lens + s0, n, parts, d_hist + static_cast<size_t>(s0) * NCODE,
                               ctx->d_wavephase + static_cast<size_t>(s0) * parts * kWaves, bp, sp);
        } else {
            hipLaunchKernelGGL((vk_bucket_kernel<K, false>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                               d_fastq, d_offs + s0, d_lens + s0, n, parts, d_hist + static_cast<size_t>(s0) * NCODE,
                               ctx->d_wavephase + static_cast<size_t>(s0) * parts * kWaves, bp, SubParams{});
        }
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((vk_bucket_count_kernel<K>), dim3(n * kQueues), dim3(kCountThreads), 0, ctx->stream, bp,
                           d_hist + static_cast<size_t>(s0) * NCODE);
        VK_HIP(ctx, hipGetLastError());
    }
    return VK_OK;
}

uint32_t choose_parts(uint32_t nsamples, uint64_t maxlen) {
    // enough workgroups to fill 256 CUs a few times over, but never ranges so
    // small that the per-wave phase sync dominates
    uint32_t parts = 1;
    while (static_cast<uint64_t>(nsamples) * parts < 512 && (maxlen / (parts * 2)) >= (1u << 20)) parts *= 2;
    return parts;
}

}  // namespace

extern "C" {

int vk_abi_version(void) { return 1; }

const char* vk_strerror(int status) {
    switch (status) {
        case VK_OK: return "ok";
        case VK_EINVAL: return "invalid argument";
        case VK_EHIP: return "HIP runtime error";
        case VK_ENOMAP: return "no k-mer mapping installed for this k";
        case VK_EFORMAT: return "inconsistent FASTQ framing";
        case VK_ENOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char* vk_last_hip_error(const vk_ctx* ctx) {
    if (!ctx || ctx->last == hipSuccess) return "";
    return hipGetErrorString(ctx->last);
}

int vk_ctx_create(int device, void* stream, int own_stream, vk_ctx** out) {
    if (!out) return VK_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return VK_EHIP;
    vk_ctx* ctx = new (std::nothrow) vk_ctx();
    if (!ctx) return VK_ENOMEM;
    ctx->device = device;
    {
        const char* e = getenv("VKIMG_IMAGE_SORT_ONLY");
        ctx->image_sort_only = e && e[0] == '1';
    }
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return VK_EHIP; }
    if (!own_stream) {
        ctx->stream = static_cast<hipStream_t>(stream);  // NULL = the device's default stream
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return VK_EHIP; }
        ctx->own_stream = true;
    }
    *out = ctx;
    return VK_OK;
}

void vk_ctx_destroy(vk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 10; ++k)
        if (ctx->d_pix[k]) (void)hipFree(ctx->d_pix[k]);
    void* ptrs[] = {ctx->d_desc, ctx->d_wavephase, ctx->d_scratch, ctx->d_spill, ctx->d_stage, ctx->d_hist1, ctx->d_status1, ctx->d_img1, ctx->d_sub};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int vk_ctx_sync(vk_ctx* ctx) {
    if (!ctx) return VK_EINVAL;
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_set_mapping(vk_ctx* ctx, int k, const uint32_t* pix, uint32_t npix) {
    if (!ctx || k < 5 || k > 9) return VK_EINVAL;
    const uint32_t ncode = 1u << (2 * k);
    VK_HIP(ctx, hipSetDevice(ctx->device));
    if (!pix) {
        if (npix != ncode) return VK_EINVAL;
    } else {
        for (uint32_t c = 0; c < ncode; ++c)
            if (pix[c] >= npix) return VK_EINVAL;
    }
    if (!ctx->d_pix[k]) VK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_pix[k]), ncode * sizeof(uint32_t)));
    if (pix) {
        VK_HIP(ctx, hipMemcpyAsync(ctx->d_pix[k], pix, ncode * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pix may be a temporary
    } else {
        hipLaunchKernelGGL(vk_cgr_lut_kernel, dim3((ncode + 255) / 256), dim3(256), 0, ctx->stream, k, ctx->d_pix[k]);
        VK_HIP(ctx, hipGetLastError());
    }
    ctx->npix[k] = npix;
    return VK_OK;
}

static int count_impl(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                      uint32_t nsamples, int k, uint32_t parts_per_sample, uint32_t* d_hist, uint32_t* d_status,
                      const uint64_t* seeds, const uint64_t* thresholds, uint64_t* d_sites) {
    if (!ctx || !offsets || !lengths || !d_hist || !d_status || k < 5 || k > 9) return VK_EINVAL;
    if (nsamples == 0) return VK_OK;
    if (!d_fastq) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < nsamples; ++i) {
        if ((offsets[i] & 15u) != 0) return VK_EINVAL;
        if (lengths[i] > maxlen) maxlen = lengths[i];
    }
    if ((reinterpret_cast<uintptr_t>(d_fastq) & 15u) != 0) return VK_EINVAL;
    uint32_t parts = parts_per_sample ? parts_per_sample : choose_parts(nsamples, maxlen);
    if (static_cast<uint64_t>(nsamples) * parts > (1u << 24)) return VK_EINVAL;

    if (ctx->desc_cap < 2ull * nsamples * sizeof(uint64_t)) ctx->desc_n = 0;  // realloc drops the cached copy
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_desc), &ctx->desc_cap, 2ull * nsamples * sizeof(uint64_t));
    if (rc) return rc;
    rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_wavephase), &ctx->wavephase_cap,
                static_cast<size_t>(nsamples) * parts * kWaves * sizeof(uint32_t));
    if (rc) return rc;
    uint64_t* d_offs = ctx->d_desc;
    uint64_t* d_lens = ctx->d_desc + nsamples;
    rc = upload_desc(ctx, offsets, lengths, nsamples);
    if (rc) return rc;
    const uint8_t* fq = static_cast<const uint8_t*>(d_fastq);
    SubParams sp{};
    const SubParams* sub = nullptr;
    if (seeds) {
        // seeds | thresholds travel like the descriptors (pageable source: the copy has read it on return)
        rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_sub), &ctx->sub_cap, 2ull * nsamples * sizeof(uint64_t));
        if (rc) return rc;
        VK_HIP(ctx, hipMemcpyAsync(ctx->d_sub, seeds, nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(ctx->d_sub + nsamples, thresholds, nsamples * sizeof(uint64_t),
                                   hipMemcpyHostToDevice, ctx->stream));
        if (d_sites) VK_HIP(ctx, hipMemsetAsync(d_sites, 0, 2ull * nsamples * sizeof(uint64_t), ctx->stream));
        sp.seeds = ctx->d_sub;
        sp.thresholds = ctx->d_sub + nsamples;
        sp.sites = reinterpret_cast<unsigned long long*>(d_sites);
        sub = &sp;
    }
    switch (k) {
        case 5: rc = launch_count<5>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, sub); break;
        case 6: rc = launch_count<6>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, sub); break;
        case 7: rc = launch_count<7>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, sub); break;
        case 8: rc = launch_spill<8>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, sub); break;
        default: rc = launch_spill<9>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, sub); break;
    }
    if (rc) return rc;
    hipLaunchKernelGGL(vk_check_kernel, dim3((nsamples + 255) / 256), dim3(256), 0, ctx->stream, fq, d_offs, d_lens,
                       nsamples, parts, ctx->d_wavephase, d_status);
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

int vk_count_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                    uint32_t nsamples, int k, uint32_t parts_per_sample, uint32_t* d_hist, uint32_t* d_status) {
    return count_impl(ctx, d_fastq, offsets, lengths, nsamples, k, parts_per_sample, d_hist, d_status, nullptr,
                      nullptr, nullptr);
}

int vk_count_sampled_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                            uint32_t nsamples, int k, uint32_t parts_per_sample, const uint64_t* seeds,
                            const uint64_t* thresholds, uint32_t* d_hist, uint32_t* d_status, uint64_t* d_sites) {
    if (!seeds || !thresholds) return VK_EINVAL;
    for (uint32_t i = 0; i < nsamples; ++i)
        if (thresholds[i] > (1ull << 32)) return VK_EINVAL;
    return count_impl(ctx, d_fastq, offsets, lengths, nsamples, k, parts_per_sample, d_hist, d_status, seeds,
                      thresholds, d_sites);
}

int vk_image_device(vk_ctx* ctx, const uint32_t* d_hist, uint32_t nsamples, int k, uint8_t* d_img) {
    if (!ctx || !d_hist || !d_img || k < 5 || k > 9) return VK_EINVAL;
    if (!ctx->d_pix[k]) return VK_ENOMAP;
    if (nsamples == 0) return VK_OK;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = ctx->npix[k];
    const uint32_t npad = npad_of(npix);
    const size_t work = static_cast<size_t>(nsamples) * 2u * npad * sizeof(uint32_t);
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_scratch), &ctx->scratch_cap,
                    work + static_cast<size_t>(nsamples) * sizeof(uint32_t));
    if (rc) return rc;
    const uint32_t* gate = nullptr;
    if (npad > kTile && !ctx->image_sort_only) {
        // large images: order statistics by counting; samples it cannot finish are flagged for the sort
        uint32_t* flags = ctx->d_scratch + work / sizeof(uint32_t);
        hipLaunchKernelGGL(vk_image_count_kernel, dim3(nsamples), dim3(kImgThreads), 0, ctx->stream, d_hist,
                           ctx->d_pix[k], k, npix, npad, ctx->d_scratch, d_img, flags);
        VK_HIP(ctx, hipGetLastError());
        gate = flags;
    }
    hipLaunchKernelGGL(vk_image_kernel, dim3(nsamples), dim3(kImgThreads), 0, ctx->stream, d_hist, ctx->d_pix[k], k,
                       npix, npad, ctx->d_scratch, d_img, gate);
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

int vk_fastq_to_image_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                             uint32_t nsamples, int k, uint32_t parts_per_sample, uint32_t* d_hist,
                             uint32_t* d_status, uint8_t* d_img) {
    if (!ctx || k < 5 || k > 9) return VK_EINVAL;
    if (!ctx->d_pix[k]) return VK_ENOMAP;
    int rc = vk_count_device(ctx, d_fastq, offsets, lengths, nsamples, k, parts_per_sample, d_hist, d_status);
    if (rc) return rc;
    return vk_image_device(ctx, d_hist, nsamples, k, d_img);
}

int vk_count_host(vk_ctx* ctx, const uint8_t* fastq, size_t nbytes, int k, uint32_t* hist, uint32_t* status) {
    if (!ctx || !hist || k < 5 || k > 9 || (nbytes && !fastq)) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ncode = static_cast<size_t>(1) << (2 * k);
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_stage), &ctx->stage_cap, nbytes + 64);
    if (rc) return rc;
    if (!ctx->d_hist1) VK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_hist1), (1u << 18) * sizeof(uint32_t)));
    if (!ctx->d_status1) VK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_status1), 64));
    if (nbytes)
        VK_HIP(ctx, hipMemcpyAsync(ctx->d_stage, fastq, nbytes, hipMemcpyHostToDevice, ctx->stream));
    uint64_t off = 0, len = nbytes;
    rc = vk_count_device(ctx, ctx->d_stage, &off, &len, 1, k, 0, ctx->d_hist1, ctx->d_status1);
    if (rc) return rc;
    uint32_t st = 0;
    VK_HIP(ctx, hipMemcpyAsync(hist, ctx->d_hist1, ncode * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(&st, ctx->d_status1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (status) *status = st;
    return st ? VK_EFORMAT : VK_OK;
}

int vk_image_host(vk_ctx* ctx, const uint32_t* hist, int k, uint8_t* img) {
    if (!ctx || !hist || !img || k < 5 || k > 9) return VK_EINVAL;
    if (!ctx->d_pix[k]) return VK_ENOMAP;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ncode = static_cast<size_t>(1) << (2 * k);
    if (!ctx->d_hist1) VK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_hist1), (1u << 18) * sizeof(uint32_t)));
    if (!ctx->d_img1) VK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_img1), 1u << 18));
    VK_HIP(ctx, hipMemcpyAsync(ctx->d_hist1, hist, ncode * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = vk_image_device(ctx, ctx->d_hist1, 1, k, ctx->d_img1);
    if (rc) return rc;
    VK_HIP(ctx, hipMemcpyAsync(img, ctx->d_img1, ctx->npix[k], hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_synth_fastq_device(vk_ctx* ctx, void* d_out, uint32_t sample0, uint32_t nsamples, uint32_t reads,
                          uint32_t readlen, uint64_t seed, int dist) {
    if (!ctx || !d_out || readlen == 0 || reads == 0 || reads > 10000000u || (dist != 0 && dist != 1)) return VK_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) return VK_EINVAL;
    if (nsamples == 0) return VK_OK;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t rec = 2ull * readlen + 20ull;
    const uint64_t total = rec * reads * nsamples;
    const uint64_t total16 = (total + 15) / 16;
    uint64_t blocks = (total16 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(vk_synth_kernel, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, ctx->stream,
                       static_cast<uint8_t*>(d_out), sample0, nsamples, reads, readlen, seed, dist, total16);
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

int vk_remap_host(vk_ctx* ctx, const uint8_t* img_in, uint32_t nimg, uint32_t npix_in, uint32_t npix_out,
                  const uint32_t* src0, const uint32_t* src1, const uint8_t* w0, const uint8_t* w1, int sum_rc,
                  uint8_t* img_out) {
    if (!ctx || !img_in || !img_out || !src0 || npix_in == 0 || npix_out == 0) return VK_EINVAL;
    if (sum_rc && (!src1 || !w0 || !w1)) return VK_EINVAL;
    if (nimg == 0) return VK_OK;
    for (uint32_t p = 0; p < npix_out; ++p) {
        if (src0[p] != 0xFFFFFFFFu && src0[p] >= npix_in) return VK_EINVAL;
        if (sum_rc && src1[p] != 0xFFFFFFFFu && src1[p] >= npix_in) return VK_EINVAL;
    }
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_b = static_cast<size_t>(nimg) * npix_in, out_b = static_cast<size_t>(nimg) * npix_out;
    const size_t lut_b = static_cast<size_t>(npix_out) * 4;
    const size_t need = in_b + out_b + 2 * lut_b + 2 * npix_out + 256;
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_stage), &ctx->stage_cap, need);
    if (rc) return rc;
    uint8_t* base = ctx->d_stage;
    uint32_t* d_s0 = reinterpret_cast<uint32_t*>(base);
    uint32_t* d_s1 = d_s0 + npix_out;
    uint8_t* d_w0 = reinterpret_cast<uint8_t*>(d_s1 + npix_out);
    uint8_t* d_w1 = d_w0 + npix_out;
    uint8_t* d_in = d_w1 + npix_out;
    uint8_t* d_out = d_in + in_b;
    VK_HIP(ctx, hipMemcpyAsync(d_s0, src0, lut_b, hipMemcpyHostToDevice, ctx->stream));
    if (sum_rc) {
        VK_HIP(ctx, hipMemcpyAsync(d_s1, src1, lut_b, hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(d_w0, w0, npix_out, hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(d_w1, w1, npix_out, hipMemcpyHostToDevice, ctx->stream));
    }
    VK_HIP(ctx, hipMemcpyAsync(d_in, img_in, in_b, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(vk_remap_kernel, dim3(nimg), dim3(256), 0, ctx->stream, d_in, npix_in, npix_out, d_s0, d_s1,
                       d_w0, d_w1, sum_rc, d_out);
    VK_HIP(ctx, hipGetLastError());
    VK_HIP(ctx, hipMemcpyAsync(img_out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

#ifdef VK_STAMPS
int vk_debug_read_stamps(unsigned long long* out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_vk_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

int vk_preprocess_device(vk_ctx* ctx, const uint8_t* d_img, uint32_t nimg, uint32_t side, uint32_t out,
                         const int32_t* bounds, const int32_t* coef, uint32_t kmax, float mean, float stdv,
                         float* d_out) {
    if (!ctx || !d_img || !d_out || !bounds || !coef || side == 0 || out == 0 || kmax == 0 || stdv == 0.0f)
        return VK_EINVAL;
    if (static_cast<size_t>(side) * out > 160u * 1024u - 1024u) return VK_EINVAL;  // LDS intermediate
    for (uint32_t i = 0; i < out; ++i) {
        const int32_t x0 = bounds[2 * i], n = bounds[2 * i + 1];
        if (x0 < 0 || n < 0 || static_cast<uint32_t>(n) > kmax || static_cast<uint32_t>(x0 + n) > side) return VK_EINVAL;
    }
    if (nimg == 0) return VK_OK;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t tb = static_cast<size_t>(out) * 2 * sizeof(int32_t), cb = static_cast<size_t>(out) * kmax * sizeof(int32_t);
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_stage), &ctx->stage_cap, tb + cb + 256);
    if (rc) return rc;
    int32_t* d_bounds = reinterpret_cast<int32_t*>(ctx->d_stage);
    int32_t* d_coef = d_bounds + out * 2;
    VK_HIP(ctx, hipMemcpyAsync(d_bounds, bounds, tb, hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d_coef, coef, cb, hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host tables may be temporaries
    const size_t lds = static_cast<size_t>(side) * out;
    VK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(vk_preprocess_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(vk_preprocess_kernel, dim3(nimg), dim3(256), lds, ctx->stream, d_img, side, out, d_bounds,
                       d_coef, kmax, mean, stdv, d_out);
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

int vk_last_count_launch(const vk_ctx* ctx, uint32_t* grid, uint32_t* block, uint32_t* lds_bytes) {
    if (!ctx) return VK_EINVAL;
    if (grid) *grid = ctx->last_grid;
    if (block) *block = ctx->last_block;
    if (lds_bytes) *lds_bytes = ctx->last_lds;
    return VK_OK;
}

}  // extern "C"
