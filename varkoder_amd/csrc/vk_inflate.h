// vk_inflate.h -- gzip (RFC 1952 / DEFLATE RFC 1951) inflate on the GPU: step D's real inputs are
// `<sample>@<bp>K.fq.gz` files (varKoder/commands/image.py:696-708), which dsk reads natively
// (:771-790).  Here the compressed bytes cross PCIe and are inflated in HBM, straight into the
// 16-byte aligned sample slots the count kernels read.
// Part of the one translation unit vkimg.hip (device code for gfx950).
//
// One wavefront per gzip file.  A DEFLATE stream is a chain: the position of every token is known
// only when the one before has been decoded.  The wave breaks that chain 64 bits at a time:
//   * every lane decodes ONE whole token (literal | length + distance | end of block) as if it
//     started at bit `pos + lane`: two table lookups in LDS and a few shifts, all 64 in parallel;
//   * a short scalar walk follows the true chain through those 64 answers (lane 0's token, then the
//     lane its end points to, ...) and collects the lanes on it: 6-10 tokens per step;
//   * the tokens on the chain go to an LDS ring; when it fills up the whole wave resolves them:
//     prefix sum of the lengths -> output offsets, literals stored by their lanes, matches copied
//     from the text already written (L2: loads bypass the CU's L1) -- in rounds, because a match
//     may read what an earlier token of the same batch writes; long matches are copied by all 64
//     lanes at once.
// Huffman tables: one lookup table per alphabet in LDS (11 bits for literal/length codes, 9 for
// distances); the rare longer codes are finished bit by bit from the canonical first-code arrays.
#ifndef VK_INFLATE_H
#define VK_INFLATE_H

#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int kLitRoot = 10, kDistRoot = 9;
constexpr uint32_t kRing = 256;          // token ring (resolved when fewer than 65 slots are free)

// Table entry (16 bits: the tables are what limits the wavefronts a CU can hold, and a lone wavefront
// issues an instruction every ~4 ns -- throughput comes from many of them):
//   bits 0-3 code length (0 = no such code, 15 = longer than the root: finish bit by bit),
//   bits 4-5 kind (0 literal, 1 length / distance, 2 end of block, 3 invalid),
//   bits 6-15 literal byte, or the index of the length (0..28) / distance (0..29) symbol, whose
//   base value and extra bits follow from the index by arithmetic (gz_len_of / gz_dist_of).
constexpr uint32_t kKindLit = 0, kKindLen = 1, kKindEob = 2, kKindBad = 3;
constexpr uint32_t kKindLong = 4;  // (in registers only) a code longer than a table's root: that token is decoded on the side
constexpr uint32_t kBadEntry = kKindBad << 4;

#ifndef VK_GZ_SYMTAB
#define VK_GZ_SYMTAB 1   // base value and extra bits of a length / distance symbol from two small LDS tables (96 bytes) instead of arithmetic: -12 vector instructions in gz_tokens, 78.2 -> 76.2 / 71.2 -> 69.2 ms at zlib levels 1 / 6 (profiles/ab/r06_gz_symtab.txt); 0: round 5's arithmetic
#endif
struct GzLds {
    uint16_t lit[1 << kLitRoot];
    uint16_t dst[1 << kDistRoot];
    uint32_t ring[kRing];        // literal: byte; match: 0x80000000 | dist << 9 | len
    uint16_t lit_sym[288];       // symbols ordered by (code length, symbol): canonical decoding of long codes
    uint16_t dst_sym[32];
    uint16_t lit_cnt[16], dst_cnt[16];
#if VK_GZ_SYMTAB
    // base = (m << extra) + 3 for a length symbol, (m << extra) + 1 for a distance symbol (gz_lds_init); 96 bytes: the
    // wavefront's LDS stays inside its five 1280-byte allocation granules (6272 + 96 <= 6400), see kGzChunkLds
    uint16_t len_tab[32];        // m | extra << 8
    uint8_t dist_tab[32];        // m | extra << 2
#endif
    union {
        uint8_t lens[320];       // block set-up: code lengths
        uint32_t start[64];      // resolver: output offsets of the 64 tokens in hand
    };
    union {
        uint8_t pre[128];                // block set-up: code-length code, 7-bit lookup, sym | len << 5
        unsigned long long mask[16];     // resolver: which elements of 16 x 64 output positions begin a token
    };
};

// length symbol 257 + i: base length and extra bits (RFC 1951 3.2.5), i = 0..28
__device__ __forceinline__ void gz_len_of(uint32_t i, uint32_t& base, uint32_t& extra) {
    if (i < 8u) {
        base = 3u + i;
        extra = 0u;
    } else if (i == 28u) {
        base = 258u;
        extra = 0u;
    } else {
        extra = (i >> 2) - 1u;
        base = ((4u + (i & 3u)) << extra) + 3u;
    }
}

// distance symbol s: base distance and extra bits, s = 0..29
__device__ __forceinline__ void gz_dist_of(uint32_t s, uint32_t& base, uint32_t& extra) {
    if (s < 4u) {
        base = s + 1u;
        extra = 0u;
    } else {
        extra = (s >> 1) - 1u;
        base = ((2u + (s & 1u)) << extra) + 1u;
    }
}

#if VK_GZ_SYMTAB
template <typename LdsT>
__device__ __forceinline__ void gz_lds_init(LdsT& L) {
    const uint32_t i = threadIdx.x & 63u;
    if (i < 32u) {
        uint32_t b, x;
        gz_len_of(i < 29u ? i : 0u, b, x);
        L.len_tab[i] = static_cast<uint16_t>(((b - 3u) >> x) | (x << 8));
        gz_dist_of(i < 30u ? i : 0u, b, x);
        L.dist_tab[i] = static_cast<uint8_t>(((b - 1u) >> x) | (x << 2));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
}
#endif

// entry of a literal/length symbol (0..285; 286, 287 are invalid) without its code length
__device__ __forceinline__ uint32_t gz_lit_entry(uint32_t s) {
    if (s < 256u) return (s << 6) | (kKindLit << 4);
    if (s == 256u) return kKindEob << 4;
    if (s < 286u) return ((s - 257u) << 6) | (kKindLen << 4);
    return kBadEntry;
}
__device__ __forceinline__ uint32_t gz_dist_entry(uint32_t s) { return s < 30u ? (s << 6) | (kKindLen << 4) : kBadEntry; }

__device__ __forceinline__ uint32_t gz_rev(uint32_t code, uint32_t len) { return __brev(code) >> (32u - len); }

// 64 bits of the stream from bit position p (zero beyond the end of the input)
__device__ __forceinline__ uint64_t gz_peek(const uint8_t* in, uint64_t nbytes, uint64_t p) {
    const uint64_t b = p >> 3;
    uint64_t lo = 0, hi = 0;
    if (b + 16 <= nbytes) {
        // two overlapping unaligned loads: bytes b..b+7 and b+8..b+15 (all lanes of a step share two lines)
        __builtin_memcpy(&lo, in + b, 8);
        __builtin_memcpy(&hi, in + b + 8, 8);
    } else {
        for (int i = 0; i < 8; ++i) {
            if (b + i < nbytes) lo |= static_cast<uint64_t>(in[b + i]) << (8 * i);
            if (b + 8 + i < nbytes) hi |= static_cast<uint64_t>(in[b + 8 + i]) << (8 * i);
        }
    }
    const uint32_t s = static_cast<uint32_t>(p & 7u);
    return s ? (lo >> s) | (hi << (64u - s)) : lo;
}

typedef const __attribute__((address_space(1))) uint32_t* GzGlobalU32;

__device__ __forceinline__ uint32_t gz_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t gz_uni64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ int gz_lane() { return static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))); }

// The stream seen through a register window: lane j holds dword j of 256 bytes of it, read with one
// coalesced load and good for some 1800 bits of progress; within 512 bytes of the end of the input the
// bounds-checked gz_peek is used instead.  peek(): 64 bits at a wave-uniform position, by three readlanes.
struct GzWindow {
    uint64_t in_addr;
    long long wpos = 0;  // stream bit position of the window's bit 0 (can be up to 31 below 0)
    uint32_t win = 0;
    bool have = false;
    __device__ __forceinline__ explicit GzWindow(const uint8_t* in) : in_addr(gz_uni64(reinterpret_cast<uint64_t>(in))) {}
    __device__ __forceinline__ bool covers(uint64_t pos, uint64_t nbytes) const { return (pos >> 3) + 512 <= nbytes; }
    __device__ __forceinline__ uint32_t seek(uint64_t pos, int lane) {  // -> bit offset of pos in the window
        if (!have || static_cast<long long>(pos) - wpos > 1800) {
            const uint64_t al = (in_addr + (pos >> 3)) & ~3ull;
            win = reinterpret_cast<GzGlobalU32>(al)[lane];
            wpos = static_cast<long long>(al - in_addr) * 8;
            have = true;
        }
        return static_cast<uint32_t>(static_cast<long long>(pos) - wpos);
    }
    __device__ __forceinline__ uint64_t peek(const uint8_t* in, uint64_t nbytes, uint64_t pos, int lane) {
        pos = gz_uni64(pos);
        if (!covers(pos, nbytes)) return gz_peek(in, nbytes, pos);
        const uint32_t o = gz_uni(seek(pos, lane));
        const uint32_t q = o >> 5, sh = o & 31u;
        const uint32_t d0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), static_cast<int>(q)));
        const uint32_t d1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), static_cast<int>(q + 1)));
        const uint32_t d2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(win), static_cast<int>(q + 2)));
        const uint64_t lo = ((static_cast<uint64_t>(d1) << 32) | d0) >> sh;
        const uint64_t hi = sh ? static_cast<uint64_t>(d2) << (64u - sh) : 0ull;
        return lo | hi;
    }
};

// the order in which a dynamic header lists the lengths of the code-length code: 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
// 12, 3, 13, 2, 14, 1, 15 (five bits each, in two words: no table in memory)
__device__ __forceinline__ uint32_t gz_pre_order(uint32_t i) {
    return static_cast<uint32_t>((i < 12u ? 0x22caa324e804a30ull >> (5u * i) : 0x3c2e1346cull >> (5u * (i - 12u))) & 31u);
}

// Build the lookup table of one alphabet from code lengths lens[0..n): returns false if the set is
// over-subscribed, or incomplete (allowed only for a distance alphabet with at most one code, as zlib does).
// Runs on the whole wave (lane = threadIdx.x & 63); lens, table, sym, cnt are in LDS.
template <bool LITLEN>
__device__ bool gz_build(const uint8_t* lens, uint32_t n, uint16_t* table, uint16_t* sym, uint16_t* cnt, int lane) {
    constexpr int ROOT = LITLEN ? kLitRoot : kDistRoot;
    // canonical codes: counts per length, first code per length (every lane computes the same scalars)
    uint32_t count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {  // 64 symbols at a time, a ballot per code length
        const uint32_t l = i0 + lane < n ? lens[i0 + lane] : 0u;
#pragma unroll
        for (int k = 1; k < 16; ++k) count[k] += static_cast<uint32_t>(__popcll(__ballot(l == static_cast<uint32_t>(k))));
    }
    uint32_t used = 0;
    int left = 1;
    uint32_t first[16], offs[16];
    first[0] = 0;
    offs[0] = 0;
    uint32_t code = 0, o = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = left * 2 - static_cast<int>(count[l]);
        if (left < 0) return false;  // over-subscribed
        code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
        first[l] = code;
        offs[l] = o;
        o += count[l];
        used += count[l];
    }
    if (left > 0 && (LITLEN || used > 1)) return false;  // incomplete
    for (uint32_t i = lane; i < (1u << ROOT); i += 64) table[i] = static_cast<uint16_t>(kBadEntry);
    if (lane < 16) cnt[lane] = static_cast<uint16_t>(lane ? count[lane] : 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // every symbol's rank among the symbols of its length = number of earlier symbols of that length
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        const uint32_t s = s0 + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < s0 + 64 && j < n; ++j) rank += (j < s && lens[j] == l) ? 1u : 0u;
        if (l != 0u) {
            uint32_t fl = 0, ol = 0;
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                fl = l == static_cast<uint32_t>(k) ? first[k] : fl;
                ol = l == static_cast<uint32_t>(k) ? offs[k] : ol;
            }
            sym[ol + rank] = static_cast<uint16_t>(s);
            const uint32_t c = fl + rank;
            uint32_t entry = LITLEN ? gz_lit_entry(s) : gz_dist_entry(s);
            if (l <= static_cast<uint32_t>(ROOT)) {
                entry |= l;
                for (uint32_t idx = gz_rev(c, l); idx < (1u << ROOT); idx += 1u << l) table[idx] = static_cast<uint16_t>(entry);
            } else {
                table[gz_rev(c, l) & ((1u << ROOT) - 1u)] = static_cast<uint16_t>(15u | kBadEntry);  // finish bit by bit
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// A code longer than the root, from the canonical arrays (bits arrive least significant first, codes
// are defined most significant first: one bit at a time).  Returns the symbol, or 0xFFFF; len = its bits.
__device__ __forceinline__ uint32_t gz_slow(uint64_t w, const uint16_t* cnt, const uint16_t* sym, uint32_t& len) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l < 16; ++l) {
        code |= static_cast<uint32_t>(w >> (l - 1)) & 1u;
        const uint32_t c = cnt[l];
        if (code - first < c) {
            len = l;
            return sym[index + (code - first)];
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    len = 15;
    return 0xFFFFu;
}

// One token decoded the slow way (any code length), for the rare token with a code longer than a table's
// root: kept out of line so that the token loop stays small.  Every lane computes the same.  Returns
// tok | used << 32 | kind << 40 (tok as in the ring).
__device__ __attribute__((noinline)) uint64_t gz_slow_token(const uint16_t* lit, const uint16_t* dst, const uint16_t* lit_cnt,
                                                           const uint16_t* lit_sym, const uint16_t* dst_cnt,
                                                           const uint16_t* dst_sym, uint64_t ww) {
    uint32_t e = lit[static_cast<uint32_t>(ww) & ((1u << kLitRoot) - 1u)];
    uint32_t used = e & 15u;
    if (used == 15u) {
        uint32_t l;
        const uint32_t s = gz_slow(ww, lit_cnt, lit_sym, l);
        used = l;
        e = gz_lit_entry(s);
    }
    uint32_t kind = (e >> 4) & 3u;
    uint32_t tok = e >> 6;  // literal byte
    if (kind == kKindLen) {
        uint32_t lbase, xb;
        gz_len_of(e >> 6, lbase, xb);
        const uint32_t len = lbase + (static_cast<uint32_t>(ww >> used) & ((1u << xb) - 1u));
        used += xb;
        const uint64_t w2 = ww >> used;
        uint32_t d = dst[static_cast<uint32_t>(w2) & ((1u << kDistRoot) - 1u)];
        uint32_t dl = d & 15u;
        if (dl == 15u) {
            uint32_t l;
            const uint32_t s = gz_slow(w2, dst_cnt, dst_sym, l);
            dl = l;
            d = gz_dist_entry(s);
        }
        if (((d >> 4) & 3u) != kKindLen || dl == 0u) {
            kind = kKindBad;
        } else {
            uint32_t dbase, dxb;
            gz_dist_of(d >> 6, dbase, dxb);
            const uint32_t dist = dbase + (static_cast<uint32_t>(w2 >> dl) & ((1u << dxb) - 1u));
            used += dl + dxb;
            tok = 0x80000000u | (dist << 9) | len;
        }
    }
    if (used == 0u) kind = kKindBad;
    return static_cast<uint64_t>(tok) | (static_cast<uint64_t>(used) << 32) | (static_cast<uint64_t>(kind) << 40);
}

// The header of a dynamic-codes block at bit `pos` (just behind the three block-type bits): HLIT, HDIST,
// HCLEN, the code-length code, and the run-length coded code lengths, which go to L.lens (literal/length
// codes at 0.., distance codes at 288..).  Advances pos; false if the header is not valid.
__device__ bool gz_dynamic_header(GzLds& L, const uint8_t* __restrict__ in, uint64_t nbytes, uint64_t nbits, uint64_t& pos,
                                  uint32_t& nlit, uint32_t& ndist, int lane) {
    GzWindow W(in);
    uint64_t w = W.peek(in, nbytes, pos, lane);
    nlit = (static_cast<uint32_t>(w) & 31u) + 257;
    ndist = (static_cast<uint32_t>(w >> 5) & 31u) + 1;
    const uint32_t ncode = (static_cast<uint32_t>(w >> 10) & 15u) + 4;
    pos += 14;
    if (nlit > 286 || ndist > 30) return false;
    // code-length code: ncode lengths of 3 bits in a fixed order.  Lane s < 19 takes the length of symbol s (its place in
    // that order: 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15); counts by ballot, the canonical code
    // of a symbol = first code of its length + the symbols of that length below it; the 7-bit lookup table is filled two
    // entries per lane, every lane going through the 19 symbols.  (Until round 5 every lane kept all 19 lengths in
    // registers behind select chains and lane 0 filled the table alone: ~1,500 instructions per header, and
    // vk_gzfind_kernel parses 47 false ones per chunk -- 70 % of its time.)
    w = W.peek(in, nbytes, pos, lane);
    const uint32_t sym_l = static_cast<uint32_t>(lane);
    const uint32_t place = sym_l == 0u ? 3u : (sym_l < 8u ? 19u - 2u * sym_l : (sym_l < 16u ? 2u * sym_l - 12u : sym_l - 16u));
    const uint32_t mylen = (sym_l < 19u && place < ncode) ? static_cast<uint32_t>(w >> (3u * place)) & 7u : 0u;
    pos += 3 * ncode;
    int left = 1;
    uint32_t code = 0, prev_cnt = 0, mycode = 0;
    bool ok = true;
#pragma unroll
    for (uint32_t l = 1; l < 8; ++l) {
        const uint32_t m = static_cast<uint32_t>(__ballot(mylen == l));   // symbols 0..18: bits 0..18
        const uint32_t c = static_cast<uint32_t>(__builtin_popcount(m));
        left = left * 2 - static_cast<int>(c);
        if (left < 0) ok = false;
        code = (code + prev_cnt) << 1;       // first code of length l
        prev_cnt = c;
        if (mylen == l) mycode = code + static_cast<uint32_t>(__builtin_popcount(m & ((1u << sym_l) - 1u)));
    }
    if (!ok || left > 0) return false;
    const uint32_t myrev = mylen ? gz_rev(mycode, mylen) : 0u;
    uint32_t e0 = 0, e1 = 0;   // entries `lane` and `lane + 64` of the table: sym | len << 5
#pragma unroll 1
    for (int sidx = 0; sidx < 19; ++sidx) {
        const uint32_t ls = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mylen), sidx));
        if (ls == 0u) continue;   // (wave-uniform)
        const uint32_t rs = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myrev), sidx));
        const uint32_t msk = (1u << ls) - 1u, ent = static_cast<uint32_t>(sidx) | (ls << 5);
        if ((sym_l & msk) == rs) e0 = ent;
        if (((sym_l + 64u) & msk) == rs) e1 = ent;
    }
    L.pre[lane] = static_cast<uint8_t>(e0);
    L.pre[lane + 64] = static_cast<uint8_t>(e1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the nlit + ndist code lengths, run-length coded with the code-length code
    uint32_t i = 0, prev = 0, kraft_lit = 0, kraft_dst = 0;  // (sums of 2^(15 - length))
    while (i < nlit + ndist) {
        if (pos > nbits) return false;
        w = W.peek(in, nbytes, pos, lane);
        const uint32_t e = L.pre[static_cast<uint32_t>(w) & 127u];
        const uint32_t l = e >> 5, s = e & 31u;
        if (l == 0) return false;
        pos += l;
        w >>= l;
        uint32_t rep = 1, val = s;
        if (s == 16) {
            if (i == 0) return false;
            rep = 3 + (static_cast<uint32_t>(w) & 3u);
            val = prev;
            pos += 2;
        } else if (s == 17) {
            rep = 3 + (static_cast<uint32_t>(w) & 7u);
            val = 0;
            pos += 3;
        } else if (s == 18) {
            rep = 11 + (static_cast<uint32_t>(w) & 127u);
            val = 0;
            pos += 7;
        }
        if (i + rep > nlit + ndist) return false;
        for (uint32_t r = lane; r < rep; r += 64) L.lens[(i + r < nlit ? i + r : 288 + (i + r - nlit))] = static_cast<uint8_t>(val);
        // An over-subscribed code can be told as soon as it is: what is not a header (vk_gzfind_kernel tests
        // thousands of those) fails here after a dozen lengths instead of running through all 300.
        if (val) {
            const uint32_t in_lit = i >= nlit ? 0u : (i + rep <= nlit ? rep : nlit - i);
            kraft_lit += in_lit * (32768u >> val);
            kraft_dst += (rep - in_lit) * (32768u >> val);
            if (kraft_lit > 32768u || kraft_dst > 32768u) return false;
        }
        i += rep;
        prev = val;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (an incomplete literal/length code is no header either -- gz_build<true> says the same, a table build later: most of what
    // vk_gzfind_kernel parses to the end fails here)
    if (kraft_lit != 32768u) return false;
    return L.lens[256] != 0;  // a block without an end-of-block code is not valid
}

// status bits of one gzip file
constexpr uint32_t kGzBadHeader = 1u, kGzBadData = 2u, kGzTruncated = 4u, kGzOverflow = 8u, kGzBadSize = 16u, kGzBadCrc = 32u;

struct GzJob {
    uint64_t in_off, in_len;     // compressed bytes in the input buffer
    uint64_t out_off, out_cap;   // where the text goes, and how much room there is
};

// A CHUNK of a large gzip file (the chunked path, below): one wavefront decodes the DEFLATE blocks
// from the block start `start_bit` (found by vk_gzfind_kernel; chunk 0: the gzip header at bit 0)
// up to the block start of a later chunk.
struct GzChunk {
    uint64_t in_off, in_len;     // the whole FILE's compressed bytes
    uint64_t out_off, out_cap;   // this chunk's u16 elements in the symbolic buffer (offset and room, in elements)
    uint32_t file_chunk0;        // index of the file's chunk 0 in the chunk arrays
    uint32_t nchunks;            // chunks of the file
    uint32_t chunk_bytes;        // compressed bytes per chunk (chosen per call, vk_inflate_device)
    uint32_t pad_;
};
constexpr uint64_t kGzNone = ~0ull;       // start_bit of a chunk in which no block start was found
constexpr uint64_t kGzPending = 0xFEFEFEFEFEFEFEFEull;   // ... of a chunk whose wavefront has not looked yet (the host's memset; vk_gzchunk_kernel)
constexpr uint32_t kGzSpinMax = 1u << 15;  // looks at a pending start before it is given up as "none" (~2 us apart: 60 ms -- the wavefront next door needs 13 at most)
constexpr uint32_t kGzEnd = 0xFFFFFFFFu;  // next[] of the chunk that decoded the file's last member
constexpr uint32_t kGzMemRec = 62;         // member trailers a wavefront records (end of the member's text in ITS output, CRC-32 word); a file with more in one chunk goes unchecked

#ifdef VK_GZ_STAMPS
// Diagnostic build only (tools/gz_stamps.py): per-chunk clocks of the chunk decoder, in a debug buffer that
// nothing else reads.  [c][0..1] wall clock at start / end, [2] resolve rounds, [3] tokens, [4..7] cycles in:
// gz_tokens, -, gz_resolve, block headers and table builds.
__device__ unsigned long long g_gz_stamps[16384][8];
__device__ unsigned long long g_gz_find[16384][8];  // vk_gzfind_kernel: wall ticks in all, cycles in full header tests, cycles in verification, candidates tested
__device__ unsigned long long g_gz_res[16384][4];   // gz_resolve: group set-up, round head + masks, element loop, store drain
#define GZ_T(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); x = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define GZ_ADD(acc, a, b) acc += (b) - (a)
#else
#define GZ_T(x) do { } while (0)
#define GZ_ADD(acc, a, b) do { } while (0)
#endif

// ---- the two hot pieces of the decoder, out of line -------------------------------------------------
// gz_wave below keeps the state of a whole gzip file (a dozen 64-bit uniform values and as many flags);
// inlined into it, the token loop and the resolver ran out of scalar and vector registers and spilled in
// their innermost loops.  As functions of their own they get a register allocation of their own; what is
// uniform comes back to scalar registers through readfirstlane on the way in and out.
typedef __attribute__((address_space(3))) GzLds* GzLdsP;


__device__ __attribute__((noinline)) uint64_t gz_peek_tail(const uint8_t* in, uint64_t nbytes, uint64_t p) { return gz_peek(in, nbytes, p); }

struct GzRun {
    uint64_t pos;     // bit position behind the last token taken
    uint32_t nring;   // tokens in the ring
    uint32_t code;    // why it stopped
};
constexpr uint32_t kRunRing = 0, kRunEob = 1, kRunBad = 2, kRunTrunc = 3;

// Tokens of one DEFLATE block from bit `pos` on, appended to the ring: until the end-of-block code, or
// until the ring has fewer than 65 free slots.  A step: every lane decodes the token that WOULD start at
// pos + lane (both tables, straight-line), then the true chain through the 64 answers is walked with one
// readlane per token.  The compressed stream is held in a register window -- lane j has dword j of 256
// bytes, read with one coalesced load every ~25 steps -- and a lane picks its 64 bits out of it with three
// bpermutes; within 512 bytes of the end of the input the bounds-checked peek is used instead.
__device__ __attribute__((noinline)) GzRun gz_tokens(GzLdsP L, const uint8_t* in_, uint64_t nbytes_, uint64_t pos_, uint32_t nring_) {
    const int lane = gz_lane();
    const uint64_t in_addr = gz_uni64(reinterpret_cast<uint64_t>(in_));
    const uint8_t* in = reinterpret_cast<const uint8_t*>(in_addr);
    const uint64_t nbytes = gz_uni64(nbytes_), nbits = nbytes * 8;
    uint64_t pos = gz_uni64(pos_);
    uint32_t nring = gz_uni(nring_);
    uint32_t code = kRunRing;
    long long wpos = 0;      // stream bit position of bit 0 of the window (can be up to 31 below 0)
    bool have = false;
    uint32_t win = 0;
    for (;;) {
        if (pos >= nbits) { code = kRunTrunc; break; }
        if (nring + 65 > kRing) { code = kRunRing; break; }
        uint64_t ww;
        const uint64_t b0 = pos >> 3;
        if (b0 + 512 <= nbytes) {
            if (!have || static_cast<long long>(pos) - wpos > 1800) {
                const uint64_t al = (in_addr + b0) & ~3ull;
                win = reinterpret_cast<GzGlobalU32>(al)[lane];
                wpos = static_cast<long long>(al - in_addr) * 8;
                have = true;
            }
            const uint32_t o = static_cast<uint32_t>(static_cast<long long>(pos) - wpos) + static_cast<uint32_t>(lane);
            const int q4 = static_cast<int>((o >> 5) << 2);
            const uint32_t sh = o & 31u;
            const uint32_t d0 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4, static_cast<int>(win)));
            const uint32_t d1 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4 + 4, static_cast<int>(win)));
            const uint32_t d2 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4 + 8, static_cast<int>(win)));
            const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
            ww = (static_cast<uint64_t>(hi) << 32) | lo;
        } else {
            ww = gz_peek_tail(in, nbytes, pos + lane);
        }
        const uint32_t e = L->lit[static_cast<uint32_t>(ww) & ((1u << kLitRoot) - 1u)];
        const uint32_t lused = e & 15u;
        const uint32_t lkind = (e >> 4) & 3u;
        uint32_t lbase, xb;
#if VK_GZ_SYMTAB
        {
            const uint32_t lt = L->len_tab[(e >> 6) & 31u];
            xb = lt >> 8;
            lbase = ((lt & 255u) << xb) + 3u;
        }
#else
        gz_len_of((e >> 6) & 31u, lbase, xb);
#endif
        const uint32_t len = lbase + (static_cast<uint32_t>(ww >> lused) & ((1u << xb) - 1u));
        const uint64_t w2 = ww >> (lused + xb);
        const uint32_t d = L->dst[static_cast<uint32_t>(w2) & ((1u << kDistRoot) - 1u)];
        const uint32_t dl = d & 15u;
        uint32_t dbase, dxb;
#if VK_GZ_SYMTAB
        {
            const uint32_t dt = L->dist_tab[(d >> 6) & 31u];
            dxb = dt >> 2;
            dbase = ((dt & 3u) << dxb) + 1u;
        }
#else
        gz_dist_of((d >> 6) & 31u, dbase, dxb);
#endif
        const uint32_t dist = dbase + (static_cast<uint32_t>(w2 >> dl) & ((1u << dxb) - 1u));
        const bool is_len = lkind == kKindLen;
        const bool longc = lused == 15u || (is_len && dl == 15u);
        const bool dbad = is_len && (((d >> 4) & 3u) != kKindLen || dl == 0u);
        const uint32_t used = is_len ? lused + xb + dl + dxb : lused;
        const uint32_t tok = is_len ? (0x80000000u | (dist << 9) | len) : (e >> 6);
        const uint32_t kind = longc ? kKindLong : ((dbad || used == 0u) ? kKindBad : lkind);
        const uint32_t packed = used | (kind << 8);
        // the true chain through the 64 answers: a scalar loop, one readlane per token (written out: the
        // compiler's version of it has twice the instructions, and this loop is a third of the decoder's time)
        unsigned long long chain;
        uint32_t at, halt;
        asm volatile(
            "s_mov_b64 %0, 0\n\t"
            "s_mov_b32 %1, 0\n\t"
            "s_mov_b32 %2, 0\n"
            "1:\n\t"
            "v_readlane_b32 s12, %3, %1\n\t"
            "s_cmpk_ge_u32 s12, 0x200\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_bitset1_b64 %0, %1\n\t"
            "s_and_b32 s12, s12, 0xff\n\t"
            "s_add_u32 %1, %1, s12\n\t"
            "s_cmpk_lt_u32 %1, 64\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_branch 3f\n"
            "2:\n\t"
            "s_mov_b32 %2, s12\n"
            "3:"
            : "=&s"(chain), "=&s"(at), "=&s"(halt)
            : "v"(packed)
            : "s12", "scc");
        if ((halt >> 8) == kKindBad) { code = kRunBad; break; }
        if ((chain >> lane) & 1ull) L->ring[nring + __popcll(chain & ((1ull << lane) - 1ull))] = tok;
        nring += static_cast<uint32_t>(__popcll(chain));
        pos += at;
        if ((halt >> 8) == kKindEob) {
            pos += halt & 0xFFu;
            code = pos > nbits ? kRunTrunc : kRunEob;
            break;
        }
        if ((halt >> 8) == kKindLong) {  // the token at `pos` has a code longer than a table's root
            const uint64_t r = gz_slow_token((const uint16_t*)(L->lit), (const uint16_t*)(L->dst),
                                             (const uint16_t*)(L->lit_cnt), (const uint16_t*)(L->lit_sym),
                                             (const uint16_t*)(L->dst_cnt), (const uint16_t*)(L->dst_sym),
                                             gz_peek_tail(in, nbytes, pos));
            const uint32_t k2 = gz_uni(static_cast<uint32_t>(r >> 40) & 7u);
            if (k2 == kKindBad) { code = kRunBad; break; }
            pos += gz_uni(static_cast<uint32_t>(r >> 32) & 0xFFu);
            if (k2 == kKindEob) {
                code = pos > nbits ? kRunTrunc : kRunEob;
                break;
            }
            if (lane == 0) L->ring[nring] = static_cast<uint32_t>(r);
            ++nring;
        }
        if (pos > nbits) { code = kRunTrunc; break; }  // the chain ran off the end of the file
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    GzRun out;
    out.pos = pos;
    out.nring = nring;
    out.code = code;
    return out;
}

struct GzRes {
    uint64_t opos;    // text elements written
    uint32_t st;      // status bits
    uint32_t rounds;  // (diagnostics) rounds it took
};

// The ring's tokens -> text, 64 output positions (a SLAB) at a time, in order.  Output offsets by prefix sum; a position's
// token is the rank of its bit among the token starts (sixteen 64-bit masks per 1024 positions, one LDS atomic per token);
// its element is the token's literal, or the element `dist` positions back (the match's own output taken modulo dist),
// which is found
//   * in `hist`, the last 1024 elements this wavefront produced, kept in LDS (u16) -- nearly every source at zlib's fast
//     levels, whose matches reach a few hundred bytes back;
//   * in this very slab: the lanes wait their turn (a lane's source lane is below it; the lowest waiting lane's source is
//     always final, so every pass of the loop settles at least one lane; chains are a few links long);
//   * further back: in the text in memory, behind a wait for this wavefront's outstanding stores, or -- SYM, before the
//     chunk -- as a mark for the unknown window.
// Rounds 2-4 wrote "the longest run of tokens whose sources are already written" per round, each round one wait for its
// loads and one for its stores: at zlib level 1 a round was SEVEN tokens (every match leans on the one before it) and the
// resolver 59 % of the chunk decoder's time (tools/gz_stamps.py, profiles/ab/r05_gz_resolver.txt).
// hist_from: this wavefront's output from this position on went through `hist` (a stored block goes around it).
// SYM = false: text bytes; SYM = true: u16 elements, positions before the chunk stand for the unknown window.
typedef __attribute__((address_space(3))) uint16_t* GzHistP;
#ifndef VK_GZ_HIST
#define VK_GZ_HIST 512   // (512: the block of loads for sources in memory is written for a batch of four slabs; 1024 / 8 measured slower, profiles/ab/r05_gz_resolver.txt)
#endif
constexpr uint32_t kGzHist = VK_GZ_HIST;   // elements of history in LDS
constexpr uint32_t kGzBatch = kGzHist / 128;   // slabs resolved together: half of hist
template <bool SYM>
__device__ __attribute__((noinline)) GzRes gz_resolve_slabs(GzLdsP L, GzHistP hist, void* out_, uint64_t cap_, uint64_t opos_, uint32_t nring_,
                                                       uint32_t window_open_, uint64_t member_text0_, uint64_t hist_from_) {
    const int lane = gz_lane();
    typedef __attribute__((address_space(1))) uint8_t* G8;
    typedef __attribute__((address_space(1))) uint16_t* G16;
    const uint64_t out_addr = gz_uni64(reinterpret_cast<uint64_t>(out_));
    G8 out8 = reinterpret_cast<G8>(out_addr);
    G16 out16 = reinterpret_cast<G16>(out_addr);
    const uint64_t cap = gz_uni64(cap_), member_text0 = gz_uni64(member_text0_), hist_from = gz_uni64(hist_from_);
    const uint32_t nring = gz_uni(nring_);
    const bool window_open = gz_uni(window_open_) != 0u;
    uint32_t st = 0, rounds = 0;
    auto load_raw = [&](long long p) -> uint32_t {  // element at output position p >= 0, written by this wave before
        if (SYM) return __hip_atomic_load(out16 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __hip_atomic_load(out8 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto store_elem = [&](uint64_t p, uint32_t v) {
        if (SYM) out16[p] = static_cast<uint16_t>(v);
        else out8[p] = static_cast<uint8_t>(v);
    };
    auto lds_fence = [&]() {   // LDS writes of this wave before, LDS reads of this wave behind
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the ring was written by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    uint64_t base = gz_uni64(opos_);   // output offset of ring[t0]
#ifdef VK_GZ_STAMPS
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, qa = 0, qb = 0, qc = 0, qd = 0;
#endif
    for (uint32_t t0 = 0; t0 < nring && st == 0; t0 += 64) {
        GZ_T(q0);
        const uint32_t t = t0 + lane;
        const uint32_t tok = t < nring ? L->ring[t] : 0u;
        const bool live = t < nring;
        const bool is_match = (tok >> 31) != 0u;
        const uint32_t len = !live ? 0u : (is_match ? (tok & 0x1FFu) : 1u);
        const uint32_t dist = (tok >> 9) & 0xFFFFu;
        // inclusive prefix sum of len over the wave
        const uint32_t incl = wave_inclusive_sum(len);   // (six DPP adds; six __shfl_up were six LDS round trips)
        // (a SCALAR: everything the three loops below turn on -- the chunk, the batch, the overflow check -- derives from it, and
        // with `total` out of a __shfl hipcc had to write all three as loops the lanes may leave one by one, around ballots and
        // ds_bpermute chains; tools/asm_lint.py, convergence)
        const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
        const uint64_t off = base + (incl - len);  // where this lane's token starts
        if (base + total > cap) { st |= kGzOverflow; break; }
        // a distance beyond the start of the gzip member is an error
        const uint64_t reach = window_open ? off + 32768u : off - member_text0;
        const bool bad = live && is_match && (dist == 0u || dist > reach || dist > 32768u);
        if (__any(bad)) { st |= kGzBadData; break; }
        const uint32_t start = incl - len;                  // relative to `base`
        L->start[lane] = start;
        const uint32_t base_lo = static_cast<uint32_t>(base) & (kGzHist - 1u);
        // the lowest position (relative to base, <= 0) that went through hist
        const long long hf = static_cast<long long>(hist_from) - static_cast<long long>(base);
        const int32_t hist_rel = hf < -100000ll ? -100000 : static_cast<int32_t>(hf);
        GZ_T(q1);
        GZ_ADD(qa, q0, q1);
        for (uint32_t c0 = 0; c0 < total; c0 += 1024) {
            GZ_T(q1);
            if (lane < 16) L->mask[lane] = 0ull;
            lds_fence();
            const uint32_t rel = start - c0;
            if (live && start >= c0 && rel < 1024u) __hip_atomic_fetch_or(&L->mask[rel >> 6], 1ull << (rel & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            uint32_t before = static_cast<uint32_t>(__popcll(__ballot(live && start < c0)));   // tokens that begin before the chunk
            lds_fence();
            const uint32_t cend = total - c0 < 1024u ? total : c0 + 1024u;
            GZ_T(q2);
            GZ_ADD(qb, q1, q2);
            // Up to eight slabs (a batch) at a time, in two passes.  Pass 1, per slab: the positions' tokens and sources; an
            // element that is a literal, or whose source lies before the batch and in hist, is final at once; a source before
            // hist is ASKED FOR from the text in memory (or is a mark for the unknown window); a source inside the batch is
            // left as a reference.  All the batch's loads are in flight before anything waits for one -- a wait for a load is a
            // wait for every older store of the wave (vmcnt counts in order), and pass 1 is what passes the time the last
            // batch's stores need.  Pass 2, per slab in order: references into earlier slabs from hist, chains inside the slab
            // in registers, the slab into hist and out to the text.
            for (uint32_t b0 = c0; b0 < cend; b0 += 64u * kGzBatch) {
                const uint32_t nslab = (cend - b0 < 64u * kGzBatch ? cend - b0 + 63u : 64u * kGzBatch) >> 6;
                const int32_t lo_ring = static_cast<int32_t>(b0) - static_cast<int32_t>(kGzHist);
                const int32_t ring_lo = lo_ring > hist_rel ? lo_ring : hist_rel;    // what hist holds of the positions before b0
                uint32_t val[kGzBatch], raw[kGzBatch];
                int32_t ref[kGzBatch];      // >= 0: the element of that position of this batch (relative to base); kFinal; kFar: raw[k]
                constexpr int32_t kFinal = -1, kFar = -2;
                bool any_far = false;
                // A source in memory (pass 1 below asks for it) lies before ring_lo = b0 - kGzHist or before hist_from.  The store
                // that wrote the last position before b0 - kGzHist has at least kGzHist / 64 = 8 younger stores of this wave
                // behind it (a slab's store covers 64 positions at most -- the partial slabs at group ends only add stores), and
                // everything before hist_from was waited for outright (the stored-block copy ends in vmcnt(0); so does every
                // block of far loads below).  vmcnt(7): at most the seven youngest vector-memory operations may still be in
                // flight, so that store has completed.  (Rounds 5 shipped vmcnt(15) with the same comment: by the count that
                // left the 9th..15th youngest stores -- positions before ring_lo -- possibly outstanding.  No test or soak run
                // ever read a stale byte: a wavefront's loads and stores of one address are performed in order by the memory
                // pipeline, which is what a C program's own store -> load relies on without any wait.  The count is now
                // right on its own; 64 files of 128 MB: the same time.)
                __builtin_amdgcn_s_waitcnt(0x0F77);   // vmcnt(7), expcnt(7), lgkmcnt(15)
#pragma unroll
                for (uint32_t k = 0; k < kGzBatch; ++k) {
                    val[k] = 0u;
                    raw[k] = 0u;
                    ref[k] = kFinal;
                    if (k < nslab) {
                        const uint32_t s0 = b0 + 64u * k;
                        const unsigned long long m = gz_uni64(L->mask[(s0 - c0) >> 6]);
                        const uint32_t e = s0 + static_cast<uint32_t>(lane);   // relative to `base`
                        const bool act = e < total;
                        const uint32_t own = before + static_cast<uint32_t>(__popcll(m & (~0ull >> (63 - lane)))) - 1u;  // the last token that begins at or before e
                        before += static_cast<uint32_t>(__popcll(m));
                        const uint32_t tk = L->ring[t0 + (own & 63u)], ts = L->start[own & 63u];
                        const bool mt = act && (tk >> 31) != 0u;
                        const uint32_t d = (tk >> 9) & 0xFFFFu;
                        uint32_t i = e - ts;
                        if (__any(mt && i >= d)) {   // a match that overlaps its own output: the source taken modulo the distance
                            // (i < 258 + 64, d <= 32768: a reciprocal and one correction step -- hipcc's 32-bit `%` is ~35 instructions,
                            // and runs of one byte or one short unit are in most slabs of a FASTQ's quality lines)
                            if (mt && i >= d) {
                                const uint32_t q = static_cast<uint32_t>(static_cast<float>(i) * __builtin_amdgcn_rcpf(static_cast<float>(d)));
                                int32_t r = static_cast<int32_t>(i) - static_cast<int32_t>(__umul24(q, d));
                                if (r < 0) r += static_cast<int32_t>(d);
                                else if (r >= static_cast<int32_t>(d)) r -= static_cast<int32_t>(d);
                                i = static_cast<uint32_t>(r);
                            }
                        }
                        const int32_t src = static_cast<int32_t>(ts + i) - static_cast<int32_t>(d);   // relative to base, below ts
                        uint32_t v = tk & 0xFFu;
                        const bool inb = mt && src >= static_cast<int32_t>(b0);
                        const bool ring = mt && !inb && src >= ring_lo;
                        const bool far = mt && !inb && !ring;
                        if (ring) v = hist[(base_lo + static_cast<uint32_t>(src)) & (kGzHist - 1u)];
                        const long long asrc = static_cast<long long>(base) + src;
                        if (far && SYM && asrc < 0) v = 0x8000u | static_cast<uint32_t>(asrc + 32768);
                        const bool ask = far && asrc >= 0;
                        any_far = any_far || __any(ask);
                        val[k] = ask ? static_cast<uint32_t>(src) : v;   // (a source in memory: its position, until the element is there)
                        ref[k] = inb ? src : (ask ? kFar : kFinal);
                    }
                }
                if (any_far) {
                    // (wave-uniform.  The batch's loads and ONE wait behind them, in one asm statement -- its outputs are good when
                    // it ends, whatever the compiler does with them next: hipcc waits for a relaxed atomic load, and for a plain
                    // load inside a branch, right behind it, which made a memory round trip of every one of these)
                    static_assert(kGzBatch == 4, "the asm statement below names four loads");
                    // Every address below is formed from token fields: fenced.  By construction 0 <= p < base + b0 <= cap (asrc >= 0
                    // was asked above; src < ring_lo <= b0; base + total <= cap), so the fence never fires on a stream that passed
                    // the distance check -- but an address outside the file's slot must not reach the hand-written loads, whatever
                    // state this function was entered with (round 5 lost a build -- two resolvers chosen between per call -- to a
                    // MEMORY_APERTURE_VIOLATION: a wild ADDRESS, which a store still in flight cannot produce; that build is gone,
                    // the loads that could have taken a garbled position are these).
                    uint64_t a[kGzBatch];
                    bool wild = false;
#pragma unroll
                    for (uint32_t k = 0; k < kGzBatch; ++k) {
                        long long p = ref[k] == kFar ? static_cast<long long>(base) + static_cast<int32_t>(val[k]) : 0ll;
                        if (static_cast<uint64_t>(p) >= cap) {
                            wild = wild || ref[k] == kFar;
                            p = 0;
                        }
                        a[k] = out_addr + static_cast<uint64_t>(p) * (SYM ? 2u : 1u);
                    }
                    if (__any(wild)) st |= kGzBadData;
                    if (SYM)
                        asm volatile("global_load_ushort %0, %4, off sc0\n\tglobal_load_ushort %1, %5, off sc0\n\t"
                                     "global_load_ushort %2, %6, off sc0\n\tglobal_load_ushort %3, %7, off sc0\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3])
                                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
                    else
                        asm volatile("global_load_ubyte %0, %4, off sc0\n\tglobal_load_ubyte %1, %5, off sc0\n\t"
                                     "global_load_ubyte %2, %6, off sc0\n\tglobal_load_ubyte %3, %7, off sc0\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(raw[0]), "=&v"(raw[1]), "=&v"(raw[2]), "=&v"(raw[3])
                                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
                }
#ifdef VK_GZ_STAMPS
                if (any_far) qd += 1000;   // (diagnostics: per "round" / 1000 = batches with a source in memory per slab)
#endif
#pragma unroll
                for (uint32_t k = 0; k < kGzBatch; ++k) {
                    if (k < nslab) {
                        const uint32_t s0 = b0 + 64u * k;
                        const uint32_t e = s0 + static_cast<uint32_t>(lane);
                        const bool act = e < total;
                        uint32_t v = ref[k] == kFar ? raw[k] : val[k];
                        const bool pend = ref[k] >= 0;
                        const bool earlier = pend && ref[k] < static_cast<int32_t>(s0);   // written by a slab of this batch before this one
                        const bool inslab = pend && !earlier;
                        if (earlier) v = hist[(base_lo + static_cast<uint32_t>(ref[k])) & (kGzHist - 1u)];
                        if (__any(inslab)) {
                            // chains inside the slab, in registers: P = the lane this lane's element comes from (itself when it is
                            // final); P = P[P] until nothing moves (a chain of n links: log2 n + 1 passes), then one fetch
                            uint32_t P = inslab ? static_cast<uint32_t>(ref[k]) - s0 : static_cast<uint32_t>(lane);
                            for (;;) {
                                const uint32_t PP = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(P << 2), static_cast<int>(P)));
                                const bool moved = PP != P;
                                P = PP;
                                if (!__any(moved)) break;
                            }
                            const uint32_t vv = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(P << 2), static_cast<int>(v)));
                            if (inslab) v = vv;
                        }
                        if (act) hist[(base_lo + e) & (kGzHist - 1u)] = static_cast<uint16_t>(v);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the next slab reads hist; the hardware keeps a wave's LDS operations in order)
                        if (act) store_elem(base + e, v);
                        ++rounds;
                    }
                }
            }
#ifdef VK_GZ_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            q3 = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
            qc += q3 - q2;
#endif
        }
        lds_fence();   // hist is read by the next group's slabs
        base += total;
    }
#ifdef VK_GZ_STAMPS
    if (lane == 0 && blockIdx.x < 16384) {
        g_gz_res[blockIdx.x][0] += qa; g_gz_res[blockIdx.x][1] += qb; g_gz_res[blockIdx.x][2] += qc; g_gz_res[blockIdx.x][3] += qd;
    }
#endif
    GzRes r;
    r.opos = base;
    r.st = st;
    r.rounds = rounds;
    return r;
}

// The decoder of one wavefront.  SYM = false: a whole file, text bytes straight to `out8`.
// SYM = true: one chunk of a file whose preceding 32 KiB of text are not known yet: u16 elements to
// `out16`, a value below 256 is a text byte, 0x8000 | w stands for byte w of that unknown window
// (w = 32768 + position relative to the chunk's first byte); vk_gzwin_kernel / vk_gzfinal_kernel
// replace them once the windows are known.
template <bool SYM>
__device__ void gz_wave(GzLds& L, uint16_t* hist, const uint8_t* __restrict__ in, uint64_t nbytes, uint8_t* out8, uint16_t* out16,
                        uint64_t cap, uint64_t start_bit, bool at_header, const uint64_t* starts, uint32_t my_chunk,
                        uint32_t nchunks, uint64_t& out_len, uint32_t& out_status, uint32_t& out_next,
                        uint64_t& out_endbit, uint32_t& out_isize_sum, uint32_t& out_members, uint32_t& out_crc, uint2* memrec,
                        uint32_t span_bits = 0) {   // span_bits: the compressed bits this wavefront is expected to decode (a chunk's; 0: not told)
    const int lane = threadIdx.x & 63;
    const uint64_t nbits = nbytes * 8;
#ifdef VK_GZ_STAMPS
    unsigned long long ta = 0, tb = 0, tc = 0, acc_tok = 0, acc_res = 0, acc_hdr = 0, n_rounds = 0, n_tok = 0;
    const unsigned long long wall0 = wall_clock64();
#endif
    uint64_t pos = start_bit;   // bit position in the input (wave-uniform, like everything that steers the loops)
    uint64_t opos = 0;          // text bytes (elements) written
    uint32_t st = 0;
    uint32_t nring = 0;
    uint32_t next = kGzEnd;
    uint32_t isize_sum = 0;      // sum (mod 2^32) of the ISIZE words of the member trailers passed: the host checks it against the text
    uint32_t members = 0, last_crc = 0;  // trailers passed, and the CRC-32 word of the last one (checked when the file has one member)
    uint32_t jn = my_chunk + 1;  // SYM: the next chunk whose block start has not been passed yet
    // how far back a distance may reach at output offset `off`: to the start of the gzip member, which in
    // a chunk that begins inside a member lies in the unknown window (at most 32768 before the chunk)
    bool window_open = SYM && !at_header;
    uint64_t member_text0 = 0;
    uint64_t hist_from = 0;      // the output from here on went through the resolver's history (a stored block does not)

    auto load_elem = [&](long long p) -> uint32_t {  // element at output position p (p < 0: the unknown window)
        if (SYM) {
            if (p < 0) return 0x8000u | static_cast<uint32_t>(p + 32768);
            return __hip_atomic_load(out16 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return __hip_atomic_load(out8 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto store_elem = [&](uint64_t p, uint32_t v) {
        if (SYM) out16[p] = static_cast<uint16_t>(v);
        else out8[p] = static_cast<uint8_t>(v);
    };

    // The ring's tokens -> text (gz_resolve, out of line: one copy of it, with registers of its own).
    auto resolve = [&]() {
#ifdef VK_GZ_STAMPS
        unsigned long long r0, r1;
        GZ_T(r0);
        n_tok += nring;
#endif
        const GzRes r = gz_resolve_slabs<SYM>((GzLdsP)(&L), (GzHistP)hist, SYM ? static_cast<void*>(out16) : static_cast<void*>(out8),
                                              cap, opos, nring, window_open ? 1u : 0u, member_text0, hist_from);
        opos = gz_uni64(r.opos);
        st |= gz_uni(r.st);
        nring = 0;
#ifdef VK_GZ_STAMPS
        GZ_T(r1);
        acc_res += r1 - r0;
        n_rounds += gz_uni(r.rounds);
#endif
    };

    // The block start of chunk jj of this file.  The chunks' wavefronts find their own starts (vk_gzchunk_kernel) and publish
    // them; one that has not looked yet (the first chunk of the next round of the launch, when this wavefront is the last of
    // its own) is waited for -- briefly, and not for ever: what is still pending after kGzSpinMax looks counts as "none",
    // this wavefront decodes on through that chunk, whose own output is then never linked (exact; slower).
    uint32_t known_jj = 0xFFFFFFFFu;
    uint64_t known_start = 0;
    auto start_at = [&](uint32_t jj) -> uint64_t {
        if (jj == known_jj) return known_start;
        uint64_t v = __hip_atomic_load(starts + jj, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t look = 0; gz_uni64(v) == kGzPending && look < kGzSpinMax; ++look) {
            __builtin_amdgcn_s_sleep(64);
            v = __hip_atomic_load(starts + jj, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        }
        v = gz_uni64(v);
        known_jj = jj;
        known_start = v == kGzPending ? kGzNone : v;
        return known_start;
    };

    // ---- pacing ------------------------------------------------------------------------------------
    // A SIMD issues for its OLDEST ready wavefront first.  The chunk decoder's wavefronts all start together, hold one chunk
    // each and never leave before it is done: the oldest got the issue slots and finished after 43 ms, the youngest after 66
    // (equal work: tools/gz_stamps.py), and for the last third of the launch a SIMD was left with one or two wavefronts that
    // cannot fill it on their own (the launch: 73 ms for wavefronts that live 52 on average).  So a wavefront that is BEHIND
    // asks for priority (s_setprio outranks age): by how far into its chunk it is -- under a half 3, under three quarters 2,
    // under nine tenths 1, then 0 -- set anew at every block start.  The launch: 73 -> 64 ms, the call -9 %.  (Coarse on
    // purpose.  A build that kept every wavefront within 1.6 % of the launch's mean progress -- two words in memory, a
    // returning atomic every fourth ring -- made them arrive within 6 % of one another and took 88 ms: wavefronts held in
    // step are in the resolver together and in gz_tokens together, and it is the mix of the two on a SIMD that fills it.)
    auto pace = [&]() {
#ifndef VK_GZ_NO_PRIO
        if (!SYM || span_bits == 0u) return;
        const uint64_t d = pos - start_bit;
        const uint32_t unit = (span_bits >> 7) ? (span_bits >> 7) : 1u;
        const uint32_t at = gz_uni(d >= span_bits ? 128u : static_cast<uint32_t>(d) / unit);   // 1/128ths of the chunk
        if (at < 64u) __builtin_amdgcn_s_setprio(3);
        else if (at < 96u) __builtin_amdgcn_s_setprio(2);
        else if (at < 116u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- gzip members ---------------------------------------------------------------------------
    bool any_member = false;
    bool in_member = !at_header;  // a chunk that starts at a block start is inside a member already
    bool stop = false;
    while (st == 0 && !stop) {
        if (!in_member) {
            // trailing zero padding after the last member is tolerated (as Python's gzip module does)
            if (any_member) {
                while ((pos >> 3) < nbytes && in[pos >> 3] == 0) pos += 8;
                if ((pos >> 3) >= nbytes) break;
            }
            // After a complete member, bytes that are not another gzip header end the data: zlib's gzread (how
            // dsk reads a .gz) and gzip(1) decode the members and ignore trailing garbage.
            if ((pos >> 3) + 18 > nbytes) {
                if (!any_member) st |= kGzBadHeader;
                else if ((pos >> 3) + 2 <= nbytes && in[pos >> 3] == 0x1f && in[(pos >> 3) + 1] == 0x8b) st |= kGzTruncated;
                break;
            }
            const uint8_t* h = in + (pos >> 3);
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) {
                if (!any_member || (h[0] == 0x1f && h[1] == 0x8b)) st |= kGzBadHeader;  // (a gzip magic with a method or flags no gzip has: damaged, not garbage)
                break;
            }
            const uint32_t flg = h[3];
            uint64_t p = (pos >> 3) + 10;
            if (flg & 4) {  // FEXTRA
                if (p + 2 > nbytes) { st |= kGzTruncated; break; }
                p += 2 + (static_cast<uint32_t>(in[p]) | (static_cast<uint32_t>(in[p + 1]) << 8));
            }
            if (flg & 8) {  // FNAME
                while (p < nbytes && in[p] != 0) ++p;
                ++p;
            }
            if (flg & 16) {  // FCOMMENT
                while (p < nbytes && in[p] != 0) ++p;
                ++p;
            }
            if (flg & 2) p += 2;  // FHCRC
            if (p + 8 > nbytes) { st |= kGzTruncated; break; }
            pos = p * 8;
            member_text0 = opos;
            window_open = false;  // a member starts with an empty window
        }
        any_member = true;
        in_member = false;  // (the block loop below runs the member to its end, or stops at a chunk boundary)

        // ---- DEFLATE blocks ---------------------------------------------------------------------
        bool last = false;
        while (!last && st == 0) {
            if (pos + 3 > nbits) { st |= kGzTruncated; break; }
            pace();
            GZ_T(ta);
            uint64_t w = gz_peek(in, nbytes, pos);
            last = (w & 1u) != 0u;
            const uint32_t type = static_cast<uint32_t>(w >> 1) & 3u;
            pos += 3;
            bool block_done = false;
            if (type == 0) {  // stored
                pos = (pos + 7) & ~7ull;
                const uint64_t b = pos >> 3;
                if (b + 4 > nbytes) { st |= kGzTruncated; break; }
                const uint32_t len = in[b] | (static_cast<uint32_t>(in[b + 1]) << 8);
                const uint32_t nlen = in[b + 2] | (static_cast<uint32_t>(in[b + 3]) << 8);
                if ((len ^ nlen) != 0xFFFFu) { st |= kGzBadData; break; }
                if (b + 4 + len > nbytes) { st |= kGzTruncated; break; }
                resolve();
                if (st) break;
                if (opos + len > cap) { st |= kGzOverflow; break; }
                for (uint32_t i = lane; i < len; i += 64) store_elem(opos + i, in[b + 4 + i]);
                __builtin_amdgcn_s_waitcnt(0x0070);
                opos += len;
                hist_from = opos;
                pos = (b + 4 + len) * 8;
                block_done = true;
            } else if (type == 3) {
                st |= kGzBadData;
                break;
            }
            if (!block_done) {
                uint32_t nlit, ndist;
                if (type == 1) {  // fixed codes
                    for (uint32_t i = lane; i < 288; i += 64) L.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
                    if (lane < 32) L.lens[288 + lane] = 5;
                    nlit = 288;
                    ndist = 32;  // 32 five-bit codes make the set complete; 30 and 31 never occur in valid data
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else if (!gz_dynamic_header(L, in, nbytes, nbits, pos, nlit, ndist, lane)) {
                    st |= kGzBadData;
                    break;
                }
                if (!gz_build<true>(L.lens, nlit, L.lit, L.lit_sym, L.lit_cnt, lane) ||
                    !gz_build<false>(L.lens + 288, ndist, L.dst, L.dst_sym, L.dst_cnt, lane)) {
                    st |= kGzBadData;
                    break;
                }

                GZ_T(tb);
                GZ_ADD(acc_hdr, ta, tb);
                // ---- tokens (gz_tokens: until the end of the block, or until the ring is nearly full) ----
                bool eob = false;
                while (!eob && st == 0) {
                    GZ_T(tb);
                    const GzRun r = gz_tokens((GzLdsP)(&L), in, nbytes, pos, nring);
                    pos = gz_uni64(r.pos);
                    nring = gz_uni(r.nring);
                    const uint32_t code = gz_uni(r.code);
                    GZ_T(tc);
                    GZ_ADD(acc_tok, tb, tc);
                    if (code == kRunRing) resolve();
                    else if (code == kRunEob) eob = true;
                    else st |= code == kRunBad ? kGzBadData : kGzTruncated;
                }
                if (st) break;
            }
            // a block has ended at `pos`: in a chunk, stop where a later chunk begins
            if (SYM && !last) {
                while (jn < nchunks && (start_at(jn) == kGzNone || start_at(jn) < pos)) ++jn;  // starts inside what was decoded: not block starts after all
                if (jn < nchunks && start_at(jn) == pos) {
                    next = jn;
                    stop = true;
                    break;
                }
            }
        }
        if (st || stop) break;
        resolve();
        if (st) break;
        // trailer: CRC32 (not verified here) and ISIZE
        pos = (pos + 7) & ~7ull;
        const uint64_t b = pos >> 3;
        if (b + 8 > nbytes) { st |= kGzTruncated; break; }
        const uint32_t isize = in[b + 4] | (static_cast<uint32_t>(in[b + 5]) << 8) | (static_cast<uint32_t>(in[b + 6]) << 16) |
                               (static_cast<uint32_t>(in[b + 7]) << 24);
        if (!window_open && isize != static_cast<uint32_t>(opos - member_text0)) { st |= kGzBadSize; break; }  // (a member that began in an earlier chunk is checked by the host: sum of the chunks)
        isize_sum += isize;
        last_crc = in[b] | (static_cast<uint32_t>(in[b + 1]) << 8) | (static_cast<uint32_t>(in[b + 2]) << 16) |
                   (static_cast<uint32_t>(in[b + 3]) << 24);
        if (memrec != nullptr && members < kGzMemRec && lane == 0) memrec[members] = make_uint2(static_cast<uint32_t>(opos), last_crc);
        ++members;
        pos = (b + 8) * 8;
        if (SYM) {  // the starts found inside what this chunk decoded (none, or false ones) are behind us
            while (jn < nchunks && (start_at(jn) == kGzNone || start_at(jn) < pos)) ++jn;
        }
    }
    if (st == 0 && stop) resolve();
#ifdef VK_GZ_STAMPS
    if (SYM && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* g = g_gz_stamps[blockIdx.x];
        g[0] = wall0; g[1] = wall_clock64(); g[2] = n_rounds; g[3] = n_tok;
        g[4] = acc_tok; g[5] = 0; g[6] = acc_res; g[7] = acc_hdr;
    }
#endif
    out_len = opos;
    out_status = st;
    out_next = next;
    out_endbit = pos;
    out_isize_sum = isize_sum;
    out_members = members;
    out_crc = last_crc;
}

__global__ __launch_bounds__(64) void vk_inflate_kernel(const uint8_t* __restrict__ gz, uint8_t* __restrict__ out,
                                                         const GzJob* __restrict__ jobs, uint32_t njobs,
                                                         unsigned long long* __restrict__ out_len,
                                                         uint32_t* __restrict__ status, uint32_t* __restrict__ members,
                                                         uint32_t* __restrict__ crc, uint2* __restrict__ memrec) {
    __shared__ GzLds L;
    __shared__ uint16_t hist[kGzHist];
    const uint32_t job = blockIdx.x;
    if (job >= njobs) return;
#if VK_GZ_SYMTAB
    gz_lds_init(L);
#endif
    uint64_t n = 0, endbit = 0;
    uint32_t st = 0, next = 0, isum = 0, nm = 0, cr = 0;
    gz_wave<false>(L, hist, gz + jobs[job].in_off, jobs[job].in_len, out + jobs[job].out_off, nullptr, jobs[job].out_cap, 0, true,
                   nullptr, 0, 0, n, st, next, endbit, isum, nm, cr, memrec + static_cast<size_t>(job) * kGzMemRec);
    if ((threadIdx.x & 63) == 0) {
        out_len[job] = n;
        status[job] = st;
        members[job] = nm;
        crc[job] = cr;
    }
}

// ---- the chunked path: many wavefronts per file ---------------------------------------------------
// A gzip file is ONE chain of tokens, but its DEFLATE blocks can be decoded independently once two
// things are known: where a block starts, and the 32 KiB of text before it (matches reach back that
// far).  (1) vk_gzfind_kernel looks for the first dynamic-codes block header in every chunk (128 or 256 KiB) of
// the compressed file: a cheap test of every bit offset by 64 lanes (block type, code counts, the
// code-length code must be complete), then the full header parse and table build of the decoder for
// the few that pass, and for what passes that a decode of the block to its end, which must be followed by
// another plausible block header.  (2) vk_gzchunk_kernel decodes every chunk from its start to the start of a later
// chunk, with the unknown window kept symbolic (gz_wave<true>).  A start that was not a real block
// start is never reached exactly by the chunk before it, which simply decodes on to the next one; the
// host follows the `next` links from chunk 0.  (3) vk_gzwin_kernel walks the chunks of a file in order
// and turns each chunk's last 32 KiB into the next chunk's window, (4) vk_gzfinal_kernel replaces the
// markers and writes the text.  Anything unexpected (no start found, a chunk that overflows its
// room, sizes that do not add up) sends the file through vk_inflate_kernel instead.
constexpr uint32_t kGzChunkMin = 1u << 17;                            // compressed bytes per chunk: this, or twice this
constexpr uint32_t kGzBigFile = 1u << 19;                             // files at least this large are cut into chunks
#ifndef VK_GZ_CHUNK_OCC
#define VK_GZ_CHUNK_OCC 6     // wavefronts of vk_gzchunk_kernel per SIMD the register budget is set for
#endif
// A workgroup's LDS is allocated in granules of 1280 bytes on this part (160 KiB = 128 of them), measured, not read: 192 bytes
// of padding in GzLds (6272 -> 6464 bytes: five granules -> six, 25 -> 21 wavefronts per CU where the register budget asks for
// 24) turned 78.0 ms into 108.2 -- the chunks are fitted to ONE round of the slots counted here, and the wavefronts that did not
// fit ran as a second one (profiles/ab/r06_gz_symtab.txt).  So the count is by granules, and growing GzLds past five is an error.
constexpr uint32_t kLdsGranule = 1280;
constexpr uint32_t kGzChunkLds = static_cast<uint32_t>(sizeof(GzLds)) + 2u * kGzHist;
constexpr uint32_t kGzChunkLdsAlloc = (kGzChunkLds + kLdsGranule - 1u) / kLdsGranule * kLdsGranule;
constexpr uint32_t kGzChunkWaves = (160u * 1024u / kGzChunkLdsAlloc) < 4u * VK_GZ_CHUNK_OCC ? (160u * 1024u / kGzChunkLdsAlloc) : 4u * VK_GZ_CHUNK_OCC;   // wavefronts of vk_gzchunk_kernel a CU holds (LDS, registers)
static_assert(VK_GZ_HIST != 512 || kGzChunkWaves == 4u * VK_GZ_CHUNK_OCC, "GzLds has outgrown its LDS granules: fewer wavefronts per CU than the register budget allows");

// The block start of chunk `c` (step (1) above): bit position in the file, 0 for a file's first chunk, kGzNone when the
// chunk holds none.  The whole wavefront; wave-uniform result.
__device__ __forceinline__ uint64_t gz_find_start(GzLds& L, const uint8_t* __restrict__ gz, const GzChunk& ch, uint32_t c) {
    const int lane = threadIdx.x & 63;
    const uint32_t j = c - ch.file_chunk0;
    if (j == 0) return 0;  // the file's first chunk starts at the gzip header
    const uint8_t* in = gz + ch.in_off;
    const uint64_t nbytes = ch.in_len, nbits = nbytes * 8;
    const uint64_t from = static_cast<uint64_t>(j) * ch.chunk_bytes * 8;
    uint64_t to = from + static_cast<uint64_t>(ch.chunk_bytes) * 8;
    if (to + 80 > nbits) to = nbits > 80 ? nbits - 80 : 0;  // (a header needs more bits than that anyway)
    uint64_t found = kGzNone;
#ifdef VK_GZ_STAMPS
    unsigned long long f0 = 0, f1 = 0, f2 = 0, acc_full = 0, acc_ver = 0, n_cand = 0, n_slow = 0, acc_slow = 0, n_iter = 0, max_full = 0;
    const unsigned long long fwall = wall_clock64();
#endif
    // (the stream is scanned through a register window, as in gz_tokens: lane j holds dword j of 256 bytes)
    const uint64_t in_addr = gz_uni64(reinterpret_cast<uint64_t>(in));
    long long wpos = 0;
    bool have = false;
    uint32_t win = 0;
    for (uint64_t p0 = from; p0 < to && found == kGzNone; p0 += 64) {
        const uint64_t p = p0 + lane;
        uint64_t w, w2;  // 64 bits at p, and at p + 17 (the code-length code's lengths)
        if ((p0 >> 3) + 512 <= nbytes) {
            if (!have || static_cast<long long>(p0) - wpos > 1800) {
                const uint64_t al = (in_addr + (p0 >> 3)) & ~3ull;
                win = reinterpret_cast<GzGlobalU32>(al)[lane];
                wpos = static_cast<long long>(al - in_addr) * 8;
                have = true;
            }
            const uint32_t o = static_cast<uint32_t>(static_cast<long long>(p0) - wpos) + static_cast<uint32_t>(lane);
            const int q4 = static_cast<int>((o >> 5) << 2);
            const uint32_t sh = o & 31u;
            const uint32_t d0 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4, static_cast<int>(win)));
            const uint32_t d1 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4 + 4, static_cast<int>(win)));
            const uint32_t d2 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4 + 8, static_cast<int>(win)));
            const uint32_t d3 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(q4 + 12, static_cast<int>(win)));
            w = (static_cast<uint64_t>(__builtin_amdgcn_alignbit(d2, d1, sh)) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
            const uint32_t s2 = sh + 17u;  // 17 .. 48
            const uint32_t a0 = s2 < 32u ? d0 : d1, a1 = s2 < 32u ? d1 : d2, a2 = s2 < 32u ? d2 : d3;
            w2 = (static_cast<uint64_t>(__builtin_amdgcn_alignbit(a2, a1, s2 & 31u)) << 32) | __builtin_amdgcn_alignbit(a1, a0, s2 & 31u);
        } else {
            w = gz_peek(in, nbytes, p);
            w2 = gz_peek(in, nbytes, p + 17);
        }
        // not the last block, dynamic codes, at most 286 literal/length and 30 distance codes
        bool cand = p < to && (w & 7u) == 4u && ((w >> 3) & 31u) <= 29u && ((w >> 8) & 31u) <= 29u;
        if (cand) {  // the code-length code must be complete: sum of 2^-len = 1
            const uint32_t ncode = (static_cast<uint32_t>(w >> 13) & 15u) + 4;
            uint32_t kraft = 0;
            for (uint32_t i = 0; i < ncode; ++i) {
                const uint32_t v = static_cast<uint32_t>(w2 >> (3 * i)) & 7u;
                kraft += v ? (128u >> v) : 0u;
            }
            cand = kraft == 128u;
        }
#ifdef VK_GZ_STAMPS
        ++n_iter;
#endif
        unsigned long long b = __ballot(cand);
        while (b && found == kGzNone) {  // the full test, by the whole wave, in stream order
            const int l = __builtin_ctzll(b);
            b &= b - 1;
            uint64_t pos = p0 + l + 3;
            uint32_t nlit, ndist;
            GZ_T(f0);
            const bool full = gz_dynamic_header(L, in, nbytes, nbits, pos, nlit, ndist, lane) &&
                              gz_build<true>(L.lens, nlit, L.lit, L.lit_sym, L.lit_cnt, lane) &&
                              gz_build<false>(L.lens + 288, ndist, L.dst, L.dst_sym, L.dst_cnt, lane);
            GZ_T(f1);
            GZ_ADD(acc_full, f0, f1);
#ifdef VK_GZ_STAMPS
            ++n_cand;
            if (f1 - f0 > 30000) { ++n_slow; acc_slow += f1 - f0; }
            if (f1 - f0 > max_full) max_full = f1 - f0;
#endif
            if (!full) continue;
            // Random bits pass all that about once in 10^8 positions, i.e. in one chunk of a hundred, and a false
            // start costs the chunk before it a second chunk's worth of decoding while every other wavefront
            // has finished.  So: decode the candidate block to its end (tokens dropped) and ask for a
            // plausible block header behind it.  A block too long to check (2 chunks) is taken on trust.
            bool ok = true, at_end = false;
            const uint64_t limit = pos + 16ull * ch.chunk_bytes;
            for (;;) {
                const GzRun r = gz_tokens((GzLdsP)(&L), in, nbytes, pos, 0);
                pos = gz_uni64(r.pos);
                const uint32_t code = gz_uni(r.code);
                if (code == kRunEob) {
                    at_end = true;
                    break;
                }
                if (code != kRunRing) ok = false;
                if (!ok || pos > limit) break;
            }
            if (ok && at_end) {
                ok = pos + 3 <= nbits;
                if (ok) {
                    const uint64_t wn = gz_peek(in, nbytes, pos);
                    const uint32_t type = static_cast<uint32_t>(wn >> 1) & 3u;
                    pos += 3;
                    if (type == 3) {
                        ok = false;
                    } else if (type == 0) {
                        const uint64_t bb = (pos + 7) >> 3;
                        ok = bb + 4 <= nbytes &&
                             ((in[bb] | (static_cast<uint32_t>(in[bb + 1]) << 8)) ^ (in[bb + 2] | (static_cast<uint32_t>(in[bb + 3]) << 8))) == 0xFFFFu;
                    } else if (type == 2) {
                        ok = gz_dynamic_header(L, in, nbytes, nbits, pos, nlit, ndist, lane) &&
                             gz_build<true>(L.lens, nlit, L.lit, L.lit_sym, L.lit_cnt, lane) &&
                             gz_build<false>(L.lens + 288, ndist, L.dst, L.dst_sym, L.dst_cnt, lane);
                    }
                }
            }
            GZ_T(f2);
            GZ_ADD(acc_ver, f1, f2);
            if (ok) found = p0 + l;
        }
    }
#ifdef VK_GZ_STAMPS
    if (lane == 0 && c < 16384) {
        g_gz_find[c][0] = wall_clock64() - fwall; g_gz_find[c][1] = acc_full; g_gz_find[c][2] = acc_ver; g_gz_find[c][3] = n_cand;
        g_gz_find[c][4] = n_slow; g_gz_find[c][5] = acc_slow; g_gz_find[c][6] = max_full; g_gz_find[c][7] = n_iter;
    }
#endif
    return found;
}

// (1) as a launch of its own (VKIMG_GZ_SPLIT_FIND=1; tests, A/B timing): vk_gzchunk_kernel finds its chunks' starts itself.
__global__ __launch_bounds__(64, 7) void vk_gzfind_kernel(const uint8_t* __restrict__ gz, const GzChunk* __restrict__ chunks,
                                                        uint32_t nchunks_total, uint64_t* __restrict__ starts) {
    __shared__ GzLds L;
    const uint32_t c = blockIdx.x;
    if (c >= nchunks_total) return;
#if VK_GZ_SYMTAB
    gz_lds_init(L);
#endif
    const GzChunk ch = chunks[c];
    const uint64_t found = gz_find_start(L, gz, ch, c);
    if ((threadIdx.x & 63) == 0) starts[c] = found;
}

__global__ __launch_bounds__(64, VK_GZ_CHUNK_OCC) void vk_gzchunk_kernel(const uint8_t* __restrict__ gz, uint16_t* __restrict__ sym,
                                                         const GzChunk* __restrict__ chunks, uint32_t nchunks_total,
                                                         uint64_t* starts,   // (fused: every entry kGzPending at launch, written by its chunk's wavefront, read by others)
                                                         unsigned long long* __restrict__ out_len, uint32_t* __restrict__ status,
                                                         uint32_t* __restrict__ next, uint32_t* __restrict__ isize_sum,
                                                         uint32_t* __restrict__ members, uint32_t* __restrict__ crc,
                                                         uint2* __restrict__ memrec, uint32_t fused) {
    __shared__ GzLds L;
    __shared__ uint16_t hist[kGzHist];
    const uint32_t c = blockIdx.x;
    if (c >= nchunks_total) return;
#if VK_GZ_SYMTAB
    gz_lds_init(L);
#endif
    const GzChunk ch = chunks[c];
    const uint32_t j = c - ch.file_chunk0;
    uint64_t n = 0, eb = 0;
    uint32_t st = 0, nx = kGzEnd, isum = 0, nm = 0, cr = 0;
    uint64_t s0;
    if (fused) {
        // Step (1) here, not in a launch of its own: that launch lasted as long as its slowest chunk (12.7 ms for a mean of
        // 6.5 per chunk); a wavefront that has its start decodes at once.  At the top priority: the decoders next to it on
        // the SIMD ask for theirs by their progress (gz_wave), and another chunk's wavefront may be waiting for this start.
        __builtin_amdgcn_s_setprio(3);
        s0 = gz_uni64(gz_find_start(L, gz, ch, c));
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(starts + c, s0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (L is handed from the finder to the decoder)
        __builtin_amdgcn_wave_barrier();
    } else {
        s0 = starts[c];
    }
    if (s0 == kGzNone) {
        st = 0x80000000u;  // no block start in this chunk: the chunk before decodes through it
    } else {
        gz_wave<true>(L, hist, gz + ch.in_off, ch.in_len, nullptr, sym + ch.out_off, ch.out_cap, s0, j == 0,
                      starts + ch.file_chunk0, j, ch.nchunks, n, st, nx, eb, isum, nm, cr, memrec + static_cast<size_t>(c) * kGzMemRec,
                      ch.chunk_bytes * 8u);
    }
    if ((threadIdx.x & 63) == 0) {
        out_len[c] = n;
        status[c] = st;
        next[c] = nx == kGzEnd ? kGzEnd : ch.file_chunk0 + nx;
        isize_sum[c] = isum;
        members[c] = nm;
        crc[c] = cr;
    }
}

// One workgroup per file: chunk after chunk of its chain, the 32 KiB window a chunk sees (win_in[k]) and the
// one it leaves behind.  items: {chunk's element offset, its length}; first/count: the file's slice of them.
struct GzItem {
    uint64_t sym_off, len, text_off;
};

__global__ __launch_bounds__(1024) void vk_gzwin_kernel(const uint16_t* __restrict__ sym, const GzItem* __restrict__ items,
                                                         const uint32_t* __restrict__ first, const uint32_t* __restrict__ count,
                                                         uint8_t* __restrict__ win_in) {
    __shared__ uint8_t win[2][32768];
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    for (uint32_t i = tid; i < 32768; i += 1024) win[0][i] = 0;
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t k = first[f]; k < first[f] + count[f]; ++k) {
        const uint8_t* old = win[cur];
        uint8_t* nw = win[cur ^ 1];
        uint8_t* dst = win_in + static_cast<uint64_t>(k) * 32768;
        for (uint32_t i = tid; i < 32768 / 16; i += 1024)
            reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(old)[i];
        const uint64_t n = items[k].len;
        const uint16_t* e = sym + items[k].sym_off;
        const uint32_t keep = n >= 32768 ? 0u : 32768u - static_cast<uint32_t>(n);  // bytes of the old window that stay
        for (uint32_t i = tid; i < 32768; i += 1024) {
            uint32_t v;
            if (i < keep) {
                v = old[i + (32768 - keep)];
            } else {
                const uint32_t x = e[n - (32768 - i)];
                v = x < 0x8000u ? x : old[x & 0x7FFFu];
            }
            nw[i] = static_cast<uint8_t>(v);
        }
        __syncthreads();
        cur ^= 1;
    }
}

__global__ __launch_bounds__(256) void vk_gzfinal_kernel(const uint16_t* __restrict__ sym, const GzItem* __restrict__ items,
                                                          uint32_t nitems, const uint8_t* __restrict__ win_in,
                                                          uint8_t* __restrict__ text) {
    const uint32_t k = blockIdx.y;
    if (k >= nitems) return;
    const GzItem it = items[k];
    const uint16_t* e = sym + it.sym_off;
    const uint8_t* w = win_in + static_cast<uint64_t>(k) * 32768;
    uint8_t* out = text + it.text_off;
    auto one = [&](uint64_t i) {
        const uint32_t x = e[i];
        out[i] = static_cast<uint8_t>(x < 0x8000u ? x : w[x & 0x7FFFu]);
    };
    // eight elements per thread from where the ELEMENTS are 16-byte aligned (16 bytes in, 8 bytes out); the few before and
    // after that by the first threads of block 0.  (The 16-byte loads want their alignment more than the 8-byte stores
    // theirs: with the text aligned instead the kernel took 8.4 ms for 64 files of 128 MB, so 7.6; sixteen elements per
    // thread, non-temporal accesses, a contiguous span per workgroup: nothing.)
    const uint64_t head = ((16u - (reinterpret_cast<uint64_t>(e) & 15u)) & 15u) >> 1;
    const uint64_t h = head < it.len ? head : it.len;
    const uint64_t groups = (it.len - h) / 8, tail0 = h + groups * 8;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) one(threadIdx.x);
        if (tail0 + threadIdx.x < it.len && threadIdx.x < 8) one(tail0 + threadIdx.x);
    }
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; g < groups; g += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t i = h + g * 8;
        uint4 v;
        __builtin_memcpy(&v, e + i, 16);
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
        uint32_t b[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b[2 * q] = x[q] & 0xFFFFu;
            b[2 * q + 1] = x[q] >> 16;
        }
        if ((v.x | v.y | v.z | v.w) & 0x80008000u) {  // a reference into the window among them
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (b[q] & 0x8000u) b[q] = w[b[q] & 0x7FFFu];
        }
        uint2 o;
        o.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        o.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        __builtin_memcpy(out + i, &o, 8);   // (wherever it falls)
    }
}

// ---- CRC-32 of the inflated text ------------------------------------------------------------------
// gzip's check word is the CRC-32 of the member's text.  It is linear over GF(2) apart from its preset and
// final inversion, so the text is cut into 64 KiB segments COUNTED FROM ITS END (leading zeros do not change
// a zero-preset remainder, so the short first segment needs no special case).  A wavefront takes a segment
// in 16 rows of 4 KiB: a row is read with full lines per load (lane = 16 consecutive bytes), turned in LDS so
// that lane l owns bytes [64 l, 64 l + 64) of the row, and pushed through the four 256-entry tables of
// "slicing by 4"; between rows the lane's remainder is advanced over the 4032 bytes that are not its own
// (operator gz_ops[33]).  The 64 lanes are merged by a tree of "shift by 64 x 2^l bytes" operators (32 x 32
// bit matrices, gz_ops[n] = the operator for 2^n zero bytes, built on the host as zlib's crc32_combine
// does), one wave per file folds the segment values, and the host adds the preset's contribution (shift by
// the text length) and the inversion.
struct GzCrcJob {
    uint64_t text_off, text_len;
    uint32_t seg0, nseg;  // this file's slice of the segment array
};

__device__ __forceinline__ uint32_t gz_apply(const uint32_t* op, uint32_t v) {  // op x v over GF(2)
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) r ^= ((v >> i) & 1u) ? op[i] : 0u;
    return r;
}

__global__ __launch_bounds__(256) void vk_crc32_seg_kernel(const uint8_t* __restrict__ text, const GzCrcJob* __restrict__ jobs,
                                                            uint32_t njobs, const uint32_t* __restrict__ seg_job,
                                                            uint32_t nseg_total, const uint32_t* __restrict__ ops,
                                                            uint32_t* __restrict__ seg_crc) {
    __shared__ uint32_t T[4][256];  // T[n][b]: byte b followed by n zero bytes (the tables of "slicing by 4")
    __shared__ uint32_t M[7][32];   // operators for 64, 128, ..., 2048 bytes, and for 4032
    __shared__ uint4 rows[4][256];  // one 4 KiB row per wavefront
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        T[0][threadIdx.x] = c;
        if (threadIdx.x < 192) M[threadIdx.x >> 5][threadIdx.x & 31] = ops[(6 + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)];
        else if (threadIdx.x < 224) M[6][threadIdx.x & 31] = ops[33 * 32 + (threadIdx.x & 31)];
        __syncthreads();
        for (int n = 1; n < 4; ++n) {
            c = T[0][c & 0xFFu] ^ (c >> 8);
            T[n][threadIdx.x] = c;
        }
    }
    __syncthreads();
    const uint32_t seg = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per segment
    if (seg >= nseg_total) return;
    const int lane = threadIdx.x & 63;
    uint4* row = rows[threadIdx.x >> 6];
    const GzCrcJob jb = jobs[seg_job[seg]];
    const uint32_t s = seg - jb.seg0;
    // segment s of nseg covers text positions [end - (nseg - s) * 64K, ...): may start before the text
    const long long seg_first = static_cast<long long>(jb.text_len) - static_cast<long long>(jb.nseg - s) * 65536ll;
    const uint8_t* t = text + jb.text_off;
    uint32_t c = 0;
    for (int r = 0; r < 16; ++r) {
        if (seg_first + 4096ll * (r + 1) <= 0) continue;  // a row that lies before the text (wave-uniform)
        c = gz_apply(M[6], c);                            // skip the other lanes' 4032 bytes (nothing to skip while c is 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long p = seg_first + 4096ll * r + 1024 * i + 16 * lane;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p >= 0) {
                __builtin_memcpy(&v, t + p, 16);
            } else if (p > -16) {  // the text begins inside these sixteen bytes
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (p + q >= 0) w[q >> 2] |= static_cast<uint32_t>(t[p + q]) << (8 * (q & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            row[i * 64 + lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = row[lane * 4 + i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                c ^= w[d];
                c = T[3][c & 0xFFu] ^ T[2][(c >> 8) & 0xFFu] ^ T[1][(c >> 16) & 0xFFu] ^ T[0][c >> 24];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the row is rewritten by other lanes next
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        const uint32_t other = __shfl_xor(c, 1 << l);
        if (lane & (1 << l)) c = gz_apply(M[l], other) ^ c;  // the left neighbour's group, shifted over this group
    }
    if (lane == 63) seg_crc[seg] = c;
}

__global__ __launch_bounds__(64) void vk_crc32_fold_kernel(const GzCrcJob* __restrict__ jobs, uint32_t njobs,
                                                            const uint32_t* __restrict__ ops, const uint32_t* __restrict__ seg_crc,
                                                            uint32_t* __restrict__ raw) {
    __shared__ uint32_t M[7][32];  // operators for 64 KiB x 1, 2, ..., 64
    const uint32_t j = blockIdx.x;
    if (j >= njobs) return;
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 7 * 32; i += 64) M[i >> 5][i & 31] = ops[(16 + (i >> 5)) * 32 + (i & 31)];
    __syncthreads();
    const GzCrcJob jb = jobs[j];
    // 64 segments per step, one per lane, merged by the same tree as inside a segment; the file's segments
    // are preceded by as many empty ones (remainder 0) as it takes to fill the first step
    const uint32_t nsteps = (jb.nseg + 63) / 64, pad = nsteps * 64 - jb.nseg;
    uint32_t c = 0;
    for (uint32_t b = 0; b < nsteps; ++b) {
        const uint32_t idx = b * 64 + lane;
        uint32_t v = idx >= pad ? seg_crc[jb.seg0 + (idx - pad)] : 0u;
#pragma unroll
        for (int l = 0; l < 6; ++l) {
            const uint32_t other = __shfl_xor(v, 1 << l);
            if (lane & (1 << l)) v = gz_apply(M[l], other) ^ v;
        }
        c = gz_apply(M[6], c) ^ __shfl(v, 63);
    }
    if (lane == 0) raw[j] = c;
}

}  // namespace

#endif  // VK_INFLATE_H
