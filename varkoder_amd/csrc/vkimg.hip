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
// The kernels live in vk_count.h (K1), vk_image.h (K2) and vk_aux.h, included below.
// Design notes live in DESIGN.md; the short version for K1:
//   * one 1024-thread workgroup per (sample, byte-range part);
//     its 16 wavefronts run WITHOUT workgroup barriers in steady state: every
//     wave streams its own contiguous byte range in 4 KiB pieces (every lane loads its
//     own 64 contiguous bytes, four 16-B loads, prefetched one piece ahead)
//     and classifies them with 32-bit SWAR arithmetic (vk_lane.h);
//   * FASTQ line phase (header/sequence/plus/quality) comes from a wave-level
//     prefix sum of newline counts; the phase at a range start is recovered
//     locally from the '@' / '+' framing, so byte ranges are independent;
//   * k-mer windows are counted forward-strand only into an LDS histogram
//     (ds_add_u32); the strand merge happens once per sample in K2;
//   * 4^k u32 > LDS for k = 8, 9: QUADS of neighbouring windows are bucketed through workgroup-shared LDS
//     queues into 256 streams per sample in HBM (half a byte per window) and replayed into LDS tables
//     (vk_bucket_kernel<K, 3>, vk_quad_count_kernel, vk_quad_merge_kernel); subsampled launches: pairs of
//     windows through wave-private queues into 16 streams (vk_bucket_kernel<K, 1>, vk_bucket_count_kernel).
//   vk_remap_kernel / vk_preprocess_kernel: `convert`'s remap and the input side of `query`.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <utility>
#include <vector>

#include "vkimg.h"
#include "vk_lane.h"

#include "vk_count.h"
#include "vk_pack.h"
#include "vk_ladder.h"
#include "vk_image.h"
#include "vk_inflate.h"
#include "vk_aux.h"

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
    uint64_t* h_desc = nullptr;   // pinned mirror of d_desc (descriptor cache): the copy uploaded last
    size_t h_desc_cap = 0;
    uint64_t* h_desc_alt = nullptr;  // the other of the two mirrors: a batch with new descriptors is staged here
    size_t h_desc_alt_cap = 0;       // while the previous batch's copy may still be reading h_desc
    hipEvent_t desc_ev = nullptr, desc_ev_alt = nullptr;  // recorded behind the copy out of each mirror
    uint32_t desc_n = 0;
    uint32_t* d_wavephase = nullptr;
    size_t wavephase_cap = 0;
    // read index of a batch (vk_read_index_device): anchors | sample bases | segment counts | sites | overflow flags
    uint8_t* d_index = nullptr;
    size_t index_cap = 0;
    IndexParams ix{};
    const void* ix_fastq = nullptr;
    uint32_t ix_parts = 0;
    std::map<std::pair<uint64_t, uint64_t>, uint32_t> ix_sample;   // (offset, length) -> sample of the index
    std::vector<uint32_t> ix_status, ix_overflow;                  // per indexed sample (host copies)
    uint32_t* d_walk = nullptr;   // per-pair arrays of a walker launch
    size_t walk_cap = 0;
    uint32_t* d_aside = nullptr;  // k <= 7: the waves' lists of lanes set aside
    size_t aside_cap = 0;
    uint32_t* d_scratch = nullptr;
    size_t scratch_cap = 0;
    uint32_t* d_spill = nullptr;  // k >= 8: bucket cursors + bucket streams
    size_t spill_cap = 0;
    size_t spill_budget = 96ull << 30;  // bytes of HBM the spill path may use at a time
    // host-call staging
    uint64_t* d_sub = nullptr;    // subsampling launches: seeds | thresholds
    uint8_t* d_gzjobs = nullptr;  // vk_inflate_device: jobs | text lengths | status words
    size_t gzjobs_cap = 0;
    uint8_t* d_gzmeta = nullptr;  // ... chunked path: chunk table and results, then the chain items
    size_t gzmeta_cap = 0;
    uint8_t* d_gzsym = nullptr;   // ... u16 elements of every chunk
    size_t gzsym_cap = 0;
    uint8_t* d_gzwin = nullptr;   // ... the 32 KiB window every chunk of a chain starts with
    size_t gzwin_cap = 0;
    uint8_t* d_gzcrc = nullptr;   // ... CRC-32 jobs, operators, segment values
    size_t gzcrc_cap = 0;
    bool gz_no_chunks = false;    // VKIMG_GZ_NO_CHUNKS=1: every file through the one-wavefront kernel (tests, A/B timing)
    uint32_t gz_chunk_bytes = 0;  // VKIMG_GZ_CHUNK_BYTES: fixed chunk size of the chunked inflate (0 = fitted to the device)
    bool gz_split_find = false;   // VKIMG_GZ_SPLIT_FIND=1: the chunks' block starts by a launch of its own (vk_gzfind_kernel; rounds 2-5) instead of by the chunk decoder's wavefronts themselves (tests, A/B timing)
    uint32_t gz_fill_pct = 0;     // VKIMG_GZ_FILL: per cent of the device's chunk-wavefront slots a round of chunks is fitted to (0 = 99; A/B timing)
    uint32_t gz_lds_pad = 0;      // VKIMG_GZ_LDS_PAD: bytes of LDS a chunk wavefront asks for on top of its own (A/B timing: fewer wavefronts per CU)
    int num_cus = 256;
    size_t sub_cap = 0;
    uint8_t* d_stage = nullptr;
    size_t stage_cap = 0;
    uint32_t* d_synth = nullptr;       // vk_synth_shaped_*: record offsets of a slab of samples
    size_t synth_cap = 0;
    uint64_t* d_synth_offs = nullptr;  // ... and the slab's sample offsets
    size_t synth_offs_cap = 0;
    uint32_t* d_hist1 = nullptr;
    uint32_t* d_status1 = nullptr;
    size_t status1_cap = 0;
    uint8_t* d_img1 = nullptr;
    uint32_t last_grid = 0, last_block = 0, last_lds = 0;
    uint64_t last_waves = 0, last_bytes = 0;   // of the last count call: wave slots in d_wavephase, FASTQ bytes
    bool image_sort_only = false;  // VKIMG_IMAGE_SORT_ONLY=1: always take the sort kernel (tests, A/B timing)
    uint32_t spill_misc_cap = 0;   // VKIMG_SPILL_MISC_CAP=n: log2 of the entries per (workgroup, bucket) region of the quad route's listed quads (tests)
    uint32_t spill_runs_cap = 0;   // VKIMG_SPILL_RUNS_CAP=n: runs per sample arena of the k = 8, 9 path (tests)
    bool spill_packed = false;     // VKIMG_SPILL_PACKED=1: k = 8, 9 pass A in two kernels, vk_pack_kernel + the partition of the packed stream (measured slower than the one kernel that classifies every byte: 21.1 against 16.3 ms per 100 samples; tests, A/B timing)
    bool spill_pairs = false;      // VKIMG_SPILL_PAIRS=1: a plain k = 8, 9 count through the pair route of rounds 1-4 (u16 per two windows, wave-private queues; what subsampled and packed launches still use) instead of the quad route (tests, A/B timing)
    bool spill_force_wide = false; // VKIMG_SPILL_FORCE_WIDE=1: every k = 8, 9 replay job through the u32 window counters (tests)
    bool k1_classic = false;       // VKIMG_K1_CLASSIC=1: k <= 7 through vk_count_kernel (every byte through the heavy stage) instead of vk_count_dense_kernel (tests, A/B timing)
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
    // Two pinned mirrors take turns: the new descriptors go into the one that was NOT uploaded last, and
    // the host waits only for the copy that read that mirror two batches ago (an event behind it) -- not,
    // as a stream synchronise would, for every kernel queued since (a pipeline's batches all differ).
    uint64_t* m = ctx->h_desc_alt;
    size_t cap = ctx->h_desc_alt_cap;
    hipEvent_t ev = ctx->desc_ev_alt;
    if (ev) VK_HIP(ctx, hipEventSynchronize(ev));
    if (cap < 2 * bytes) {
        if (m) VK_HIP(ctx, hipHostFree(m));
        m = nullptr;
        cap = 0;
        ctx->h_desc_alt = nullptr;
        ctx->h_desc_alt_cap = 0;
        VK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&m), 2 * bytes + 4096, hipHostMallocDefault));
        cap = 2 * bytes + 4096;
    }
    if (!ev) VK_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    memcpy(m, offsets, bytes);
    memcpy(m + n, lengths, bytes);
    VK_HIP(ctx, hipMemcpyAsync(ctx->d_desc, m, 2 * bytes, hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipEventRecord(ev, ctx->stream));
    // swap: what was uploaded just now becomes the cache the next call compares against
    ctx->h_desc_alt = ctx->h_desc;
    ctx->h_desc_alt_cap = ctx->h_desc_cap;
    ctx->desc_ev_alt = ctx->desc_ev;
    ctx->h_desc = m;
    ctx->h_desc_cap = cap;
    ctx->desc_ev = ev;
    ctx->desc_n = n;
    return VK_OK;
}

uint32_t npad_of(uint32_t npix) {
    uint32_t p = 1;
    while (p < npix) p <<= 1;
    return p;
}

template <int K>
int launch_count(vk_ctx* ctx, const uint8_t* d_fastq, const uint64_t* d_offs, const uint64_t* d_lens,
                 uint32_t nsamples, uint32_t parts, uint64_t maxlen, uint32_t* d_hist, const SubParams* sub,
                 const IndexParams* index = nullptr) {
    const uint32_t grid = nsamples * parts;
    const int atomic_flush = parts > 1 ? 1 : 0;
    if (atomic_flush)
        VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * (1u << (2 * K)) * sizeof(uint32_t),
                                   ctx->stream));
    ctx->last_grid = grid;
    ctx->last_block = kCountThreads;
    ctx->last_lds = (1u << (2 * K)) * 4u + kWaves * 64 + 2 * 66 * 16;
    if (sub)
        hipLaunchKernelGGL((vk_count_kernel<K, true>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                           d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, *sub);
    else if (ctx->k1_classic && !index)
        hipLaunchKernelGGL((vk_count_kernel<K, false>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                           d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, SubParams{});
    else {
        ctx->last_lds = (1u << (2 * K)) * 4u + kWaves * 1024;
        // the waves' lists of lanes set aside (vk_count.h): room for one lane per piece of the longest range (fastp-shaped
        // reads need 0.45), at least 64 entries; a wave whose list is full sends its pieces down the general path instead
        const uint64_t wave_bytes = maxlen / (static_cast<uint64_t>(parts) * kWaves) + 64;
        uint64_t cap = wave_bytes / kPiece + 64;
        if (cap > (1u << 20)) cap = 1u << 20;
        const size_t nwaves = static_cast<size_t>(grid) * kWaves;
        const size_t need = nwaves * (cap + 1) * sizeof(uint32_t);   // lists, then their lengths
        int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_aside), &ctx->aside_cap, need);
        if (rc) return rc;
        uint32_t* const d_aside_n = ctx->d_aside + nwaves * cap;
        // vk_aside_kernel: about a thousand workgroups, each serving `upb` workgroups of the count launch (of one sample)
        const uint32_t awant = nsamples >= 1024u ? 1u : (1024u + nsamples - 1u) / nsamples;
        const uint32_t upb = (parts + (awant < parts ? awant : parts) - 1u) / (awant < parts ? awant : parts);
        const uint32_t ablocks = (parts + upb - 1u) / upb;
        if (index) {   // the count and the read index of the samples in one pass (vk_count_index_device)
            hipLaunchKernelGGL((vk_count_dense_kernel<K, true>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                               d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, ctx->d_aside,
                               static_cast<uint32_t>(cap), d_aside_n, *index);
            VK_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL((vk_aside_kernel<K, true>), dim3(nsamples * ablocks), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                               d_offs, d_lens, nsamples, parts, d_hist, ctx->d_aside, static_cast<uint32_t>(cap), d_aside_n, *index, upb);
        } else {
            hipLaunchKernelGGL((vk_count_dense_kernel<K, false>), dim3(grid), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                               d_offs, d_lens, nsamples, parts, d_hist, ctx->d_wavephase, atomic_flush, ctx->d_aside,
                               static_cast<uint32_t>(cap), d_aside_n, IndexParams{});
            VK_HIP(ctx, hipGetLastError());
#ifndef VK_DIAG_ASIDE_INLINE   // (diagnostic: the count kernel counts its own lists, vk_count.h)
            hipLaunchKernelGGL((vk_aside_kernel<K, false>), dim3(nsamples * ablocks), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                               d_offs, d_lens, nsamples, parts, d_hist, ctx->d_aside, static_cast<uint32_t>(cap), d_aside_n, IndexParams{}, upb);
#endif
        }
    }
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

// k = 8, 9, the quad route (vk_count.h: vk_bucket_kernel<K, 3>, vk_quad_list / _count / _merge_kernel),
// in sub-batches that fit the spill budget.
template <int K>
int launch_spill_quad(vk_ctx* ctx, const uint8_t* d_fastq, const uint64_t* d_offs, const uint64_t* d_lens,
                      uint32_t nsamples, uint32_t parts, uint64_t maxlen, uint32_t* d_hist) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    constexpr size_t kOutWords = 4u << (2 * K - 8);   // what a pass B job stores: four window arrays of 4^(K-4) counters
    // Arena of one sample, in 4 KiB runs: a quad entry is 2 bytes and a FASTQ holds at most len / 8 quads (sequence
    // lines are less than half of the text), so len / 4 bytes however they spread over the 256 buckets; plus the
    // blocks a drain may leave unused at the end of a run, one open run per (workgroup, queue), one reserve per wave.
    uint64_t runs = (maxlen / 4 + maxlen / 32) / kRunBytes + 16 + static_cast<uint64_t>(parts) * (kQuadBuckets + kWaves * kPoolRuns);
    if (ctx->spill_runs_cap) runs = ctx->spill_runs_cap;  // VKIMG_SPILL_RUNS_CAP: tests force the arena-full fallback
    if (runs >= (1u << 24)) return VK_EINVAL;
    // the regions of quads of which only some windows count, one per (workgroup of pass A, bucket): ~1 per 215 bytes of
    // 150-base reads over 256 buckets; room for four times that (uniform bases), at least 64
    const uint64_t wg_bytes = maxlen / parts + 64 * kWaves;
    uint32_t pshift = 6;                                   // (a power of two: the place is a shift and an or)
    while (pshift < 21 && (1ull << pshift) < wg_bytes / 16384 + 64) ++pshift;   // (at most 2^21: a workgroup's 256 regions stay below 2^31 bytes, the size a buffer descriptor takes as an int; what does not fit is counted directly)
    if (ctx->spill_misc_cap) pshift = ctx->spill_misc_cap < 21 ? ctx->spill_misc_cap : 21;   // VKIMG_SPILL_MISC_CAP (log2): tests force the region-full fallback
    const uint64_t pcap = 1ull << pshift;
    const size_t preg_words = static_cast<size_t>(parts) * kQuadBuckets * (pcap + 1);   // per sample: regions, populations
    const size_t per_sample = static_cast<size_t>(runs) * (kRunBytes + 2 * sizeof(uint32_t)) + (3 * kQuadBuckets + 2) * sizeof(uint32_t) +
                              kQuadBuckets * kOutWords * sizeof(uint32_t) + preg_words * sizeof(uint32_t) + 64;
    size_t free_b = 0, total_b = 0;
    VK_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    size_t budget = ctx->spill_budget;
    const size_t avail = free_b / 4 * 3 + ctx->spill_cap;
    if (budget > avail) budget = avail;
    uint32_t batch = static_cast<uint32_t>(budget / per_sample);
    if (batch == 0) batch = 1;
    if (batch > nsamples) batch = nsamples;
    // workspace: [zeroed per sub-batch: cursors[batch] | hdrs[batch][runs]] | qfirst[batch][257] | qlist[batch][runs] |
    //            preg_n[batch * parts][256] | preg[batch * parts][256][pcap] | window arrays[batch][256][4][4^(K-4)] | arena[batch][runs][4 KiB]
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t head_bytes = up(static_cast<size_t>(batch) * (1 + runs) * sizeof(uint32_t));
    const size_t list_bytes = up(static_cast<size_t>(batch) * (3 * kQuadBuckets + 1 + runs) * sizeof(uint32_t));   // qfirst | qlist | job sizes | job order
    const size_t misc_bytes = up(static_cast<size_t>(batch) * preg_words * sizeof(uint32_t));
    const size_t bh_bytes = static_cast<size_t>(batch) * kQuadBuckets * kOutWords * sizeof(uint32_t);
    const size_t arena_bytes = static_cast<size_t>(batch) * runs * kRunBytes;
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_spill), &ctx->spill_cap, head_bytes + list_bytes + misc_bytes + bh_bytes + arena_bytes);
    if (rc) return rc;
    uint8_t* const base = reinterpret_cast<uint8_t*>(ctx->d_spill);
    BucketParams bp{};
    bp.cursors = ctx->d_spill;
    bp.hdrs = ctx->d_spill + batch;
    bp.qfirst = reinterpret_cast<uint32_t*>(base + head_bytes);
    bp.qlist = bp.qfirst + static_cast<size_t>(batch) * (kQuadBuckets + 1);
    bp.bsize = bp.qlist + static_cast<size_t>(batch) * runs;
    bp.order = bp.bsize + static_cast<size_t>(batch) * kQuadBuckets;
    bp.preg_n = reinterpret_cast<uint32_t*>(base + head_bytes + list_bytes);
    bp.preg = bp.preg_n + static_cast<size_t>(batch) * parts * kQuadBuckets;
    bp.preg_shift = pshift;
    bp.parts = parts;
    bp.bucket_hist = reinterpret_cast<uint32_t*>(base + head_bytes + list_bytes + misc_bytes);
    bp.arena = base + head_bytes + list_bytes + misc_bytes + bh_bytes;
    bp.runs_cap = static_cast<uint32_t>(runs);
    VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * NCODE * sizeof(uint32_t), ctx->stream));
    ctx->last_block = kCountThreads;
    ctx->last_lds = kLdsBucketBytes;
    for (uint32_t s0 = 0; s0 < nsamples; s0 += batch) {
        const uint32_t n = nsamples - s0 < batch ? nsamples - s0 : batch;
        VK_HIP(ctx, hipMemsetAsync(ctx->d_spill, 0, head_bytes, ctx->stream));  // cursors and run headers
        ctx->last_grid = n * parts;
        uint32_t* const hist0 = d_hist + static_cast<size_t>(s0) * NCODE;
        uint32_t* const wph0 = ctx->d_wavephase + static_cast<size_t>(s0) * parts * kWaves;
        hipLaunchKernelGGL((vk_bucket_kernel<K, 3>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                           d_fastq, d_offs + s0, d_lens + s0, n, parts, hist0, wph0, bp, SubParams{}, PackParams{});
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(vk_quad_list_kernel, dim3(n), dim3(1024), 0, ctx->stream, bp);
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(vk_bucket_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, bp, n * kQuadBuckets);
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((vk_quad_count_kernel<K>), dim3(n * kQuadBuckets), dim3(512), 0, ctx->stream, bp);
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((vk_quad_merge_kernel<K>), dim3(n * (NCODE / kMergeTile)), dim3(256), 0, ctx->stream, bp, hist0);
        VK_HIP(ctx, hipGetLastError());
    }
    return VK_OK;
}

// k = 8, 9: bucket pass + replay pass, in sub-batches that fit the spill budget.
template <int K>
int launch_spill(vk_ctx* ctx, const uint8_t* d_fastq, const uint64_t* d_offs, const uint64_t* d_lens,
                 uint32_t nsamples, uint32_t parts, uint64_t maxlen, uint32_t* d_hist, const SubParams* sub) {
    constexpr uint32_t NCODE = 1u << (2 * K);
    if (sub == nullptr && !ctx->spill_packed && !ctx->spill_pairs)
        return launch_spill_quad<K>(ctx, d_fastq, d_offs, d_lens, nsamples, parts, maxlen, d_hist);
    // Arena of one sample, in 4 KiB runs: a pair entry is 2 bytes and a FASTQ holds at most len / 4
    // pairs (sequence lines are less than half of the text), so len / 2 bytes however the pairs spread
    // over the 16 buckets; plus the blocks a drain may leave unused at the end of a run (at most 3 of
    // 64), plus one open run per (workgroup, wave, queue).
    uint64_t runs = (maxlen / 2 + maxlen / 16) / kRunBytes + 16 + static_cast<uint64_t>(parts) * kWaves * (kQueues + kPoolRuns);   // (+ one open run per queue and one reserve per wave)
    if (ctx->spill_runs_cap) runs = ctx->spill_runs_cap;  // VKIMG_SPILL_RUNS_CAP: tests force the arena-full fallback
    if (runs >= (1u << 24)) return VK_EINVAL;             // a run number travels in 24 bits (64 GiB of entries per sample)
    constexpr size_t kBucketHistBytes = static_cast<size_t>(kQueues) * (2u << (2 * K - 4)) * sizeof(uint32_t);  // pass B -> merge
    // the packed stream of a sample (vk_pack.h): a record per 16 bytes of text at most, a few per wavefront on top
    const bool packed = sub == nullptr && ctx->spill_packed;
    const uint64_t pack_recs = packed ? ((maxlen + 15) / 16 + 8ull * parts * kWaves + 16 + 3) / 4 * 4 : 0;
    const size_t per_sample = static_cast<size_t>(runs) * (kRunBytes + sizeof(uint32_t)) + sizeof(uint32_t) + 3 * kQueues * sizeof(uint32_t) + kBucketHistBytes +
                              pack_recs * 8 + 64;
    // never plan for more than three quarters of what is free (plus what this context already holds)
    size_t free_b = 0, total_b = 0;
    VK_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    size_t budget = ctx->spill_budget;
    const size_t avail = free_b / 4 * 3 + ctx->spill_cap;
    if (budget > avail) budget = avail;
    uint32_t batch = static_cast<uint32_t>(budget / per_sample);
    if (batch == 0) batch = 1;
    if (batch > nsamples) batch = nsamples;
    // workspace: cursors[batch] | hdrs[batch][runs] | bucket sizes[batch][16] | job order[batch * 16] | wide flags[batch * 16] |
    //            bucket histograms[batch][16][2 * 4^K / 16] | arena[batch][runs][4 KiB]
    const size_t head_bytes = ((static_cast<size_t>(batch) * (1 + runs + 3 * kQueues)) * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t bh_bytes = static_cast<size_t>(batch) * kBucketHistBytes;
    const size_t arena_bytes = static_cast<size_t>(batch) * runs * kRunBytes;
    // packed stream: codes | masks (+ a block of slack: the last wavefront's loads run to the end of its last block) | sample bases | counts
    const size_t pack_arr = packed ? (static_cast<size_t>(batch) * pack_recs + kPackBlock) * sizeof(uint32_t) : 0;
    const size_t pack_meta = packed ? (static_cast<size_t>(batch) * sizeof(uint64_t) + static_cast<size_t>(batch) * parts * kWaves * sizeof(uint32_t) + 255) / 256 * 256 : 0;
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_spill), &ctx->spill_cap,
                    head_bytes + bh_bytes + arena_bytes + 2 * pack_arr + pack_meta);
    if (rc) return rc;
    BucketParams bp;
    bp.cursors = ctx->d_spill;
    bp.hdrs = ctx->d_spill + batch;
    bp.bsize = bp.hdrs + static_cast<size_t>(batch) * runs;
    bp.order = bp.bsize + static_cast<size_t>(batch) * kQueues;
    bp.wide = bp.order + static_cast<size_t>(batch) * kQueues;
    bp.force_wide = ctx->spill_force_wide ? 1u : 0u;
    bp.bucket_hist = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ctx->d_spill) + head_bytes);
    bp.arena = reinterpret_cast<uint8_t*>(ctx->d_spill) + head_bytes + bh_bytes;
    bp.runs_cap = static_cast<uint32_t>(runs);
    PackParams pk{};
    uint32_t aside_cap = 0;
    uint32_t* d_aside_n = nullptr;
    if (packed) {
        uint8_t* at = reinterpret_cast<uint8_t*>(ctx->d_spill) + head_bytes + bh_bytes + arena_bytes;
        pk.c = reinterpret_cast<uint32_t*>(at);
        pk.m = reinterpret_cast<uint32_t*>(at + pack_arr);
        uint64_t* d_base = reinterpret_cast<uint64_t*>(at + 2 * pack_arr);
        pk.base = d_base;
        pk.count = reinterpret_cast<uint32_t*>(d_base + batch);
        std::vector<uint64_t> base(batch);
        for (uint32_t i = 0; i < batch; ++i) base[i] = static_cast<uint64_t>(i) * pack_recs;
        VK_HIP(ctx, hipMemcpyAsync(d_base, base.data(), batch * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (base is a temporary; once per call)
        // the waves' lists of lanes set aside, as for the k <= 7 kernel
        const uint64_t wave_bytes = maxlen / (static_cast<uint64_t>(parts) * kWaves) + 64;
        uint64_t cap = wave_bytes / kPiece + 64;
        if (cap > (1u << 20)) cap = 1u << 20;
        aside_cap = static_cast<uint32_t>(cap);
        const size_t nwaves = static_cast<size_t>(batch) * parts * kWaves;
        rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_aside), &ctx->aside_cap, nwaves * (cap + 1) * sizeof(uint32_t));
        if (rc) return rc;
        d_aside_n = ctx->d_aside + nwaves * cap;
    }
    VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * NCODE * sizeof(uint32_t), ctx->stream));
    ctx->last_block = kCountThreads;
    ctx->last_lds = kLdsBucketBytes;
    for (uint32_t s0 = 0; s0 < nsamples; s0 += batch) {
        const uint32_t n = nsamples - s0 < batch ? nsamples - s0 : batch;
        VK_HIP(ctx, hipMemsetAsync(ctx->d_spill, 0, head_bytes, ctx->stream));  // cursors and run headers
        ctx->last_grid = n * parts;
        uint32_t* const hist0 = d_hist + static_cast<size_t>(s0) * NCODE;
        uint32_t* const wph0 = ctx->d_wavephase + static_cast<size_t>(s0) * parts * kWaves;
        if (sub) {
            SubParams sp = *sub;  // this sub-batch's slice of the per-sample arrays
            sp.seeds += s0;
            sp.thresholds += s0;
            if (sp.sites) sp.sites += 2ull * s0;
            hipLaunchKernelGGL((vk_bucket_kernel<K, 1>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                               d_fastq, d_offs + s0, d_lens + s0, n, parts, hist0, wph0, bp, sp, PackParams{});
        } else if (packed) {
            // pass A in two kernels: the text packed once (the k <= 7 kernel's front end), the stream partitioned
            hipLaunchKernelGGL(vk_pack_kernel, dim3(n * parts), dim3(kCountThreads), 0, ctx->stream, d_fastq, d_offs + s0,
                               d_lens + s0, n, parts, pk, wph0, ctx->d_aside, aside_cap, d_aside_n);
            VK_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL((vk_aside_kernel<K, false>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream, d_fastq,
                               d_offs + s0, d_lens + s0, n, parts, hist0, ctx->d_aside, aside_cap, d_aside_n, IndexParams{}, 1u);
            VK_HIP(ctx, hipGetLastError());
            hipLaunchKernelGGL((vk_bucket_kernel<K, 2>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                               d_fastq, d_offs + s0, d_lens + s0, n, parts, hist0, wph0, bp, SubParams{}, pk);
        } else {
            hipLaunchKernelGGL((vk_bucket_kernel<K, 0>), dim3(n * parts), dim3(kCountThreads), 0, ctx->stream,
                               d_fastq, d_offs + s0, d_lens + s0, n, parts, hist0, wph0, bp, SubParams{}, PackParams{});
        }
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(vk_bucket_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, bp, n * kQueues);
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((vk_bucket_count_kernel<K>), dim3(n * kQueues), dim3(kCountThreads), 0, ctx->stream, bp);
        VK_HIP(ctx, hipGetLastError());
        // (jobs whose u16 pair counters wrapped: replayed into u32 window counters; every other workgroup leaves at once)
        hipLaunchKernelGGL((vk_bucket_count_wide_kernel<K>), dim3(n * kQueues), dim3(kCountThreads), 0, ctx->stream, bp);
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL((vk_bucket_merge_kernel<K>), dim3(n * (NCODE / 256)), dim3(256), 0, ctx->stream, bp,
                           d_hist + static_cast<size_t>(s0) * NCODE);
        VK_HIP(ctx, hipGetLastError());
    }
    return VK_OK;
}

uint32_t choose_parts(uint32_t nsamples, uint64_t maxlen, int k) {
    // The count kernels keep two workgroups per CU resident: 512 slots on the 256 CUs.
    constexpr uint64_t kSlots = 512;
    if (k <= 7) {
        // Round 6: MANY more workgroups than slots.  Rounds 2-5 launched 1000 samples as 1000 workgroups of 320 MB -- two
        // "rounds" of the chip in lock-step -- and the wave-cycle counter read 85 % of the launch (profiles/r05a): a
        // workgroup's slot is held until its slowest wavefront is through its sixteenth of the sample, and a CU that runs
        // a few per cent behind (another XCD's L2, a busier memory channel) ends the launch alone.  In workgroups of
        // 10-27 MB (20 MiB is the rule) the dispatcher hands the faster CUs more of them: 65.6 -> 62.6 ms for 1000 samples, 7.13 -> 6.63 for 100,
        // 74.1 -> 71.1 on fastp-shaped reads (profiles/ab/r06_parts.txt; flat from 12 to 32 parts, slower again from 64:
        // every workgroup zeroes and flushes a 64 KB histogram and every wavefront finds its line phase).  At least four
        // rounds of the chip where that leaves workgroups of 4 MiB.
        const uint64_t cap4 = maxlen / (4ull << 20) > 1 ? maxlen / (4ull << 20) : 1;
        uint64_t parts = (maxlen + (10ull << 20)) / (20ull << 20);
        if (parts < 1) parts = 1;
        const uint64_t want = (4 * kSlots + nsamples - 1) / nsamples;
        if (parts < want) parts = want < cap4 ? want : cap4;
        if (parts > 512) parts = 512;
        if (nsamples * parts >= 4 * kSlots) return static_cast<uint32_t>(parts);
    } else if (nsamples >= 4 * kSlots) {
        return 1;
    }
    // Few workgroups after all (a handful of samples; k = 8, 9, whose workspace is sized by the workgroup): a launch of G equal
    // workgroups runs in ceil(G / 512) rounds, so G just below a multiple of 512 wastes the least (100 samples: 5 parts = 500
    // workgroups, not 8 = 800 in two rounds).  Parts never get so small (< 1 MiB) that the per-wave range sync shows.
    uint32_t best = 1;
    double best_eff = 0.0;
    for (uint32_t parts = 1; parts <= 512; ++parts) {
        if (parts > 1 && maxlen / parts < (1u << 20)) break;
        const uint64_t g = static_cast<uint64_t>(nsamples) * parts;
        const double eff = static_cast<double>(g) / static_cast<double>((g + kSlots - 1) / kSlots * kSlots);
        if (eff > best_eff) {
            best_eff = eff;
            best = parts;
        }
        if (eff >= 0.95) break;  // good enough: more parts only add range syncs, flushes and open runs
    }
    return best;
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
        const char* g = getenv("VKIMG_GZ_NO_CHUNKS");
        ctx->gz_no_chunks = g && g[0] == '1';
        const char* cb = getenv("VKIMG_GZ_CHUNK_BYTES");
        if (cb && cb[0]) ctx->gz_chunk_bytes = static_cast<uint32_t>(strtoul(cb, nullptr, 10));
        const char* sf = getenv("VKIMG_GZ_SPLIT_FIND");
        ctx->gz_split_find = sf && sf[0] == '1';
        const char* fp = getenv("VKIMG_GZ_FILL");
        if (fp && fp[0]) ctx->gz_fill_pct = static_cast<uint32_t>(strtoul(fp, nullptr, 10));
        const char* lp = getenv("VKIMG_GZ_LDS_PAD");
        if (lp && lp[0]) ctx->gz_lds_pad = static_cast<uint32_t>(strtoul(lp, nullptr, 10));
        const char* r = getenv("VKIMG_SPILL_RUNS_CAP");
        if (r && r[0]) ctx->spill_runs_cap = static_cast<uint32_t>(strtoul(r, nullptr, 10));
        const char* tf = getenv("VKIMG_SPILL_PACKED");
        ctx->spill_packed = tf && tf[0] == '1';
        const char* mc = getenv("VKIMG_SPILL_MISC_CAP");
        if (mc && mc[0]) ctx->spill_misc_cap = static_cast<uint32_t>(strtoul(mc, nullptr, 10));
        const char* sq = getenv("VKIMG_SPILL_PAIRS");
        ctx->spill_pairs = sq && sq[0] == '1';
        const char* fw = getenv("VKIMG_SPILL_FORCE_WIDE");
        ctx->spill_force_wide = fw && fw[0] == '1';
        const char* kc = getenv("VKIMG_K1_CLASSIC");
        ctx->k1_classic = kc && kc[0] == '1';
    }
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return VK_EHIP; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }
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
    void* ptrs[] = {ctx->d_desc, ctx->d_wavephase, ctx->d_scratch, ctx->d_spill, ctx->d_stage, ctx->d_hist1, ctx->d_status1, ctx->d_img1, ctx->d_sub, ctx->d_gzjobs, ctx->d_gzmeta, ctx->d_gzsym, ctx->d_gzwin, ctx->d_gzcrc, ctx->d_synth, ctx->d_synth_offs, ctx->d_aside, ctx->d_index, ctx->d_walk};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
    if (ctx->h_desc_alt) (void)hipHostFree(ctx->h_desc_alt);
    if (ctx->desc_ev) (void)hipEventDestroy(ctx->desc_ev);
    if (ctx->desc_ev_alt) (void)hipEventDestroy(ctx->desc_ev_alt);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int vk_ctx_sync(vk_ctx* ctx) {
    if (!ctx) return VK_EINVAL;
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VK_OK;
}

int vk_host_register(vk_ctx* ctx, const void* p, uint64_t nbytes) {
    if (!ctx || !p || nbytes == 0) return VK_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return VK_EHIP;   // (the calling thread may be an I/O thread that never chose a device)
    const hipError_t e = hipHostRegister(const_cast<void*>(p), nbytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return VK_EHIP;   // (ctx->last is left alone: this may run beside the context's own thread)
    }
    return VK_OK;
}

int vk_host_unregister(vk_ctx* ctx, const void* p) {
    if (!ctx || !p) return VK_EINVAL;
    if (hipSetDevice(ctx->device) != hipSuccess) return VK_EHIP;
    return hipHostUnregister(const_cast<void*>(p)) == hipSuccess ? VK_OK : VK_EHIP;
}

int vk_upload_mapped(vk_ctx* ctx, void* d_dst, const uint64_t* dst_offsets, const void* const* h_src,
                     const uint64_t* nbytes, const uint8_t* registered, uint32_t nfiles, uint32_t* status) {
    if (!ctx || !d_dst || (nfiles && (!dst_offsets || !h_src || !nbytes || !status))) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<char> reg(nfiles, 0), mine(nfiles, 0);
    auto pin = [&](uint32_t i) {   // the file's page-cache pages, pinned and mapped for the DMA engines
        status[i] = 0u;
        if (nbytes[i] == 0) return;
        if (registered && registered[i]) {
            reg[i] = 1;
            return;
        }
        mine[i] = 1;
        if (!h_src[i] || hipHostRegister(const_cast<void*>(h_src[i]), nbytes[i], hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            status[i] = 1u;
            mine[i] = 0;
        } else {
            reg[i] = 1;
        }
    };
    if (nfiles) pin(0);
    int rc = VK_OK;
    for (uint32_t i = 0; i < nfiles; ++i) {
        if (reg[i]) {
            const hipError_t e = hipMemcpyAsync(static_cast<uint8_t*>(d_dst) + dst_offsets[i], h_src[i], nbytes[i],
                                                hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) {
                ctx->last = e;
                status[i] = 2u;
                rc = VK_EHIP;
            }
        }
        if (i + 1 < nfiles) pin(i + 1);   // while file i is in flight
    }
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    for (uint32_t i = 0; i < nfiles; ++i)
        if (mine[i]) (void)hipHostUnregister(const_cast<void*>(h_src[i]));
    if (es != hipSuccess) {
        ctx->last = es;
        return VK_EHIP;
    }
    return rc;
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

}  // extern "C"

// Subsampled counts through the read index (vk_ladder.h): every (offset, length) of the call is a sample the context
// holds an index of.  Returns VK_OK with *done = true when the walker ran.
constexpr uint64_t kWalkMaxThresholdSpill = (1ull << 32) / 32;   // k = 8, 9: largest fraction of the reads (as a threshold) the walker takes

template <int K>
static void launch_walk(vk_ctx* ctx, const uint8_t* fq, const uint64_t* d_offs, const uint64_t* d_lens, uint32_t npairs,
                        const WalkParams& wp, uint32_t* d_hist, int atomic_flush) {
    hipLaunchKernelGGL((vk_walk_kernel<K>), dim3(npairs * wp.iparts), dim3(kCountThreads), 0, ctx->stream, fq, d_offs, d_lens, npairs,
                       ctx->ix, wp, d_hist, atomic_flush);
}

static int count_walk(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths, uint32_t nsamples, int k,
                      uint32_t* d_hist, uint32_t* d_status, const uint64_t* seeds, const uint64_t* thresholds, uint64_t* d_sites,
                      bool* done) {
    *done = false;
    if (ctx->ix_fastq != d_fastq || ctx->ix_sample.empty() || getenv("VKIMG_NO_READ_INDEX")) return VK_OK;
    std::vector<uint32_t> isample(nsamples), status(nsamples);
    for (uint32_t i = 0; i < nsamples; ++i) {
        // k = 8, 9: the walker counts into the subsample's row in HBM, one atomic per window (~28 G/s measured): it beats a
        // streamed pass of the spill path only for subsamples of a few per cent of the reads
        if (k > 7 && thresholds[i] > kWalkMaxThresholdSpill) return VK_OK;
        const auto it = ctx->ix_sample.find(std::make_pair(offsets[i], lengths[i]));
        if (it == ctx->ix_sample.end() || ctx->ix_overflow[it->second]) return VK_OK;   // not indexed: the streaming kernels
        if (lengths[i] >= (1ull << 32) - 4096) return VK_OK;                             // (the walker's offsets are 32-bit)
        isample[i] = it->second;
        status[i] = ctx->ix_status[it->second];
    }
    const uint32_t parts = ctx->ix_parts;
    if (static_cast<uint64_t>(nsamples) * parts > (1u << 24)) return VK_OK;
    // per-pair arrays: offsets | lengths | seeds | thresholds (u64 each) | isample (u32)
    const size_t need = static_cast<size_t>(nsamples) * (4 * sizeof(uint64_t) + sizeof(uint32_t)) + 64;
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_walk), &ctx->walk_cap, need);
    if (rc) return rc;
    uint64_t* d64 = reinterpret_cast<uint64_t*>(ctx->d_walk);
    uint32_t* d_is = reinterpret_cast<uint32_t*>(d64 + 4ull * nsamples);
    VK_HIP(ctx, hipMemcpyAsync(d64, offsets, nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d64 + nsamples, lengths, nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d64 + 2ull * nsamples, seeds, nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d64 + 3ull * nsamples, thresholds, nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d_is, isample.data(), nsamples * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d_status, status.data(), nsamples * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the host vectors are temporaries)
    if (d_sites) VK_HIP(ctx, hipMemsetAsync(d_sites, 0, 2ull * nsamples * sizeof(uint64_t), ctx->stream));
    const int atomic_flush = (parts > 1 || k > 7) ? 1 : 0;
    if (atomic_flush)
        VK_HIP(ctx, hipMemsetAsync(d_hist, 0, static_cast<size_t>(nsamples) * (1ull << (2 * k)) * sizeof(uint32_t), ctx->stream));
    WalkParams wp;
    wp.isample = d_is;
    wp.seeds = d64 + 2ull * nsamples;
    wp.thresholds = d64 + 3ull * nsamples;
    wp.sites = reinterpret_cast<unsigned long long*>(d_sites);
    wp.iparts = parts;
    const uint8_t* fq = static_cast<const uint8_t*>(d_fastq);
    ctx->last_grid = nsamples * parts;
    ctx->last_block = kCountThreads;
    ctx->last_lds = (k <= 7 ? (1u << (2 * k)) * 4u : 4u) + kWaves * kWalkQueue * 4u;
    switch (k) {
        case 5: launch_walk<5>(ctx, fq, d64, d64 + nsamples, nsamples, wp, d_hist, atomic_flush); break;
        case 6: launch_walk<6>(ctx, fq, d64, d64 + nsamples, nsamples, wp, d_hist, atomic_flush); break;
        case 7: launch_walk<7>(ctx, fq, d64, d64 + nsamples, nsamples, wp, d_hist, atomic_flush); break;
        case 8: launch_walk<8>(ctx, fq, d64, d64 + nsamples, nsamples, wp, d_hist, atomic_flush); break;
        default: launch_walk<9>(ctx, fq, d64, d64 + nsamples, nsamples, wp, d_hist, atomic_flush); break;
    }
    VK_HIP(ctx, hipGetLastError());
    *done = true;
    return VK_OK;
}

extern "C" {

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
    if (seeds) {   // subsamples of samples the context holds a read index of: the walker (vk_ladder.h)
        bool done = false;
        const int rcw = count_walk(ctx, d_fastq, offsets, lengths, nsamples, k, d_hist, d_status, seeds, thresholds, d_sites, &done);
        if (rcw || done) return rcw;
    }
    uint32_t parts = parts_per_sample ? parts_per_sample : choose_parts(nsamples, maxlen, k);
    if (!parts_per_sample && k >= 8 && seeds == nullptr && !ctx->spill_pairs && !ctx->spill_packed) {
        // The quad route's workgroups meet at barriers: twice as many, half as long, when that fills the chip's 512 slots as
        // well (100 samples: 5 -> 10 parts, 500 -> 1000 workgroups), ends a launch with less of a tail: 13.09 against 13.18 ms.
        const uint64_t g1 = static_cast<uint64_t>(nsamples) * parts, g2 = 2 * g1;
        const double e1 = static_cast<double>(g1) / static_cast<double>((g1 + 511) / 512 * 512);
        const double e2 = static_cast<double>(g2) / static_cast<double>((g2 + 511) / 512 * 512);
        if (g2 <= 2048 && e2 >= e1 - 0.005 && maxlen / (2ull * parts) >= (1u << 20)) parts *= 2;
    }
    // a wavefront addresses its byte range through a 32-bit buffer descriptor (vk_count.h, wave_stream):
    // keep every range below 2 GiB, whatever the caller asked for
    while (maxlen / (static_cast<uint64_t>(parts) * kWaves) >= (1ull << 31)) parts *= 2;
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
    ctx->last_waves = static_cast<uint64_t>(nsamples) * parts * kWaves;
    ctx->last_bytes = 0;
    for (uint32_t i = 0; i < nsamples; ++i) ctx->last_bytes += lengths[i];
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

}  // extern "C"

namespace {

// Workspace and descriptors of a read index over the samples of a call (vk_ladder.h): *ip describes it, *parts_out is the
// split the index is laid out for.  The samples' descriptors are on the device (ctx->d_desc) on return.
int index_prepare(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths, uint32_t nsamples,
                  uint32_t parts_per_sample, IndexParams* ip, uint32_t* parts_out, uint64_t* maxlen_out) {
    ctx->ix_fastq = nullptr;
    ctx->ix_sample.clear();
    if (!d_fastq || (reinterpret_cast<uintptr_t>(d_fastq) & 15u) != 0) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t maxlen = 0;
    for (uint32_t i = 0; i < nsamples; ++i) {
        if ((offsets[i] & 15u) != 0) return VK_EINVAL;
        if (lengths[i] > maxlen) maxlen = lengths[i];
    }
    // (the walker runs a workgroup per (subsample, part) of this split: the rule of rounds 2-5, one round of the chip)
    uint32_t parts = parts_per_sample ? parts_per_sample : choose_parts(nsamples, maxlen, 9);
    while (maxlen / (static_cast<uint64_t>(parts) * kWaves) >= (1ull << 31)) parts *= 2;
    if (static_cast<uint64_t>(nsamples) * parts > (1u << 24)) return VK_EINVAL;
    const size_t nwaves = static_cast<size_t>(nsamples) * parts * kWaves;
    // anchors: a sample's region holds one per 32 bytes of text, and eight per wavefront on top
    std::vector<uint64_t> base(nsamples);
    uint64_t total = 0;
    for (uint32_t i = 0; i < nsamples; ++i) {
        base[i] = total;
        total += lengths[i] / 32 + 8ull * parts * kWaves + 16;
    }
    const size_t o_base = (total * sizeof(uint32_t) + 255) / 256 * 256, o_count = o_base + nsamples * sizeof(uint64_t),
                 o_sites = (o_count + nwaves * sizeof(uint32_t) + 7) / 8 * 8, o_over = o_sites + nsamples * sizeof(uint64_t),
                 bytes = o_over + nsamples * sizeof(uint32_t);
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_index), &ctx->index_cap, bytes + 256);
    if (rc) return rc;
    if (ctx->desc_cap < 2ull * nsamples * sizeof(uint64_t)) ctx->desc_n = 0;
    rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_desc), &ctx->desc_cap, 2ull * nsamples * sizeof(uint64_t));
    if (rc) return rc;
    rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_wavephase), &ctx->wavephase_cap, nwaves * sizeof(uint32_t));
    if (rc) return rc;
    rc = upload_desc(ctx, offsets, lengths, nsamples);
    if (rc) return rc;
    uint8_t* m = ctx->d_index;
    ip->anchors = reinterpret_cast<uint32_t*>(m);
    ip->base = reinterpret_cast<const uint64_t*>(m + o_base);
    ip->count = reinterpret_cast<uint32_t*>(m + o_count);
    ip->sites = reinterpret_cast<unsigned long long*>(m + o_sites);
    ip->overflow = reinterpret_cast<uint32_t*>(m + o_over);
    VK_HIP(ctx, hipMemcpyAsync(m + o_base, base.data(), nsamples * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (base is a temporary)
    VK_HIP(ctx, hipMemsetAsync(m + o_sites, 0, bytes - o_sites, ctx->stream));
    *parts_out = parts;
    *maxlen_out = maxlen;
    return VK_OK;
}

// The index is filled and d_status holds the samples' status words: bring sites, overflow flags and status to the host and
// remember which samples the index covers.
int index_finish(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths, uint32_t nsamples,
                 const IndexParams& ip, uint32_t parts, const uint32_t* d_status, uint64_t* sites, uint32_t* status) {
    ctx->ix_status.assign(nsamples, 0u);
    ctx->ix_overflow.assign(nsamples, 0u);
    VK_HIP(ctx, hipMemcpyAsync(sites, ip.sites, nsamples * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(ctx->ix_overflow.data(), ip.overflow, nsamples * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(ctx->ix_status.data(), d_status, nsamples * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < nsamples; ++i) {
        status[i] = ctx->ix_status[i];
        ctx->ix_sample[std::make_pair(offsets[i], lengths[i])] = i;
    }
    ctx->ix = ip;
    ctx->ix_fastq = d_fastq;
    ctx->ix_parts = parts;
    return VK_OK;
}

}  // namespace

extern "C" {

int vk_read_index_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths, uint32_t nsamples,
                         uint32_t parts_per_sample, uint64_t* sites, uint32_t* status) {
    if (!ctx || !offsets || !lengths || !sites || !status) return VK_EINVAL;
    ctx->ix_fastq = nullptr;
    ctx->ix_sample.clear();
    if (nsamples == 0) return VK_OK;
    IndexParams ip;
    uint32_t parts = 0;
    uint64_t maxlen = 0;
    int rc = index_prepare(ctx, d_fastq, offsets, lengths, nsamples, parts_per_sample, &ip, &parts, &maxlen);
    if (rc) return rc;
    rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_status1), &ctx->status1_cap, nsamples * sizeof(uint32_t) + 64);
    if (rc) return rc;
    const uint8_t* fq = static_cast<const uint8_t*>(d_fastq);
    hipLaunchKernelGGL(vk_index_kernel, dim3(nsamples * parts), dim3(kCountThreads), 0, ctx->stream, fq, ctx->d_desc, ctx->d_desc + nsamples,
                       nsamples, parts, ip, ctx->d_wavephase);
    VK_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(vk_check_kernel, dim3((nsamples + 255) / 256), dim3(256), 0, ctx->stream, fq, ctx->d_desc, ctx->d_desc + nsamples,
                       nsamples, parts, ctx->d_wavephase, ctx->d_status1);
    VK_HIP(ctx, hipGetLastError());
    return index_finish(ctx, d_fastq, offsets, lengths, nsamples, ip, parts, ctx->d_status1, sites, status);
}

int vk_count_index_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths, uint32_t nsamples, int k,
                          uint32_t parts_per_sample, uint32_t* d_hist, uint32_t* d_status, uint64_t* sites, uint32_t* status) {
    if (!ctx || !offsets || !lengths || !d_hist || !d_status || !sites || !status || k < 5 || k > 9) return VK_EINVAL;
    ctx->ix_fastq = nullptr;
    ctx->ix_sample.clear();
    if (nsamples == 0) return VK_OK;
    if (k > 7) {   // the spill path has no index mode: the index in a pass of its own, then the count
        int rc = vk_read_index_device(ctx, d_fastq, offsets, lengths, nsamples, parts_per_sample, sites, status);
        if (rc) return rc;
        return vk_count_device(ctx, d_fastq, offsets, lengths, nsamples, k, parts_per_sample, d_hist, d_status);
    }
    IndexParams ip;
    uint32_t parts = 0;
    uint64_t maxlen = 0;
    int rc = index_prepare(ctx, d_fastq, offsets, lengths, nsamples, parts_per_sample, &ip, &parts, &maxlen);
    if (rc) return rc;
    const uint8_t* fq = static_cast<const uint8_t*>(d_fastq);
    uint64_t* d_offs = ctx->d_desc;
    uint64_t* d_lens = ctx->d_desc + nsamples;
    switch (k) {
        case 5: rc = launch_count<5>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, nullptr, &ip); break;
        case 6: rc = launch_count<6>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, nullptr, &ip); break;
        default: rc = launch_count<7>(ctx, fq, d_offs, d_lens, nsamples, parts, maxlen, d_hist, nullptr, &ip); break;
    }
    if (rc) return rc;
    ctx->last_waves = static_cast<uint64_t>(nsamples) * parts * kWaves;
    ctx->last_bytes = 0;
    for (uint32_t i = 0; i < nsamples; ++i) ctx->last_bytes += lengths[i];
    hipLaunchKernelGGL(vk_check_kernel, dim3((nsamples + 255) / 256), dim3(256), 0, ctx->stream, fq, d_offs, d_lens,
                       nsamples, parts, ctx->d_wavephase, d_status);
    VK_HIP(ctx, hipGetLastError());
    return index_finish(ctx, d_fastq, offsets, lengths, nsamples, ip, parts, d_status, sites, status);
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
#ifndef VK_IMAGE_COUNT_FROM
#define VK_IMAGE_COUNT_FROM 8192u   // padded pixels from which the order statistics are counted instead of sorted: k >= 7 (round 5: k = 7 too -- 1000 images 0.84 -> 0.23 ms)
#endif
    if (npad >= VK_IMAGE_COUNT_FROM && !ctx->image_sort_only) {
        // order statistics by counting; samples it cannot finish are flagged for the sort
        uint32_t* flags = ctx->d_scratch + work / sizeof(uint32_t);
        // the scatter first, spread over the device (vk_image.h); a batch too large for one grid keeps it in the kernel
        const int scattered = nsamples <= 65535u ? 1 : 0;
        if (scattered) {
            VK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, work, ctx->stream));
            hipLaunchKernelGGL(vk_image_scatter_kernel, dim3((1u << (2 * k)) / 256u, nsamples), dim3(256), 0, ctx->stream, d_hist,
                               ctx->d_pix[k], k, npad, ctx->d_scratch);
            VK_HIP(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(vk_image_count_kernel, dim3(nsamples), dim3(kImgThreads), 0, ctx->stream, d_hist,
                           ctx->d_pix[k], k, npix, npad, ctx->d_scratch, d_img, flags, scattered);
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
    {
        const int rcs = ensure(ctx, reinterpret_cast<void**>(&ctx->d_status1), &ctx->status1_cap, 64);
        if (rcs) return rcs;
    }
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

}  // extern "C"

namespace {

// Operators of the CRC-32 (reflected polynomial 0xEDB88320) over GF(2), as zlib's crc32_combine builds them:
// op[k] advances a remainder over 2^k zero BYTES; 32 columns each.
struct GzCrcOps {
    uint32_t op[34][32];  // [33]: 4032 bytes, the step between a lane's pieces in vk_crc32_seg_kernel
    GzCrcOps() {
        uint32_t bit1[32], bit2[32], bit4[32];
        bit1[0] = 0xEDB88320u;  // one zero bit
        for (int n = 1; n < 32; ++n) bit1[n] = 1u << (n - 1);
        square(bit2, bit1);
        square(bit4, bit2);
        square(op[0], bit4);  // eight zero bits
        for (int k = 1; k <= 32; ++k) square(op[k], op[k - 1]);
        for (int n = 0; n < 32; ++n) op[33][n] = shift(1u << n, 4032);
    }
    static uint32_t times(const uint32_t* m, uint32_t v) {
        uint32_t r = 0;
        for (int i = 0; v; v >>= 1, ++i)
            if (v & 1u) r ^= m[i];
        return r;
    }
    static void square(uint32_t* sq, const uint32_t* m) {
        for (int n = 0; n < 32; ++n) sq[n] = times(m, m[n]);
    }
    uint32_t shift(uint32_t v, uint64_t nbytes) const {
        for (int k = 0; nbytes; nbytes >>= 1, ++k)
            if (nbytes & 1u) v = times(op[k], v);
        return v;
    }
};
const GzCrcOps& gz_crc_ops() {
    static const GzCrcOps ops;
    return ops;
}

// CRC-32 of texts resident on the device (vk_inflate.h, "CRC-32 of the inflated text"): crc[i] for every job.
int gz_text_crc(vk_ctx* ctx, const uint8_t* text, const std::vector<GzCrcJob>& jobs_in, std::vector<uint32_t>& crc) {
    std::vector<GzCrcJob> jobs = jobs_in;
    std::vector<uint32_t> seg_job;
    for (size_t j = 0; j < jobs.size(); ++j) {
        jobs[j].seg0 = static_cast<uint32_t>(seg_job.size());
        jobs[j].nseg = static_cast<uint32_t>((jobs[j].text_len + 65535) / 65536);
        seg_job.insert(seg_job.end(), jobs[j].nseg, static_cast<uint32_t>(j));
    }
    const uint32_t nj = static_cast<uint32_t>(jobs.size()), ns = static_cast<uint32_t>(seg_job.size());
    crc.assign(nj, 0u);
    const GzCrcOps& ops = gz_crc_ops();
    if (ns != 0) {
        const size_t o_jobs = 0, o_segjob = o_jobs + nj * sizeof(GzCrcJob), o_ops = o_segjob + ns * 4ull,
                     o_seg = o_ops + sizeof(ops.op), o_raw = o_seg + ns * 4ull, total = o_raw + nj * 4ull;
        int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzcrc), &ctx->gzcrc_cap, total + 256);
        if (rc) return rc;
        uint8_t* m = ctx->d_gzcrc;
        VK_HIP(ctx, hipMemcpyAsync(m + o_jobs, jobs.data(), nj * sizeof(GzCrcJob), hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(m + o_segjob, seg_job.data(), ns * 4ull, hipMemcpyHostToDevice, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(m + o_ops, ops.op, sizeof(ops.op), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(vk_crc32_seg_kernel, dim3((ns + 3) / 4), dim3(256), 0, ctx->stream, text,
                           reinterpret_cast<const GzCrcJob*>(m + o_jobs), nj, reinterpret_cast<const uint32_t*>(m + o_segjob), ns,
                           reinterpret_cast<const uint32_t*>(m + o_ops), reinterpret_cast<uint32_t*>(m + o_seg));
        VK_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(vk_crc32_fold_kernel, dim3(nj), dim3(64), 0, ctx->stream, reinterpret_cast<const GzCrcJob*>(m + o_jobs),
                           nj, reinterpret_cast<const uint32_t*>(m + o_ops), reinterpret_cast<const uint32_t*>(m + o_seg),
                           reinterpret_cast<uint32_t*>(m + o_raw));
        VK_HIP(ctx, hipGetLastError());
        VK_HIP(ctx, hipMemcpyAsync(crc.data(), m + o_raw, nj * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // zero-preset remainder -> CRC-32: the preset 0xFFFFFFFF shifted over the text, and the final inversion
    for (uint32_t j = 0; j < nj; ++j) crc[j] ^= ops.shift(0xFFFFFFFFu, jobs[j].text_len) ^ 0xFFFFFFFFu;
    return VK_OK;
}

}  // namespace

extern "C" {

int vk_inflate_device(vk_ctx* ctx, const void* d_gz, const uint64_t* gz_offsets, const uint64_t* gz_lengths,
                      uint32_t nfiles, void* d_out, const uint64_t* out_offsets, const uint64_t* out_caps,
                      uint64_t* out_lengths, uint32_t* status) {
    if (!ctx || !gz_offsets || !gz_lengths || !out_offsets || !out_caps || !out_lengths || !status) return VK_EINVAL;
    if (nfiles == 0) return VK_OK;
    if (!d_gz || !d_out) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    const uint8_t* gz = static_cast<const uint8_t*>(d_gz);
    uint8_t* out = static_cast<uint8_t*>(d_out);
    std::vector<uint32_t> direct;  // files that go through the one-wavefront-per-file kernel
    std::vector<GzCrcJob> crc_jobs;        // one per gzip member with text: its CRC-32 is checked against the member's trailer
    std::vector<uint32_t> crc_file, crc_want;
    std::vector<std::pair<uint64_t, uint32_t>> members;  // of the file in hand: (end of the member's text in the file's text, CRC-32 word)
    // every member of file i (text at text_off, text_len bytes; `members` in order) gets a check of its own;
    // an empty member's check word must be that of no bytes
    auto add_member_checks = [&](uint32_t i, uint64_t text_off, uint64_t text_len) {
        uint64_t from = 0;
        for (const auto& mb : members) {
            if (mb.first < from || mb.first > text_len) return;  // (cannot happen for a file whose sizes added up)
            if (mb.first == from) {
                if (mb.second != 0u) status[i] |= VK_GZ_BAD_CRC;
            } else {
                crc_jobs.push_back(GzCrcJob{text_off + from, mb.first - from, 0, 0});
                crc_file.push_back(i);
                crc_want.push_back(mb.second);
            }
            from = mb.first;
        }
    };

    // ---- large files: many wavefronts per file (vk_inflate.h, "the chunked path") ------------------
    std::vector<uint32_t> big;
    for (uint32_t i = 0; i < nfiles; ++i) {
        out_lengths[i] = 0;
        status[i] = 0;
        if (!ctx->gz_no_chunks && gz_lengths[i] >= kGzBigFile) big.push_back(i);
        else direct.push_back(i);
    }
    if (!big.empty()) {
        std::vector<GzChunk> chunks;
        std::vector<uint32_t> chunk0(big.size());
        uint64_t sym_total = 0;
        // Chunk size: 128 KiB gives a small call the most wavefronts (8 files of 24 MB: 48 ms against 72 ms with
        // 256 KiB); once there are more chunks than the device holds wavefronts, throughput no longer
        // depends on it (measured flat from 128 to 320 KiB) and 256 KiB halves the per-chunk work of the finder.
        // Round 6: the chunk decoder's wavefronts are all resident from the start when there are no more of them than the device
        // holds (slots), and a launch of 1.26 rounds is the worst there is (64 files of 24 MB, level 1: 96.9 ms at 192 KiB = 1.26
        // rounds, 88.0 at 128 KiB = 1.9, 80.9 at 256 KiB = 0.95, 86.4 at 320 KiB = 0.76): the size is fitted so that the chunks
        // fill 0.99 of the slots a whole number of times, the fewest times that keep a chunk under 352 KiB -- by the exact count
        // (every file's length is known), so that a SIMD holds six wavefronts or, here and there, five: at 0.96 a quarter of the
        // SIMDs held five and were done a sixth early (64 files of 33 MB: 113.5 ms at 0.96, 102.1 at 0.99 and at 1.00; 0.90: 115.9).
        uint32_t chunk_bytes = ctx->gz_chunk_bytes;
        if (chunk_bytes == 0) {
            uint64_t big_total = 0;
            for (uint32_t i : big) big_total += gz_lengths[i];
            const uint64_t slots = static_cast<uint64_t>(ctx->num_cus) * kGzChunkWaves;
            chunk_bytes = kGzChunkMin;
            if (big_total / kGzChunkMin > slots) {
                for (uint64_t rounds = 1; rounds <= 64; ++rounds) {
                    // (every file ends in a part of a chunk: half a chunk per file goes off the count)
                    const double fill = ctx->gz_fill_pct ? 0.01 * ctx->gz_fill_pct : 0.99;
                    const double want = fill * static_cast<double>(slots * rounds) - 0.5 * static_cast<double>(big.size());
                    if (want < 1.0) continue;
                    uint64_t cb = (static_cast<uint64_t>(static_cast<double>(big_total) / want) + 4095u) & ~4095ull;
                    if (cb < kGzChunkMin) cb = kGzChunkMin;
                    // (the count is known exactly: never one chunk more than the rounds hold)
                    for (;;) {
                        uint64_t count = 0;
                        for (uint32_t i : big) count += (gz_lengths[i] + cb - 1) / cb;
                        if (count <= static_cast<uint64_t>(fill * static_cast<double>(slots * rounds)) || cb > 352u * 1024u) break;
                        cb += 4096;
                    }
                    if (cb <= 352u * 1024u) {
                        chunk_bytes = cb < kGzChunkMin ? kGzChunkMin : static_cast<uint32_t>(cb);
                        break;
                    }
                    chunk_bytes = 2 * kGzChunkMin;   // (never reached by a sane input: a thousand rounds)
                }
            }
        }
        if (chunk_bytes < 65536u) chunk_bytes = 65536u;
        for (size_t b = 0; b < big.size(); ++b) {
            const uint32_t i = big[b];
            const uint32_t nch = static_cast<uint32_t>((gz_lengths[i] + chunk_bytes - 1) / chunk_bytes);
            // room per chunk (u16 elements): twice what the file's overall ratio predicts for a chunk, at least 8x
            double ratio = static_cast<double>(out_caps[i]) / static_cast<double>(gz_lengths[i]);
            if (ratio < 4.0) ratio = 4.0;
            uint64_t cap = static_cast<uint64_t>(2.0 * ratio * chunk_bytes) + 65536;
            if (cap > (1ull << 25)) cap = 1ull << 25;
            chunk0[b] = static_cast<uint32_t>(chunks.size());
            for (uint32_t j = 0; j < nch; ++j) {
                chunks.push_back(GzChunk{gz_offsets[i], gz_lengths[i], sym_total, cap, chunk0[b], nch, chunk_bytes, 0});
                sym_total += cap;
            }
        }
        const uint32_t nc = static_cast<uint32_t>(chunks.size());
        // metadata: chunks | starts u64 | len u64 | status u32 | next u32 | isize u32
        const size_t o_chunks = 0, o_starts = o_chunks + nc * sizeof(GzChunk), o_len = o_starts + nc * 8ull,
                     o_st = o_len + nc * 8ull, o_next = o_st + nc * 4ull, o_is = o_next + nc * 4ull, o_nm = o_is + nc * 4ull,
                     o_cr = o_nm + nc * 4ull, o_rec = (o_cr + nc * 4ull + 15) / 16 * 16,
                     meta_b = o_rec + static_cast<size_t>(nc) * kGzMemRec * sizeof(uint2);
        int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzmeta), &ctx->gzmeta_cap, meta_b + 256);
        if (rc) return rc;
        rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzsym), &ctx->gzsym_cap, sym_total * 2 + 256);
        if (rc) return rc;
        uint8_t* m = ctx->d_gzmeta;
        GzChunk* d_chunks = reinterpret_cast<GzChunk*>(m + o_chunks);
        uint64_t* d_starts = reinterpret_cast<uint64_t*>(m + o_starts);
        unsigned long long* d_len = reinterpret_cast<unsigned long long*>(m + o_len);
        uint32_t* d_st = reinterpret_cast<uint32_t*>(m + o_st);
        uint32_t* d_next = reinterpret_cast<uint32_t*>(m + o_next);
        uint32_t* d_is = reinterpret_cast<uint32_t*>(m + o_is);
        uint32_t* d_nm = reinterpret_cast<uint32_t*>(m + o_nm);
        uint32_t* d_cr = reinterpret_cast<uint32_t*>(m + o_cr);
        uint2* d_rec = reinterpret_cast<uint2*>(m + o_rec);
        uint16_t* d_sym = reinterpret_cast<uint16_t*>(ctx->d_gzsym);
        VK_HIP(ctx, hipMemcpyAsync(d_chunks, chunks.data(), nc * sizeof(GzChunk), hipMemcpyHostToDevice, ctx->stream));
        if (ctx->gz_split_find) {
            hipLaunchKernelGGL(vk_gzfind_kernel, dim3(nc), dim3(64), 0, ctx->stream, gz, d_chunks, nc, d_starts);
            VK_HIP(ctx, hipGetLastError());
        } else {
            VK_HIP(ctx, hipMemsetAsync(d_starts, 0xFE, nc * 8ull, ctx->stream));   // kGzPending: the chunks' wavefronts find their own starts
        }
        hipLaunchKernelGGL(vk_gzchunk_kernel, dim3(nc), dim3(64), ctx->gz_lds_pad, ctx->stream, gz, d_sym, d_chunks, nc, d_starts, d_len,
                           d_st, d_next, d_is, d_nm, d_cr, d_rec, ctx->gz_split_find ? 0u : 1u);
        VK_HIP(ctx, hipGetLastError());
        std::vector<unsigned long long> h_len(nc);
        std::vector<uint32_t> h_st(nc), h_next(nc), h_is(nc), h_nm(nc), h_cr(nc);
        std::vector<uint2> h_rec(static_cast<size_t>(nc) * kGzMemRec);
        VK_HIP(ctx, hipMemcpyAsync(h_rec.data(), d_rec, h_rec.size() * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_len.data(), d_len, nc * 8ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_st.data(), d_st, nc * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_next.data(), d_next, nc * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_is.data(), d_is, nc * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_nm.data(), d_nm, nc * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_cr.data(), d_cr, nc * 4ull, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // follow every file's chain of chunks; a file with anything odd in it goes the direct way instead
        std::vector<GzItem> items;
        std::vector<uint32_t> first, count, okfile;
        for (size_t b = 0; b < big.size(); ++b) {
            const uint32_t i = big[b];
            const size_t mark = items.size();
            uint64_t total = 0;
            uint32_t isum = 0, nmem = 0, lastcrc = 0;
            bool ok = true;
            uint32_t c = chunk0[b];
            members.clear();
            bool members_known = true;   // every trailer's (end of text, check word) on record?
            for (;;) {
                if (h_st[c] != 0) { ok = false; break; }
                items.push_back(GzItem{chunks[c].out_off, h_len[c], out_offsets[i] + total});
                if (h_nm[c] > kGzMemRec) members_known = false;
                for (uint32_t r = 0; r < h_nm[c] && r < kGzMemRec; ++r) {
                    const uint2 rec = h_rec[static_cast<size_t>(c) * kGzMemRec + r];
                    members.push_back({total + rec.x, rec.y});
                }
                total += h_len[c];
                isum += h_is[c];
                if (h_nm[c]) {
                    nmem += h_nm[c];
                    lastcrc = h_cr[c];
                }
                const uint32_t nx = h_next[c];
                if (nx == kGzEnd) break;
                if (nx <= c || nx >= chunk0[b] + chunks[c].nchunks) { ok = false; break; }
                c = nx;
            }
            if (ok && isum != static_cast<uint32_t>(total)) ok = false;  // the members' size words do not add up to the text
            if (ok && total > out_caps[i]) {  // e.g. several members and a caller who knew only the last one's size
                status[i] = VK_GZ_OVERFLOW;
                out_lengths[i] = total;       // ... who now learns the size a second call needs
                items.resize(mark);
                continue;
            }
            if (!ok) {
                items.resize(mark);
                direct.push_back(i);
                continue;
            }
            first.push_back(static_cast<uint32_t>(mark));
            count.push_back(static_cast<uint32_t>(items.size() - mark));
            okfile.push_back(i);
            out_lengths[i] = total;
            (void)lastcrc;
            if (members_known && members.size() == nmem) add_member_checks(i, out_offsets[i], total);
        }
        if (!okfile.empty()) {
            const uint32_t ni = static_cast<uint32_t>(items.size()), nf = static_cast<uint32_t>(okfile.size());
            const size_t o_items = 0, o_first = o_items + ni * sizeof(GzItem), o_count = o_first + nf * 4ull,
                         tail_b = o_count + nf * 4ull;
            // the chunk metadata has been read back: its buffer now carries the items
            rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzmeta), &ctx->gzmeta_cap, tail_b + 256);
            if (rc) return rc;
            rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzwin), &ctx->gzwin_cap, static_cast<size_t>(ni) * 32768 + 256);
            if (rc) return rc;
            m = ctx->d_gzmeta;
            GzItem* d_items = reinterpret_cast<GzItem*>(m + o_items);
            uint32_t* d_first = reinterpret_cast<uint32_t*>(m + o_first);
            uint32_t* d_count = reinterpret_cast<uint32_t*>(m + o_count);
            VK_HIP(ctx, hipMemcpyAsync(d_items, items.data(), ni * sizeof(GzItem), hipMemcpyHostToDevice, ctx->stream));
            VK_HIP(ctx, hipMemcpyAsync(d_first, first.data(), nf * 4ull, hipMemcpyHostToDevice, ctx->stream));
            VK_HIP(ctx, hipMemcpyAsync(d_count, count.data(), nf * 4ull, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(vk_gzwin_kernel, dim3(nf), dim3(1024), 0, ctx->stream, reinterpret_cast<uint16_t*>(ctx->d_gzsym),
                               d_items, d_first, d_count, ctx->d_gzwin);
            VK_HIP(ctx, hipGetLastError());
            for (uint32_t i0 = 0; i0 < ni; i0 += 65535u) {  // (gridDim.y holds 65535 at most: 8 GiB of 128 KiB chunks)
                const uint32_t n = ni - i0 < 65535u ? ni - i0 : 65535u;
                hipLaunchKernelGGL(vk_gzfinal_kernel, dim3(32, n), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<uint16_t*>(ctx->d_gzsym), d_items + i0, n,
                                   ctx->d_gzwin + static_cast<size_t>(i0) * 32768, out);
                VK_HIP(ctx, hipGetLastError());
            }
            VK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (items / first / count are read by the kernels)
        }
    }

    // ---- small files, and large ones the chunked path gave up on: one wavefront per file -----------
    if (!direct.empty()) {
        const uint32_t nd = static_cast<uint32_t>(direct.size());
        const size_t jobs_b = static_cast<size_t>(nd) * sizeof(GzJob), len_b = static_cast<size_t>(nd) * 8,
                     st_b = static_cast<size_t>(nd) * 4;
        const size_t rec_o = (jobs_b + len_b + 3 * st_b + 15) / 16 * 16, rec_b = static_cast<size_t>(nd) * kGzMemRec * sizeof(uint2);
        int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_gzjobs), &ctx->gzjobs_cap, rec_o + rec_b);
        if (rc) return rc;
        std::vector<GzJob> jobs(nd);
        for (uint32_t j = 0; j < nd; ++j) {
            const uint32_t i = direct[j];
            jobs[j] = GzJob{gz_offsets[i], gz_lengths[i], out_offsets[i], out_caps[i]};
        }
        GzJob* d_jobs = reinterpret_cast<GzJob*>(ctx->d_gzjobs);
        unsigned long long* d_len = reinterpret_cast<unsigned long long*>(ctx->d_gzjobs + jobs_b);
        uint32_t* d_st = reinterpret_cast<uint32_t*>(ctx->d_gzjobs + jobs_b + len_b);
        uint32_t* d_nm = d_st + nd;
        uint32_t* d_cr = d_nm + nd;
        VK_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs_b, hipMemcpyHostToDevice, ctx->stream));
        uint2* d_rec = reinterpret_cast<uint2*>(ctx->d_gzjobs + rec_o);
        hipLaunchKernelGGL(vk_inflate_kernel, dim3(nd), dim3(64), 0, ctx->stream, gz, out, d_jobs, nd, d_len, d_st, d_nm, d_cr, d_rec);
        VK_HIP(ctx, hipGetLastError());
        std::vector<unsigned long long> h_len(nd);
        std::vector<uint32_t> h_st(3 * static_cast<size_t>(nd));
        std::vector<uint2> h_rec(static_cast<size_t>(nd) * kGzMemRec);
        VK_HIP(ctx, hipMemcpyAsync(h_rec.data(), d_rec, rec_b, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_len.data(), d_len, len_b, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipMemcpyAsync(h_st.data(), d_st, 3 * st_b, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (also keeps `jobs` alive until the copy has read it)
        for (uint32_t j = 0; j < nd; ++j) {
            const uint32_t i = direct[j];
            out_lengths[i] = h_len[j];
            status[i] = h_st[j];
            const uint32_t nm = h_st[nd + j];
            if (h_st[j] == 0 && nm >= 1 && nm <= kGzMemRec && h_len[j] < (1ull << 32)) {
                members.clear();
                for (uint32_t r = 0; r < nm; ++r) members.push_back({h_rec[static_cast<size_t>(j) * kGzMemRec + r].x, h_rec[static_cast<size_t>(j) * kGzMemRec + r].y});
                add_member_checks(i, out_offsets[i], h_len[j]);
            }
        }
    }
    // the check word of every member against the CRC-32 of the text it inflated to
    if (!crc_jobs.empty()) {
        std::vector<uint32_t> got;
        int rc = gz_text_crc(ctx, out, crc_jobs, got);
        if (rc) return rc;
        for (size_t j = 0; j < crc_jobs.size(); ++j)
            if (got[j] != crc_want[j]) status[crc_file[j]] |= VK_GZ_BAD_CRC;
    }
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

// dist 2 (vk_aux.h, "reads shaped like what step B hands to step D"): the record offsets of `n` samples in the
// context's workspace, n * (reads + 1) u32
static int synth_shapes(vk_ctx* ctx, uint32_t sample0, uint32_t n, uint32_t reads, uint32_t readlen, uint64_t seed) {
    int rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_synth), &ctx->synth_cap, static_cast<size_t>(n) * (reads + 1ull) * sizeof(uint32_t));
    if (rc) return rc;
    hipLaunchKernelGGL(vk_synth_shape_kernel, dim3(n), dim3(1024), 0, ctx->stream, ctx->d_synth, sample0, reads, readlen, seed);
    VK_HIP(ctx, hipGetLastError());
    return VK_OK;
}

int vk_synth_shaped_lengths(vk_ctx* ctx, uint32_t sample0, uint32_t nsamples, uint32_t reads, uint32_t readlen,
                            uint64_t seed, uint64_t* lengths) {
    if (!ctx || !lengths || readlen < 64 || readlen > 1000 || reads == 0 || reads > 4000000u) return VK_EINVAL;
    if (static_cast<uint64_t>(reads) * (74ull + 4ull * readlen) >= (1ull << 32)) return VK_EINVAL;   // record offsets are u32
    VK_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t kSlab = 64;
    std::vector<uint32_t> tot(kSlab);
    for (uint32_t s0 = 0; s0 < nsamples; s0 += kSlab) {
        const uint32_t n = nsamples - s0 < kSlab ? nsamples - s0 : kSlab;
        int rc = synth_shapes(ctx, sample0 + s0, n, reads, readlen, seed);
        if (rc) return rc;
        VK_HIP(ctx, hipMemcpy2DAsync(tot.data(), sizeof(uint32_t), ctx->d_synth + reads, (reads + 1ull) * sizeof(uint32_t),
                                     sizeof(uint32_t), n, hipMemcpyDeviceToHost, ctx->stream));
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < n; ++i) lengths[s0 + i] = tot[i];
    }
    return VK_OK;
}

int vk_synth_shaped_device(vk_ctx* ctx, void* d_out, const uint64_t* offsets, uint32_t sample0, uint32_t nsamples,
                           uint32_t reads, uint32_t readlen, uint64_t seed) {
    if (!ctx || !d_out || !offsets || readlen < 64 || readlen > 1000 || reads == 0 || reads > 4000000u) return VK_EINVAL;
    if (static_cast<uint64_t>(reads) * (74ull + 4ull * readlen) >= (1ull << 32)) return VK_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_out) & 15u) != 0) return VK_EINVAL;
    for (uint32_t i = 0; i < nsamples; ++i)
        if (offsets[i] & 15u) return VK_EINVAL;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t kSlab = 64;
    const uint32_t chunks = (reads + 255u) / 256u;
    for (uint32_t s0 = 0; s0 < nsamples; s0 += kSlab) {
        const uint32_t n = nsamples - s0 < kSlab ? nsamples - s0 : kSlab;
        int rc = synth_shapes(ctx, sample0 + s0, n, reads, readlen, seed);
        if (rc) return rc;
        rc = ensure(ctx, reinterpret_cast<void**>(&ctx->d_synth_offs), &ctx->synth_offs_cap, kSlab * sizeof(uint64_t));
        if (rc) return rc;
        // (pageable source: the copy has read it on return)
        VK_HIP(ctx, hipMemcpyAsync(ctx->d_synth_offs, offsets + s0, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(vk_synth_shaped_kernel, dim3(n * chunks), dim3(256), 0, ctx->stream, static_cast<uint8_t*>(d_out),
                           ctx->d_synth_offs, ctx->d_synth, sample0 + s0, reads, readlen, seed);
        VK_HIP(ctx, hipGetLastError());
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the workspaces are reused by the next slab
    }
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

#ifdef VK_GZ_STAMPS
int vk_debug_read_gz_find(unsigned long long* out, uint32_t nchunks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gz_find), static_cast<size_t>(nchunks) * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
int vk_debug_read_gz_res(unsigned long long* out, uint32_t nchunks, int clear) {
    if (clear) {
        static unsigned long long zeros[16384 * 4];
        return hipMemcpyToSymbol(HIP_SYMBOL(g_gz_res), zeros, sizeof(zeros)) == hipSuccess ? 0 : 2;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gz_res), static_cast<size_t>(nchunks) * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
int vk_debug_read_gz_stamps(unsigned long long* out, uint32_t nchunks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gz_stamps), static_cast<size_t>(nchunks) * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#endif

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

int vk_last_count_general(vk_ctx* ctx, uint64_t* general_pieces, uint64_t* pieces) {
    if (!ctx || !general_pieces || !pieces) return VK_EINVAL;
    *general_pieces = 0;
    *pieces = 0;
    if (ctx->last_waves == 0) return VK_OK;
    VK_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> w(ctx->last_waves);
    VK_HIP(ctx, hipMemcpyAsync(w.data(), ctx->d_wavephase, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t g = 0, active = 0;
    for (uint32_t v : w) {
        if (v & 0x80u) continue;
        ++active;
        g += v >> 8;
    }
    *general_pieces = g;
    // every wave's range starts one 64-byte block early and ends with a partial piece: about one piece per wave on top
    *pieces = (ctx->last_bytes + kPiece - 1) / kPiece + active;
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
