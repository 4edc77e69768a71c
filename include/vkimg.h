/*
 * vkimg.h -- C ABI of libvkimg_hip.so: varKoder's `image` hot path on MI355X (gfx950).
 *
 * The reference (brunoasm/varKoder) has no FFI layer: the path sits behind two
 * Python functions that shell out to `dsk` / `dsk2ascii`.  Each
 * entry point below names the reference interface it replaces.  Plain pointers
 * and sizes only; status codes, never exceptions; caller-allocated buffers.
 *
 * Conventions
 *   - k in [5, 9]  (varKoder/commands/image.py:1209-1210).
 *   - k-mer code: bases A0 C1 G2 T3, code = sum_i b_i * 4^(k-1-i) (first base
 *     most significant), so codes sort like the k-mer strings.
 *   - hist[4^k] (u32): FORWARD-strand window counts.  The canonical class count
 *     dsk reports for {s, rc(s)} is hist[s] + hist[rc(s)] (hist[s] for a
 *     palindrome); vk_image_* performs that merge.
 *   - pix[4^k] (u32): image pixel index (row-major, row 0 on top) of every code,
 *     i.e. (side-1-y)*side + x of the reference's mapping table
 *     (core/utils.py:152-217, image.py:906-913).
 *   - "d_" pointers are device (HBM) addresses of the context's GPU; the others
 *     are host addresses.  FASTQ samples on the device must start at 16-byte
 *     aligned addresses and the buffer must be readable up to the 16-byte
 *     rounded end of the last sample.
 *   - All work is enqueued on the context's stream; *_host calls synchronise
 *     before returning, *_device calls do not.
 *   - A context owns device workspaces that grow on demand and is NOT re-entrant: use it
 *     from one thread at a time (one context per worker thread / process is the intended
 *     model: the reference runs one sample per pool worker, image.py:1281-1284).
 *   - k = 8, 9 count through a bucketed two-pass path whose workspace is about half the
 *     FASTQ bytes of the samples in flight (subsampled counts: about the FASTQ bytes); large
 *     batches are processed in sub-batches.
 */
#ifndef VKIMG_H
#define VKIMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_OK 0
#define VK_EINVAL 1      /* bad argument (k, null pointer, alignment) */
#define VK_EHIP 2        /* a HIP runtime call failed; see vk_last_hip_error */
#define VK_ENOMAP 3      /* vk_set_mapping not called for this k */
#define VK_EFORMAT 4     /* FASTQ framing inconsistent (per-sample status word) */
#define VK_ENOMEM 5

/* per-file status bits of vk_inflate_device */
#define VK_GZ_BAD_HEADER 1u    /* not a gzip member (magic, method, reserved flags) */
#define VK_GZ_BAD_DATA 2u      /* invalid DEFLATE data (block type, code set, distance too far back) */
#define VK_GZ_TRUNCATED 4u     /* the stream ends before its last block / trailer */
#define VK_GZ_OVERFLOW 8u      /* the text does not fit out_caps[i]: call again with more room */
#define VK_GZ_BAD_SIZE 16u     /* a member's ISIZE differs from the bytes it inflated to */
#define VK_GZ_BAD_CRC 32u      /* single-member file: the trailer's CRC-32 is not the CRC-32 of the inflated text */

/* per-sample status bits written by the count stage */
#define VK_ST_BAD_START 1u     /* first record malformed: no leading '@', or third line not '+' */
#define VK_ST_BAD_PHASE 2u     /* line count mod 4 inconsistent between byte ranges / at EOF */

typedef struct vk_ctx vk_ctx;

int vk_abi_version(void);
const char* vk_strerror(int status);
/* hipGetErrorString of the last failing HIP call on this context ("" if none) */
const char* vk_last_hip_error(const vk_ctx* ctx);

/* Context: one per process per GPU.  own_stream != 0: the library creates its own
 * non-blocking stream (`stream` ignored); own_stream == 0: all work is enqueued on the
 * caller's hipStream_t `stream` (NULL = the device's default stream), so that it
 * orders with the caller's other work, e.g. a PyTorch stream. */
int vk_ctx_create(int device, void* stream, int own_stream, vk_ctx** out);
void vk_ctx_destroy(vk_ctx* ctx);
int vk_ctx_sync(vk_ctx* ctx);

/* Replaces get_kmer_mapping(k, method) (core/utils.py:152-171): installs the
 * code->pixel table for k.  pix == NULL selects the CGR closed form of
 * get_cgr (core/utils.py:174-217; npix must be 4^k); otherwise pix[4^k] is a
 * host table (varKode LUT) with every entry < npix. */
int vk_set_mapping(vk_ctx* ctx, int k, const uint32_t* pix, uint32_t npix);

/* Replaces count_kmers() = `dsk -kmer-size k -abundance-min 1 -file IN`
 * (commands/image.py:727-806, argv :771-790) for a batch of samples resident in
 * HBM.  Sample i is the FASTQ text d_fastq[offsets[i] .. offsets[i]+lengths[i]).
 * d_hist[nsamples][4^k] receives forward-strand counts, d_status[nsamples] the
 * VK_ST_* bits.  parts_per_sample = 0 lets the library choose how many
 * workgroups split one sample. */
int vk_count_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets,
                    const uint64_t* lengths, uint32_t nsamples, int k,
                    uint32_t parts_per_sample, uint32_t* d_hist, uint32_t* d_status);

/* vk_count_device restricted to a pseudo-random subset of each sample's reads: stands where the
 * reference runs `reformat.sh samplebasestarget=N sampleseed=S` once per output size before dsk
 * (split_fastq / run_parallel_reformats, commands/image.py:577-725).  Not bit-compatible with
 * BBTools' sampler (statistical equivalent, opt-in): read r of sample i is counted iff
 * hash32(seeds[i], offset of the newline ending r's header line) < thresholds[i], thresholds in
 * [0, 2^32] (2^32 = every read) -- a pure function of the file's bytes, independent of how the
 * library splits the sample.  seeds/thresholds are host arrays.  d_sites[nsamples][2] (device,
 * may be NULL) receives the bytes of all sequence lines (the reference's `nsites`, :669-675) and
 * of the sequence lines of the reads taken. */
int vk_count_sampled_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets,
                            const uint64_t* lengths, uint32_t nsamples, int k,
                            uint32_t parts_per_sample, const uint64_t* seeds,
                            const uint64_t* thresholds, uint32_t* d_hist, uint32_t* d_status,
                            uint64_t* d_sites);

/* The read index of a batch of samples resident in HBM: one streaming pass per sample that lists every read's
 * anchor (the newline that ends its header line) and adds up the bytes of all sequence lines -- `nsites`, which
 * split_fastq computes before it derives its ladder of subsample sizes (commands/image.py:663-675).  sites[i]
 * and status[i] (VK_ST_* bits) are HOST arrays; the call synchronises.  The context keeps the index until the
 * next call: vk_count_sampled_device calls whose samples (same d_fastq, same offsets and lengths, in any order
 * and any number of times) are all in it then WALK the reads each subsample takes instead of streaming the text
 * once per subsample (the reference runs reformat.sh + dsk once per subsample, :577-627, :682-695) -- same
 * counts, same sites.  A sample with more than one read per 32 bytes of text, or of 4 GiB and more, gets no index
 * (such calls stream as before).  VKIMG_NO_READ_INDEX=1 disables the walker. */
int vk_read_index_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                         uint32_t nsamples, uint32_t parts_per_sample, uint64_t* sites, uint32_t* status);

/* vk_count_device and vk_read_index_device in ONE pass over the text (k <= 7: the count kernel lists the anchors and
 * adds up the sites on its way; k = 8, 9: the two passes, one after the other): what a ladder whose first step takes
 * every read wants (split_fastq when the file holds less than --max-bp, commands/image.py:677-680).  d_hist /
 * d_status as vk_count_device; sites / status (host) as vk_read_index_device; the call synchronises. */
int vk_count_index_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets, const uint64_t* lengths,
                          uint32_t nsamples, int k, uint32_t parts_per_sample, uint32_t* d_hist, uint32_t* d_status,
                          uint64_t* sites, uint32_t* status);

/* Replaces make_image()'s arithmetic = `dsk2ascii` dump + join/groupby +
 * count+1 scatter + 256-quantile rank binning (commands/image.py:864-919) for a
 * batch of histograms.  d_img[nsamples][npix] receives the uint8 pixels. */
int vk_image_device(vk_ctx* ctx, const uint32_t* d_hist, uint32_t nsamples, int k,
                    uint8_t* d_img);

/* Both stages back to back (run_clean2img steps D+E, commands/image.py:1054-1127). */
int vk_fastq_to_image_device(vk_ctx* ctx, const void* d_fastq, const uint64_t* offsets,
                             const uint64_t* lengths, uint32_t nsamples, int k,
                             uint32_t parts_per_sample, uint32_t* d_hist,
                             uint32_t* d_status, uint8_t* d_img);

/* Replaces the gzip reader inside dsk (`-file IN.fq.gz`, commands/image.py:771-790; the files are
 * written by split_fastq, :696-708): inflates nfiles gzip files,
 * d_gz[gz_offsets[i] .. +gz_lengths[i]), into d_out[out_offsets[i] ..), at most out_caps[i] bytes each.
 * d_gz is memory the device can read: HBM, or pinned host memory (hipHostMalloc) -- the kernels then read the
 * compressed bytes over PCIe where they lie (twice: block-start finder and decoder) and no copy is needed
 * (single-member files: the little-endian u32 in the file's last four bytes is the text length).
 * Multi-member files are accepted; after a complete member, zero padding or bytes that are not a gzip
 * header end the data (as for zlib's gzread, which is how dsk reads a .gz).  out_lengths[i] (host)
 * receives the bytes written and status[i] (host) the VK_GZ_* bits; the call synchronises.  A file whose
 * text does not fit out_caps[i] gets VK_GZ_OVERFLOW and no text; where the whole file was decoded before the
 * slot was looked at (files of 512 KiB and more) out_lengths[i] then holds the size a second call needs
 * (> out_caps[i]), otherwise the bytes that fitted.
 * Integrity: structure and every member's ISIZE are checked, and every member's CRC-32 word is verified (on the
 * GPU) against the stretch of text the member inflated to (VK_GZ_BAD_CRC) -- as zlib's gzread, which dsk reads
 * through, does; only a file with more than 62 members inside one 128 KiB stretch of compressed bytes goes
 * unchecked. */
int vk_inflate_device(vk_ctx* ctx, const void* d_gz, const uint64_t* gz_offsets, const uint64_t* gz_lengths,
                      uint32_t nfiles, void* d_out, const uint64_t* out_offsets, const uint64_t* out_caps,
                      uint64_t* out_lengths, uint32_t* status);

/* Host-buffer conveniences (one sample): H2D copy, kernels, D2H copy, sync.
 * vk_count_host returns VK_EFORMAT when the sample's status word is non-zero
 * (hist is still written). */
int vk_count_host(vk_ctx* ctx, const uint8_t* fastq, size_t nbytes, int k, uint32_t* hist,
                  uint32_t* status);
int vk_image_host(vk_ctx* ctx, const uint32_t* hist, int k, uint8_t* img);

/* Synthetic FASTQ of BASELINE.md section 4, written straight into HBM: samples
 * sample0 .. sample0+nsamples-1, `reads` reads of `readlen` bases each, record =
 * "@sSSSSS.RRRRRRR\n" + bases + "\n+\n" + 'I'*readlen + "\n" (2*readlen+20 bytes;
 * 320 at readlen 150).  dist 0 = uniform ACGT, 1 = GC-skewed + homopolymer
 * reads; N injected at ~1e-3 per base.  Sample j starts at j*reads*(2*readlen+20).
 * varkoder_amd/synth.py is the bit-identical host generator. */
int vk_synth_fastq_device(vk_ctx* ctx, void* d_out, uint32_t sample0, uint32_t nsamples,
                          uint32_t reads, uint32_t readlen, uint64_t seed, int dist);

/* Synthetic FASTQ shaped like the files step B of the reference hands to step D: fastp runs with --merge
 * --include_unmerged and --disable_length_filtering (commands/image.py:405,426-427,494-495), so a cleaned file
 * holds reads of every length from 0 to about 2 x readlen under long headers.  Per read: 65 % readlen bases,
 * 20 % merged pairs (readlen+1 .. 2*readlen-10), 10 % trimmed (45 .. readlen-1), 5 % of 0 .. 44 bases (empty
 * reads included); header lines of 40 .. 70 bytes; quality characters '!' .. 'I' ('@' and '+' among them).
 * Samples differ in size: vk_synth_shaped_lengths fills lengths[nsamples] (host; synchronises), the caller lays
 * the samples out at 16-byte aligned offsets[] and vk_synth_shaped_device writes them (bytes up to each
 * sample's 16-byte rounded end are zeroed; synchronises).  64 <= readlen <= 1000.
 * varkoder_amd/synth.py (dist=2) is the bit-identical host generator. */
int vk_synth_shaped_lengths(vk_ctx* ctx, uint32_t sample0, uint32_t nsamples, uint32_t reads, uint32_t readlen,
                            uint64_t seed, uint64_t* lengths);
int vk_synth_shaped_device(vk_ctx* ctx, void* d_out, const uint64_t* offsets, uint32_t sample0, uint32_t nsamples,
                           uint32_t reads, uint32_t readlen, uint64_t seed);

/* Replaces remap() of `varKoder convert` (commands/convert.py:34-77) for a batch of host
 * images: out[p] = in[src0[p]] (src 0xFFFFFFFF = pixel without a k-mer -> 0); with sum_rc the
 * uint8-wrapping sum w0[p]*in[src0[p]] + w1[p]*in[src1[p]] followed by the reference's
 * (v - min) / max * 255 rescale in float64.  The source maps are built by
 * varkoder_amd/convert.py from the two k-mer mappings. */
int vk_remap_host(vk_ctx* ctx, const uint8_t* img_in, uint32_t nimg, uint32_t npix_in,
                  uint32_t npix_out, const uint32_t* src0, const uint32_t* src1,
                  const uint8_t* w0, const uint8_t* w1, int sum_rc, uint8_t* img_out);

/* Input side of `varKoder query` (commands/query.py:283-324 with the item transform of
 * commands/train.py:236-245): uint8 images d_img[nimg][side*side] (device) -> float32
 * d_out[nimg][3][out][out] = ((PIL BOX-resampled pixel)/255 - mean)/std, grey replicated to three
 * channels.  bounds[out][2] = (first source index, count) and coef[out][kmax] = PIL's 22-bit
 * fixed-point BOX coefficients of one axis (host tables, built by varkoder_amd/query.py). */
int vk_preprocess_device(vk_ctx* ctx, const uint8_t* d_img, uint32_t nimg, uint32_t side,
                         uint32_t out, const int32_t* bounds, const int32_t* coef, uint32_t kmax,
                         float mean, float stdv, float* d_out);

/* Replaces: dsk opening and reading `-file <sample>.fq` (commands/image.py:771-796), for plain-text files: nfiles
 * host regions -- each a page-aligned, read-only MAP_SHARED mapping of a whole file -- are copied to
 * d_dst + dst_offsets[i] by DMA from the page-cache pages where they lie: no read() into a staging buffer (a copy
 * that costs a core per ~3 GB/s; ranks that share a host's cores cannot feed a 57 GB/s link with it).
 * registered (may be NULL): registered[i] != 0 says the caller has pinned region i with vk_host_register -- ahead of
 * time, on another thread -- and will release it; any other region is registered for the duration of its copy
 * (file i + 1 while file i is in flight) and released before the call returns.  The call synchronises.
 * status[i] (host): 0 copied, 1 the pages could not be registered (the caller copies that file through its own
 * buffer), 2 the copy failed.  Regions of 0 bytes are skipped. */
int vk_upload_mapped(vk_ctx* ctx, void* d_dst, const uint64_t* dst_offsets, const void* const* h_src,
                     const uint64_t* nbytes, const uint8_t* registered, uint32_t nfiles, uint32_t* status);

/* Pin / release a host region for vk_upload_mapped (hipHostRegister / hipHostUnregister: the pages stay where they
 * are, in the page cache).  Thread-safe; VK_EHIP if the platform refuses (the caller then leaves the flag 0). */
int vk_host_register(vk_ctx* ctx, const void* p, uint64_t nbytes);
int vk_host_unregister(vk_ctx* ctx, const void* p);

/* Introspection used by bench.py / tests: workgroups and LDS bytes of the last
 * vk_count_device launch. */
int vk_last_count_launch(const vk_ctx* ctx, uint32_t* grid, uint32_t* block, uint32_t* lds_bytes);

/* Introspection (bench.py): of the last vk_count_device call with k <= 7, how many 4 KiB pieces of text left the
 * sequence-only fast path for the general one (reads under ~45 bases, non-ASCII bytes, low complexity, the first
 * and last piece of every wavefront's range), and about how many pieces there were.  Synchronises. */
int vk_last_count_general(vk_ctx* ctx, uint64_t* general_pieces, uint64_t* pieces);

#ifdef __cplusplus
}
#endif
#endif /* VKIMG_H */
