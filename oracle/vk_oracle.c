/*
 * vk_oracle.c -- CPU restatement of varKoder's `image` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.
 *
 * Parity status
 *   - image stage (vko_image, vko_cgr_lut, strand merge): PINNED against the
 *     reference's own make_image()/get_cgr()/get_kmer_mapping() run in the
 *     build container (oracle/gen_golden.py -> tests/golden/).
 *   - counting stage (vko_count_fastq): "parity unpinned".  The reference
 *     shells out to GATB dsk 2.3.3 (conda_environments/linux.yml:11), whose
 *     source is not in /root/reference and whose binary is absent; the
 *     semantics below restate dsk's documented behaviour and
 *     are anchored on first-principles known-answer tests only.
 *
 * Reference lines followed (all relative to /root/reference/varKoder):
 *   commands/image.py:771-790   dsk argv: -kmer-size k -abundance-min 1
 *                               => every canonical class with count >= 1
 *   commands/image.py:897-903   join with mapping, groupby(x,y).mean, fillna(0)
 *                               => each pixel of s or rc(s) gets c(s)+c(rc s)
 *   commands/image.py:906-913   arr[x,y] = count+1; transpose; flip(0)
 *                               => img[side-1-y, x] = count+1
 *   commands/image.py:916-919   np.quantile(arr, arange(0,1,1/256)) (linear),
 *                               np.digitize(arr, bins) - 1, uint8
 *   core/utils.py:174-217       get_cgr corners A(0,0) C(0,1) G(1,1) T(1,0)
 *
 * Code convention used across the repo (oracle, C-ABI, HIP kernels, LUT files):
 *   base codes A0 C1 G2 T3; code(s) = sum_i b_i * 4^(k-1-i), i.e. the k-mer
 *   read as a base-4 number, first base most significant.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VKO_OK 0
#define VKO_EINVAL 1
#define VKO_EFORMAT 2
#define VKO_ENOMEM 3

/* 0..3 for ACGTacgt, -1 for every other byte (a table: this loop is also the CPU baseline) */
static const signed char kBaseCode[256] = {
    ['A'] = 1, ['a'] = 1, ['C'] = 2, ['c'] = 2, ['G'] = 3, ['g'] = 3, ['T'] = 4, ['t'] = 4,
};
static inline int base_code(uint8_t b) { return kBaseCode[b] - 1; }

uint32_t vko_revcomp(uint32_t code, int k) {
    uint32_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3u - (code & 3u));
        code >>= 2;
    }
    return r;
}

/* Count forward-strand k-mers of every read of a 4-line FASTQ.
 * dsk semantics: per-read windows; reads shorter than k give
 * nothing; a window containing any byte outside ACGTacgt is skipped.
 * fwd[4^k] is ADDED to (caller zeroes).  nwin receives the number of windows
 * counted.  Records: header '@...', sequence, '+...', quality; line ends are
 * '\n' (a preceding '\r' is an ordinary non-ACGT byte); the last line may
 * lack its newline. */
int vko_count_fastq(const uint8_t* buf, size_t n, int k, uint32_t* fwd, uint64_t* nwin) {
    if (k < 1 || k > 15 || !fwd) return VKO_EINVAL;
    const uint32_t mask = (k == 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    uint64_t windows = 0;
    size_t pos = 0;
    unsigned line = 0;
    int status = VKO_OK;
    while (pos < n) {
        const uint8_t* nl = memchr(buf + pos, '\n', n - pos);
        size_t e = nl ? (size_t)(nl - buf) : n;
        unsigned ph = line & 3u;
        if (ph == 0) {
            if (buf[pos] != '@') status = VKO_EFORMAT;
        } else if (ph == 2) {
            if (e == pos || buf[pos] != '+') status = VKO_EFORMAT;
        } else if (ph == 1) {
            uint32_t fw = 0;
            int run = 0;
            for (size_t i = pos; i < e; i++) {
                int c = base_code(buf[i]);
                if (c < 0) { run = 0; continue; }
                fw = ((fw << 2) | (uint32_t)c) & mask;
                if (++run >= k) { fwd[fw]++; windows++; }
            }
        }
        line++;
        pos = e + 1;
    }
    /* a well-formed file ends after a quality line (with or without '\n') */
    if (n > 0 && (line & 3u) != 0) status = VKO_EFORMAT;
    if (nwin) *nwin = windows;
    return status;
}

/* Read subsampling (the build's stand-in for `reformat.sh samplebasestarget`, which the reference
 * runs before dsk, commands/image.py:577-630; BBTools' sampler itself is not reproducible here, so
 * this is the build's own rule, restated independently of the kernels): a read is taken iff
 * hash32(seed, offset of the newline that ends its header line) < threshold, threshold in [0, 2^32]. */
static uint32_t vko_sample_hash(uint64_t seed, uint64_t anchor) {
    uint32_t h = (uint32_t)anchor ^ (uint32_t)seed;
    h += (uint32_t)(anchor >> 32) * 0x9E3779B1u + (uint32_t)(seed >> 32);
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* vko_count_fastq over the taken reads only.  sites[0] = bytes of all sequence lines (the
 * reference's nsites, image.py:669-675: len(line) - 1), sites[1] = of the taken reads' lines.
 * The reference draws its subsamples with `reformat.sh ... breaklength=500` (image.py:586-588):
 * BBTools cuts a read longer than 500 bases into pieces of 500 before sampling, so no k-mer of a
 * subsample spans a multiple of 500 bases of its read -- restated here as a window reset at every
 * such position (VKO_BREAK_LENGTH). */
#define VKO_BREAK_LENGTH 500
int vko_count_fastq_sampled(const uint8_t* buf, size_t n, int k, uint64_t seed, uint64_t threshold,
                            uint32_t* fwd, uint64_t* nwin, uint64_t* sites) {
    if (k < 1 || k > 15 || !fwd || threshold > (1ull << 32)) return VKO_EINVAL;
    const uint32_t mask = (1u << (2 * k)) - 1u;
    uint64_t windows = 0, all = 0, taken = 0;
    size_t pos = 0;
    unsigned line = 0;
    int status = VKO_OK, take = 0;
    while (pos < n) {
        const uint8_t* nl = memchr(buf + pos, '\n', n - pos);
        size_t e = nl ? (size_t)(nl - buf) : n;
        unsigned ph = line & 3u;
        if (ph == 0) {
            if (buf[pos] != '@') status = VKO_EFORMAT;
            take = (e < n) && ((uint64_t)vko_sample_hash(seed, (uint64_t)e) < threshold);
        } else if (ph == 2) {
            if (e == pos || buf[pos] != '+') status = VKO_EFORMAT;
        } else if (ph == 1) {
            all += e - pos;
            if (take) {
                taken += e - pos;
                uint32_t fw = 0;
                int run = 0;
                for (size_t i = pos; i < e; i++) {
                    int c = base_code(buf[i]);
                    if ((i - pos) % VKO_BREAK_LENGTH == 0) run = 0;
                    if (c < 0) { run = 0; continue; }
                    fw = ((fw << 2) | (uint32_t)c) & mask;
                    if (++run >= k) { fwd[fw]++; windows++; }
                }
            }
        }
        line++;
        pos = e + 1;
    }
    if (n > 0 && (line & 3u) != 0) status = VKO_EFORMAT;
    if (nwin) *nwin = windows;
    if (sites) { sites[0] = all; sites[1] = taken; }
    return status;
}

/* Strand merge: tot[c] = number of windows whose canonical class is {c, rc(c)}.
 * This is what reaches a pixel after image.py:900-903 whichever spelling dsk
 * printed.  Palindromes (even k) are counted once. */
void vko_strand_merge(const uint32_t* fwd, int k, uint32_t* tot) {
    uint32_t n = 1u << (2 * k);
    for (uint32_t c = 0; c < n; c++) {
        uint32_t r = vko_revcomp(c, k);
        tot[c] = (r == c) ? fwd[c] : fwd[c] + fwd[r];
    }
}

/* Closed form of get_cgr (core/utils.py:174-217): with i = position in the
 * k-mer (0 = first base), x = sum ((b_i>>1)&1) 2^i, y = sum (((b_i>>1)^b_i)&1) 2^i,
 * side = 2^k, image pixel index = (side-1-y)*side + x  (image.py:910-913). */
void vko_cgr_lut(int k, uint32_t* pix) {
    uint32_t n = 1u << (2 * k), side = 1u << k;
    for (uint32_t c = 0; c < n; c++) {
        uint32_t x = 0, y = 0;
        for (int i = 0; i < k; i++) {
            uint32_t b = (c >> (2 * (k - 1 - i))) & 3u;
            x |= ((b >> 1) & 1u) << i;
            y |= (((b >> 1) ^ b) & 1u) << i;
        }
        pix[c] = (side - 1u - y) * side + x;
    }
}

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

/* Image stage.  tot[4^k] strand-merged counts, pix[4^k] pixel index per code,
 * npix = side*side, img[npix] out.
 * Integer restatement of np.quantile(linear)+np.digitize:
 *   a = sort(val);  pos_j = j*(n-1); i = pos_j >> 8; g = pos_j & 255
 *   B_j = 256*a[i] + (a[min(i+1,n-1)] - a[i])*g        (= 256 * bin_j, exact)
 *   img = #{ j : B_j <= 256*val } - 1 */
int vko_image(const uint32_t* tot, int k, const uint32_t* pix, uint32_t npix, uint8_t* img) {
    uint32_t ncode = 1u << (2 * k);
    uint32_t* val = (uint32_t*)calloc(npix, sizeof(uint32_t));
    uint32_t* srt = (uint32_t*)malloc((size_t)npix * sizeof(uint32_t));
    if (!val || !srt) { free(val); free(srt); return VKO_ENOMEM; }
    for (uint32_t c = 0; c < ncode; c++) {
        if (pix[c] >= npix) { free(val); free(srt); return VKO_EINVAL; }
        val[pix[c]] = tot[c] + 1u;
    }
    memcpy(srt, val, (size_t)npix * sizeof(uint32_t));
    qsort(srt, npix, sizeof(uint32_t), cmp_u32);
    uint64_t B[256];
    for (uint32_t j = 0; j < 256; j++) {
        uint64_t pos = (uint64_t)j * (npix - 1u);
        uint32_t i = (uint32_t)(pos >> 8), g = (uint32_t)(pos & 255u);
        uint32_t i1 = (i + 1u < npix) ? i + 1u : npix - 1u;
        B[j] = 256ull * srt[i] + (uint64_t)(srt[i1] - srt[i]) * g;
    }
    for (uint32_t p = 0; p < npix; p++) {
        uint64_t v = 256ull * val[p];
        /* upper_bound over the non-decreasing B */
        uint32_t lo = 0, hi = 256;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (B[mid] <= v) lo = mid + 1; else hi = mid;
        }
        img[p] = (uint8_t)(lo - 1u);   /* B[0] = 256*min <= v always, so lo >= 1 */
    }
    free(val); free(srt);
    return VKO_OK;
}

/* Whole path for one sample, used as the "port" CPU baseline in bench.py. */
int vko_fastq_to_image(const uint8_t* buf, size_t n, int k, const uint32_t* pix,
                       uint32_t npix, uint32_t* fwd_scratch, uint32_t* tot_scratch,
                       uint8_t* img, uint64_t* nwin) {
    uint32_t ncode = 1u << (2 * k);
    memset(fwd_scratch, 0, (size_t)ncode * sizeof(uint32_t));
    int st = vko_count_fastq(buf, n, k, fwd_scratch, nwin);
    if (st != VKO_OK) return st;
    vko_strand_merge(fwd_scratch, k, tot_scratch);
    return vko_image(tot_scratch, k, pix, npix, img);
}
