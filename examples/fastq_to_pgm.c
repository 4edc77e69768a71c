/*
 * fastq_to_pgm.c -- the C ABI of include/vkimg.h used from plain C, no Python, no PyTorch:
 * one FASTQ file in, one k-mer image out (binary PGM), with the CGR mapping.
 *
 *   gcc -O2 -Iinclude examples/fastq_to_pgm.c -o fastq_to_pgm \
 *       -Lvarkoder_amd -l:libvkimg_hip.so -Wl,-rpath,$PWD/varkoder_amd
 *   ./fastq_to_pgm reads.fq 7 out.pgm
 *
 * This is the whole of steps D+E of the reference's run_clean2img (commands/image.py:1054-1127)
 * for one sample; PNG + text chunks are left to the caller.
 */
#include <stdio.h>
#include <stdlib.h>

#include "vkimg.h"

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s reads.fq k out.pgm\n", argv[0]);
        return 2;
    }
    const int k = atoi(argv[2]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* text = (uint8_t*)malloc((size_t)n + 1);
    if (!text || fread(text, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "read failed\n"); return 2; }
    fclose(f);

    vk_ctx* ctx = NULL;
    int rc = vk_ctx_create(0, NULL, 1, &ctx);                 /* device 0, library-owned stream */
    if (rc) { fprintf(stderr, "vk_ctx_create: %s\n", vk_strerror(rc)); return 1; }
    const uint32_t side = 1u << k, npix = side * side;
    rc = vk_set_mapping(ctx, k, NULL, npix);                  /* NULL = CGR closed form */
    if (rc) { fprintf(stderr, "vk_set_mapping: %s\n", vk_strerror(rc)); return 1; }

    uint32_t* hist = (uint32_t*)malloc(sizeof(uint32_t) << (2 * k));
    uint8_t* img = (uint8_t*)malloc(npix);
    uint32_t status = 0;
    rc = vk_count_host(ctx, text, (size_t)n, k, hist, &status);   /* replaces `dsk` */
    if (rc) { fprintf(stderr, "vk_count_host: %s (status bits %u) %s\n", vk_strerror(rc), status, vk_last_hip_error(ctx)); return 1; }
    rc = vk_image_host(ctx, hist, k, img);                        /* replaces dsk2ascii + pandas/NumPy */
    if (rc) { fprintf(stderr, "vk_image_host: %s\n", vk_strerror(rc)); return 1; }

    unsigned long long windows = 0;
    for (uint32_t c = 0; c < (1u << (2 * k)); ++c) windows += hist[c];
    FILE* o = fopen(argv[3], "wb");
    if (!o) { perror(argv[3]); return 2; }
    fprintf(o, "P5\n%u %u\n255\n", side, side);
    fwrite(img, 1, npix, o);
    fclose(o);
    printf("%s: %ld bytes, %llu k-mer windows, %ux%u image -> %s\n", argv[1], n, windows, side, side, argv[3]);
    vk_ctx_destroy(ctx);
    free(text); free(hist); free(img);
    return 0;
}
