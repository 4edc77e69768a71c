// Micro-benchmark: issue cost of the instructions the count kernel (K1) is made of, on gfx950, as a
// function of waves per SIMD.  Every kernel is a loop of 32 instructions of ONE kind
// (eight independent accumulators x four rounds) written in inline asm so hipcc cannot fold or re-select them;
// cycles are read inside the kernel with s_memtime (shader clock), so no clock is assumed.
// (Rows named a_then_b hold two or four instructions per slot: divide by that.)
// Reported: s_memtime ticks per instruction per SIMD = wave cycles / (instructions per wave x waves per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o tools/issue_rate.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

// One round = the instruction applied to each of the eight accumulators, as ONE asm statement (the
// compiler pads separate asm statements with s_nop, which would be measured too).
struct v_and_b32 {
    static constexpr const char* name = "v_and_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", " "%0" ", %8\n\t" "v_and_b32 " "%1" ", " "%1" ", %8\n\t" "v_and_b32 " "%2" ", " "%2" ", %8\n\t" "v_and_b32 " "%3" ", " "%3" ", %8\n\t" "v_and_b32 " "%4" ", " "%4" ", %8\n\t" "v_and_b32 " "%5" ", " "%5" ", %8\n\t" "v_and_b32 " "%6" ", " "%6" ", %8\n\t" "v_and_b32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add_u32 {
    static constexpr const char* name = "v_add_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add_u32 " "%0" ", " "%0" ", %8\n\t" "v_add_u32 " "%1" ", " "%1" ", %8\n\t" "v_add_u32 " "%2" ", " "%2" ", %8\n\t" "v_add_u32 " "%3" ", " "%3" ", %8\n\t" "v_add_u32 " "%4" ", " "%4" ", %8\n\t" "v_add_u32 " "%5" ", " "%5" ", %8\n\t" "v_add_u32 " "%6" ", " "%6" ", %8\n\t" "v_add_u32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_lshl_add_u32 {
    static constexpr const char* name = "v_lshl_add_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_lshl_add_u32 " "%0" ", " "%0" ", 2, %8\n\t" "v_lshl_add_u32 " "%1" ", " "%1" ", 2, %8\n\t" "v_lshl_add_u32 " "%2" ", " "%2" ", 2, %8\n\t" "v_lshl_add_u32 " "%3" ", " "%3" ", 2, %8\n\t" "v_lshl_add_u32 " "%4" ", " "%4" ", 2, %8\n\t" "v_lshl_add_u32 " "%5" ", " "%5" ", 2, %8\n\t" "v_lshl_add_u32 " "%6" ", " "%6" ", 2, %8\n\t" "v_lshl_add_u32 " "%7" ", " "%7" ", 2, %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_lshl_or_b32 {
    static constexpr const char* name = "v_lshl_or_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_lshl_or_b32 " "%0" ", " "%0" ", 2, %8\n\t" "v_lshl_or_b32 " "%1" ", " "%1" ", 2, %8\n\t" "v_lshl_or_b32 " "%2" ", " "%2" ", 2, %8\n\t" "v_lshl_or_b32 " "%3" ", " "%3" ", 2, %8\n\t" "v_lshl_or_b32 " "%4" ", " "%4" ", 2, %8\n\t" "v_lshl_or_b32 " "%5" ", " "%5" ", 2, %8\n\t" "v_lshl_or_b32 " "%6" ", " "%6" ", 2, %8\n\t" "v_lshl_or_b32 " "%7" ", " "%7" ", 2, %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_perm_b32 {
    static constexpr const char* name = "v_perm_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_perm_b32 " "%0" ", " "%0" ", %8, %9\n\t" "v_perm_b32 " "%1" ", " "%1" ", %8, %9\n\t" "v_perm_b32 " "%2" ", " "%2" ", %8, %9\n\t" "v_perm_b32 " "%3" ", " "%3" ", %8, %9\n\t" "v_perm_b32 " "%4" ", " "%4" ", %8, %9\n\t" "v_perm_b32 " "%5" ", " "%5" ", %8, %9\n\t" "v_perm_b32 " "%6" ", " "%6" ", %8, %9\n\t" "v_perm_b32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_alignbit_b32 {
    static constexpr const char* name = "v_alignbit_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_alignbit_b32 " "%0" ", " "%0" ", %8, 7\n\t" "v_alignbit_b32 " "%1" ", " "%1" ", %8, 7\n\t" "v_alignbit_b32 " "%2" ", " "%2" ", %8, 7\n\t" "v_alignbit_b32 " "%3" ", " "%3" ", %8, 7\n\t" "v_alignbit_b32 " "%4" ", " "%4" ", %8, 7\n\t" "v_alignbit_b32 " "%5" ", " "%5" ", %8, 7\n\t" "v_alignbit_b32 " "%6" ", " "%6" ", %8, 7\n\t" "v_alignbit_b32 " "%7" ", " "%7" ", %8, 7\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_or_b32 {
    static constexpr const char* name = "v_and_or_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_or_b32 " "%0" ", " "%0" ", %9, %8\n\t" "v_and_or_b32 " "%1" ", " "%1" ", %9, %8\n\t" "v_and_or_b32 " "%2" ", " "%2" ", %9, %8\n\t" "v_and_or_b32 " "%3" ", " "%3" ", %9, %8\n\t" "v_and_or_b32 " "%4" ", " "%4" ", %9, %8\n\t" "v_and_or_b32 " "%5" ", " "%5" ", %9, %8\n\t" "v_and_or_b32 " "%6" ", " "%6" ", %9, %8\n\t" "v_and_or_b32 " "%7" ", " "%7" ", %9, %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_bitop3_b32 {
    static constexpr const char* name = "v_bitop3_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_bitop3_b32 " "%0" ", " "%0" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%1" ", " "%1" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%2" ", " "%2" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%3" ", " "%3" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%4" ", " "%4" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%5" ", " "%5" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%6" ", " "%6" ", %8, %9 bitop3:0x78\n\t" "v_bitop3_b32 " "%7" ", " "%7" ", %8, %9 bitop3:0x78\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_xad_u32 {
    static constexpr const char* name = "v_xad_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_xad_u32 " "%0" ", " "%0" ", %9, %8\n\t" "v_xad_u32 " "%1" ", " "%1" ", %9, %8\n\t" "v_xad_u32 " "%2" ", " "%2" ", %9, %8\n\t" "v_xad_u32 " "%3" ", " "%3" ", %9, %8\n\t" "v_xad_u32 " "%4" ", " "%4" ", %9, %8\n\t" "v_xad_u32 " "%5" ", " "%5" ", %9, %8\n\t" "v_xad_u32 " "%6" ", " "%6" ", %9, %8\n\t" "v_xad_u32 " "%7" ", " "%7" ", %9, %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_or3_b32 {
    static constexpr const char* name = "v_or3_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_or3_b32 " "%0" ", " "%0" ", %8, %9\n\t" "v_or3_b32 " "%1" ", " "%1" ", %8, %9\n\t" "v_or3_b32 " "%2" ", " "%2" ", %8, %9\n\t" "v_or3_b32 " "%3" ", " "%3" ", %8, %9\n\t" "v_or3_b32 " "%4" ", " "%4" ", %8, %9\n\t" "v_or3_b32 " "%5" ", " "%5" ", %8, %9\n\t" "v_or3_b32 " "%6" ", " "%6" ", %8, %9\n\t" "v_or3_b32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_bfe_u32 {
    static constexpr const char* name = "v_bfe_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_bfe_u32 " "%0" ", " "%0" ", 3, 9\n\t" "v_bfe_u32 " "%1" ", " "%1" ", 3, 9\n\t" "v_bfe_u32 " "%2" ", " "%2" ", 3, 9\n\t" "v_bfe_u32 " "%3" ", " "%3" ", 3, 9\n\t" "v_bfe_u32 " "%4" ", " "%4" ", 3, 9\n\t" "v_bfe_u32 " "%5" ", " "%5" ", 3, 9\n\t" "v_bfe_u32 " "%6" ", " "%6" ", 3, 9\n\t" "v_bfe_u32 " "%7" ", " "%7" ", 3, 9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_bcnt_u32_b32 {
    static constexpr const char* name = "v_bcnt_u32_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_bcnt_u32_b32 " "%0" ", " "%0" ", %8\n\t" "v_bcnt_u32_b32 " "%1" ", " "%1" ", %8\n\t" "v_bcnt_u32_b32 " "%2" ", " "%2" ", %8\n\t" "v_bcnt_u32_b32 " "%3" ", " "%3" ", %8\n\t" "v_bcnt_u32_b32 " "%4" ", " "%4" ", %8\n\t" "v_bcnt_u32_b32 " "%5" ", " "%5" ", %8\n\t" "v_bcnt_u32_b32 " "%6" ", " "%6" ", %8\n\t" "v_bcnt_u32_b32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_ffbl_b32 {
    static constexpr const char* name = "v_ffbl_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_ffbl_b32 " "%0" ", " "%0" "\n\t" "v_ffbl_b32 " "%1" ", " "%1" "\n\t" "v_ffbl_b32 " "%2" ", " "%2" "\n\t" "v_ffbl_b32 " "%3" ", " "%3" "\n\t" "v_ffbl_b32 " "%4" ", " "%4" "\n\t" "v_ffbl_b32 " "%5" ", " "%5" "\n\t" "v_ffbl_b32 " "%6" ", " "%6" "\n\t" "v_ffbl_b32 " "%7" ", " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_min3_u32 {
    static constexpr const char* name = "v_min3_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_min3_u32 " "%0" ", " "%0" ", %8, %9\n\t" "v_min3_u32 " "%1" ", " "%1" ", %8, %9\n\t" "v_min3_u32 " "%2" ", " "%2" ", %8, %9\n\t" "v_min3_u32 " "%3" ", " "%3" ", %8, %9\n\t" "v_min3_u32 " "%4" ", " "%4" ", %8, %9\n\t" "v_min3_u32 " "%5" ", " "%5" ", %8, %9\n\t" "v_min3_u32 " "%6" ", " "%6" ", %8, %9\n\t" "v_min3_u32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_mul_u32_u24 {
    static constexpr const char* name = "v_mul_u32_u24";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_mul_u32_u24 " "%0" ", " "%0" ", %8\n\t" "v_mul_u32_u24 " "%1" ", " "%1" ", %8\n\t" "v_mul_u32_u24 " "%2" ", " "%2" ", %8\n\t" "v_mul_u32_u24 " "%3" ", " "%3" ", %8\n\t" "v_mul_u32_u24 " "%4" ", " "%4" ", %8\n\t" "v_mul_u32_u24 " "%5" ", " "%5" ", %8\n\t" "v_mul_u32_u24 " "%6" ", " "%6" ", %8\n\t" "v_mul_u32_u24 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_mul_lo_u32 {
    static constexpr const char* name = "v_mul_lo_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_mul_lo_u32 " "%0" ", " "%0" ", %8\n\t" "v_mul_lo_u32 " "%1" ", " "%1" ", %8\n\t" "v_mul_lo_u32 " "%2" ", " "%2" ", %8\n\t" "v_mul_lo_u32 " "%3" ", " "%3" ", %8\n\t" "v_mul_lo_u32 " "%4" ", " "%4" ", %8\n\t" "v_mul_lo_u32 " "%5" ", " "%5" ", %8\n\t" "v_mul_lo_u32 " "%6" ", " "%6" ", %8\n\t" "v_mul_lo_u32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add_u32_dpp_row_shr1 {
    static constexpr const char* name = "v_add_u32_dpp_row_shr1";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add_u32_dpp " "%0" ", " "%0" ", " "%0" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%1" ", " "%1" ", " "%1" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%2" ", " "%2" ", " "%2" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%3" ", " "%3" ", " "%3" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%4" ", " "%4" ", " "%4" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%5" ", " "%5" ", " "%5" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%6" ", " "%6" ", " "%6" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" "v_add_u32_dpp " "%7" ", " "%7" ", " "%7" " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_mov_b32_dpp_wave_shr1 {
    static constexpr const char* name = "v_mov_b32_dpp_wave_shr1";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_mov_b32_dpp " "%0" ", " "%0" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%1" ", " "%1" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%2" ", " "%2" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%3" ", " "%3" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%4" ", " "%4" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%5" ", " "%5" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%6" ", " "%6" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" "v_mov_b32_dpp " "%7" ", " "%7" " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_b32_sdwa_word1 {
    static constexpr const char* name = "v_and_b32_sdwa_word1";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32_sdwa " "%0" ", " "%0" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%1" ", " "%1" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%2" ", " "%2" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%3" ", " "%3" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%4" ", " "%4" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%5" ", " "%5" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%6" ", " "%6" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" "v_and_b32_sdwa " "%7" ", " "%7" ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add_co_u32_sgpr_carry {
    static constexpr const char* name = "v_add_co_u32_sgpr_carry";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add_co_u32_e64 " "%0" ", s[20:21], " "%0" ", " "%0" "\n\t" "v_add_co_u32_e64 " "%1" ", s[20:21], " "%1" ", " "%1" "\n\t" "v_add_co_u32_e64 " "%2" ", s[20:21], " "%2" ", " "%2" "\n\t" "v_add_co_u32_e64 " "%3" ", s[20:21], " "%3" ", " "%3" "\n\t" "v_add_co_u32_e64 " "%4" ", s[20:21], " "%4" ", " "%4" "\n\t" "v_add_co_u32_e64 " "%5" ", s[20:21], " "%5" ", " "%5" "\n\t" "v_add_co_u32_e64 " "%6" ", s[20:21], " "%6" ", " "%6" "\n\t" "v_add_co_u32_e64 " "%7" ", s[20:21], " "%7" ", " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add_co_u32_vcc {
    static constexpr const char* name = "v_add_co_u32_vcc";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add_co_u32_e32 " "%0" ", vcc, " "%0" ", " "%0" "\n\t" "v_add_co_u32_e32 " "%1" ", vcc, " "%1" ", " "%1" "\n\t" "v_add_co_u32_e32 " "%2" ", vcc, " "%2" ", " "%2" "\n\t" "v_add_co_u32_e32 " "%3" ", vcc, " "%3" ", " "%3" "\n\t" "v_add_co_u32_e32 " "%4" ", vcc, " "%4" ", " "%4" "\n\t" "v_add_co_u32_e32 " "%5" ", vcc, " "%5" ", " "%5" "\n\t" "v_add_co_u32_e32 " "%6" ", vcc, " "%6" ", " "%6" "\n\t" "v_add_co_u32_e32 " "%7" ", vcc, " "%7" ", " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_readlane_b32 {
    static constexpr const char* name = "v_readlane_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_readlane_b32 s22, " "%0" ", 63\n\t" "v_readlane_b32 s22, " "%1" ", 63\n\t" "v_readlane_b32 s22, " "%2" ", 63\n\t" "v_readlane_b32 s22, " "%3" ", 63\n\t" "v_readlane_b32 s22, " "%4" ", 63\n\t" "v_readlane_b32 s22, " "%5" ", 63\n\t" "v_readlane_b32 s22, " "%6" ", 63\n\t" "v_readlane_b32 s22, " "%7" ", 63\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct s_mov_then_v_and {
    static constexpr const char* name = "s_mov_then_v_and";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("s_mov_b32 s22, %9\n\tv_and_b32 " "%0" ", " "%0" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%1" ", " "%1" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%2" ", " "%2" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%3" ", " "%3" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%4" ", " "%4" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%5" ", " "%5" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%6" ", " "%6" ", %8\n\t" "s_mov_b32 s22, %9\n\tv_and_b32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_then_s_nop {
    static constexpr const char* name = "v_and_then_s_nop";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", " "%0" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%1" ", " "%1" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%2" ", " "%2" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%3" ", " "%3" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%4" ", " "%4" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%5" ", " "%5" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%6" ", " "%6" ", %8\n\ts_nop 0\n\t" "v_and_b32 " "%7" ", " "%7" ", %8\n\ts_nop 0\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};

struct v_and_b32_sgpr {
    static constexpr const char* name = "v_and_b32_sgpr";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", %9, " "%0" "\n\t" "v_and_b32 " "%1" ", %9, " "%1" "\n\t" "v_and_b32 " "%2" ", %9, " "%2" "\n\t" "v_and_b32 " "%3" ", %9, " "%3" "\n\t" "v_and_b32 " "%4" ", %9, " "%4" "\n\t" "v_and_b32 " "%5" ", %9, " "%5" "\n\t" "v_and_b32 " "%6" ", %9, " "%6" "\n\t" "v_and_b32 " "%7" ", %9, " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_b32_literal {
    static constexpr const char* name = "v_and_b32_literal";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", 0x7070707, " "%0" "\n\t" "v_and_b32 " "%1" ", 0x7070707, " "%1" "\n\t" "v_and_b32 " "%2" ", 0x7070707, " "%2" "\n\t" "v_and_b32 " "%3" ", 0x7070707, " "%3" "\n\t" "v_and_b32 " "%4" ", 0x7070707, " "%4" "\n\t" "v_and_b32 " "%5" ", 0x7070707, " "%5" "\n\t" "v_and_b32 " "%6" ", 0x7070707, " "%6" "\n\t" "v_and_b32 " "%7" ", 0x7070707, " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_b32_e64 {
    static constexpr const char* name = "v_and_b32_e64";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32_e64 " "%0" ", " "%0" ", %8\n\t" "v_and_b32_e64 " "%1" ", " "%1" ", %8\n\t" "v_and_b32_e64 " "%2" ", " "%2" ", %8\n\t" "v_and_b32_e64 " "%3" ", " "%3" ", %8\n\t" "v_and_b32_e64 " "%4" ", " "%4" ", %8\n\t" "v_and_b32_e64 " "%5" ", " "%5" ", %8\n\t" "v_and_b32_e64 " "%6" ", " "%6" ", %8\n\t" "v_and_b32_e64 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add_u32_literal {
    static constexpr const char* name = "v_add_u32_literal";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add_u32 " "%0" ", 0x7c7c7c7c, " "%0" "\n\t" "v_add_u32 " "%1" ", 0x7c7c7c7c, " "%1" "\n\t" "v_add_u32 " "%2" ", 0x7c7c7c7c, " "%2" "\n\t" "v_add_u32 " "%3" ", 0x7c7c7c7c, " "%3" "\n\t" "v_add_u32 " "%4" ", 0x7c7c7c7c, " "%4" "\n\t" "v_add_u32 " "%5" ", 0x7c7c7c7c, " "%5" "\n\t" "v_add_u32 " "%6" ", 0x7c7c7c7c, " "%6" "\n\t" "v_add_u32 " "%7" ", 0x7c7c7c7c, " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_lshlrev_b32_imm {
    static constexpr const char* name = "v_lshlrev_b32_imm";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_lshlrev_b32 " "%0" ", 2, " "%0" "\n\t" "v_lshlrev_b32 " "%1" ", 2, " "%1" "\n\t" "v_lshlrev_b32 " "%2" ", 2, " "%2" "\n\t" "v_lshlrev_b32 " "%3" ", 2, " "%3" "\n\t" "v_lshlrev_b32 " "%4" ", 2, " "%4" "\n\t" "v_lshlrev_b32 " "%5" ", 2, " "%5" "\n\t" "v_lshlrev_b32 " "%6" ", 2, " "%6" "\n\t" "v_lshlrev_b32 " "%7" ", 2, " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_lshrrev_b32_imm {
    static constexpr const char* name = "v_lshrrev_b32_imm";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_lshrrev_b32 " "%0" ", 7, " "%0" "\n\t" "v_lshrrev_b32 " "%1" ", 7, " "%1" "\n\t" "v_lshrrev_b32 " "%2" ", 7, " "%2" "\n\t" "v_lshrrev_b32 " "%3" ", 7, " "%3" "\n\t" "v_lshrrev_b32 " "%4" ", 7, " "%4" "\n\t" "v_lshrrev_b32 " "%5" ", 7, " "%5" "\n\t" "v_lshrrev_b32 " "%6" ", 7, " "%6" "\n\t" "v_lshrrev_b32 " "%7" ", 7, " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_or_b32 {
    static constexpr const char* name = "v_or_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_or_b32 " "%0" ", " "%0" ", %8\n\t" "v_or_b32 " "%1" ", " "%1" ", %8\n\t" "v_or_b32 " "%2" ", " "%2" ", %8\n\t" "v_or_b32 " "%3" ", " "%3" ", %8\n\t" "v_or_b32 " "%4" ", " "%4" ", %8\n\t" "v_or_b32 " "%5" ", " "%5" ", %8\n\t" "v_or_b32 " "%6" ", " "%6" ", %8\n\t" "v_or_b32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_xor_b32 {
    static constexpr const char* name = "v_xor_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_xor_b32 " "%0" ", " "%0" ", %8\n\t" "v_xor_b32 " "%1" ", " "%1" ", %8\n\t" "v_xor_b32 " "%2" ", " "%2" ", %8\n\t" "v_xor_b32 " "%3" ", " "%3" ", %8\n\t" "v_xor_b32 " "%4" ", " "%4" ", %8\n\t" "v_xor_b32 " "%5" ", " "%5" ", %8\n\t" "v_xor_b32 " "%6" ", " "%6" ", %8\n\t" "v_xor_b32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_mov_b32 {
    static constexpr const char* name = "v_mov_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_mov_b32 " "%0" ", %8\n\t" "v_mov_b32 " "%1" ", %8\n\t" "v_mov_b32 " "%2" ", %8\n\t" "v_mov_b32 " "%3" ", %8\n\t" "v_mov_b32 " "%4" ", %8\n\t" "v_mov_b32 " "%5" ", %8\n\t" "v_mov_b32 " "%6" ", %8\n\t" "v_mov_b32 " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_not_b32 {
    static constexpr const char* name = "v_not_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_not_b32 " "%0" ", " "%0" "\n\t" "v_not_b32 " "%1" ", " "%1" "\n\t" "v_not_b32 " "%2" ", " "%2" "\n\t" "v_not_b32 " "%3" ", " "%3" "\n\t" "v_not_b32 " "%4" ", " "%4" "\n\t" "v_not_b32 " "%5" ", " "%5" "\n\t" "v_not_b32 " "%6" ", " "%6" "\n\t" "v_not_b32 " "%7" ", " "%7" "\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_sub_u32 {
    static constexpr const char* name = "v_sub_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_sub_u32 " "%0" ", " "%0" ", %8\n\t" "v_sub_u32 " "%1" ", " "%1" ", %8\n\t" "v_sub_u32 " "%2" ", " "%2" ", %8\n\t" "v_sub_u32 " "%3" ", " "%3" ", %8\n\t" "v_sub_u32 " "%4" ", " "%4" ", %8\n\t" "v_sub_u32 " "%5" ", " "%5" ", %8\n\t" "v_sub_u32 " "%6" ", " "%6" ", %8\n\t" "v_sub_u32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_max_u32 {
    static constexpr const char* name = "v_max_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_max_u32 " "%0" ", " "%0" ", %8\n\t" "v_max_u32 " "%1" ", " "%1" ", %8\n\t" "v_max_u32 " "%2" ", " "%2" ", %8\n\t" "v_max_u32 " "%3" ", " "%3" ", %8\n\t" "v_max_u32 " "%4" ", " "%4" ", %8\n\t" "v_max_u32 " "%5" ", " "%5" ", %8\n\t" "v_max_u32 " "%6" ", " "%6" ", %8\n\t" "v_max_u32 " "%7" ", " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_bfi_b32 {
    static constexpr const char* name = "v_bfi_b32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_bfi_b32 " "%0" ", %9, " "%0" ", %8\n\t" "v_bfi_b32 " "%1" ", %9, " "%1" ", %8\n\t" "v_bfi_b32 " "%2" ", %9, " "%2" ", %8\n\t" "v_bfi_b32 " "%3" ", %9, " "%3" ", %8\n\t" "v_bfi_b32 " "%4" ", %9, " "%4" ", %8\n\t" "v_bfi_b32 " "%5" ", %9, " "%5" ", %8\n\t" "v_bfi_b32 " "%6" ", %9, " "%6" ", %8\n\t" "v_bfi_b32 " "%7" ", %9, " "%7" ", %8\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_dot4_u32_u8 {
    static constexpr const char* name = "v_dot4_u32_u8";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_dot4_u32_u8 %0, %0, %9, %8\n\t" "v_dot4_u32_u8 %1, %1, %9, %8\n\t" "v_dot4_u32_u8 %2, %2, %9, %8\n\t" "v_dot4_u32_u8 %3, %3, %9, %8\n\t" "v_dot4_u32_u8 %4, %4, %9, %8\n\t" "v_dot4_u32_u8 %5, %5, %9, %8\n\t" "v_dot4_u32_u8 %6, %6, %9, %8\n\t" "v_dot4_u32_u8 %7, %7, %9, %8\n\t" 
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_dot4_u32_u8_acc0 {
    static constexpr const char* name = "v_dot4_u32_u8_acc0";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_dot4_u32_u8 %0, %0, %9, 0\n\t" "v_dot4_u32_u8 %1, %1, %9, 0\n\t" "v_dot4_u32_u8 %2, %2, %9, 0\n\t" "v_dot4_u32_u8 %3, %3, %9, 0\n\t" "v_dot4_u32_u8 %4, %4, %9, 0\n\t" "v_dot4_u32_u8 %5, %5, %9, 0\n\t" "v_dot4_u32_u8 %6, %6, %9, 0\n\t" "v_dot4_u32_u8 %7, %7, %9, 0\n\t" 
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_mbcnt_lo {
    static constexpr const char* name = "v_mbcnt_lo";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0\n\t" "v_mbcnt_lo_u32_b32 %1, -1, %1\n\t" "v_mbcnt_lo_u32_b32 %2, -1, %2\n\t" "v_mbcnt_lo_u32_b32 %3, -1, %3\n\t" "v_mbcnt_lo_u32_b32 %4, -1, %4\n\t" "v_mbcnt_lo_u32_b32 %5, -1, %5\n\t" "v_mbcnt_lo_u32_b32 %6, -1, %6\n\t" "v_mbcnt_lo_u32_b32 %7, -1, %7\n\t" 
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_cmp_e64_sgpr {
    static constexpr const char* name = "v_cmp_e64_sgpr";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %1, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %2, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %3, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %4, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %5, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %6, %8\n\t" "v_cmp_lt_u32_e64 s[20:21], %7, %8\n\t" 
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_cndmask_e64 {
    static constexpr const char* name = "v_cndmask_e64";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n\t" "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n\t" 
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_add3_u32 {
    static constexpr const char* name = "v_add3_u32";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_add3_u32 " "%0" ", " "%0" ", %8, %9\n\t" "v_add3_u32 " "%1" ", " "%1" ", %8, %9\n\t" "v_add3_u32 " "%2" ", " "%2" ", %8, %9\n\t" "v_add3_u32 " "%3" ", " "%3" ", %8, %9\n\t" "v_add3_u32 " "%4" ", " "%4" ", %8, %9\n\t" "v_add3_u32 " "%5" ", " "%5" ", %8, %9\n\t" "v_add3_u32 " "%6" ", " "%6" ", %8, %9\n\t" "v_add3_u32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_then_v_perm {
    static constexpr const char* name = "v_and_then_v_perm";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", " "%0" ", %8\n\tv_perm_b32 " "%0" ", " "%0" ", %8, %9\n\t" "v_and_b32 " "%1" ", " "%1" ", %8\n\tv_perm_b32 " "%1" ", " "%1" ", %8, %9\n\t" "v_and_b32 " "%2" ", " "%2" ", %8\n\tv_perm_b32 " "%2" ", " "%2" ", %8, %9\n\t" "v_and_b32 " "%3" ", " "%3" ", %8\n\tv_perm_b32 " "%3" ", " "%3" ", %8, %9\n\t" "v_and_b32 " "%4" ", " "%4" ", %8\n\tv_perm_b32 " "%4" ", " "%4" ", %8, %9\n\t" "v_and_b32 " "%5" ", " "%5" ", %8\n\tv_perm_b32 " "%5" ", " "%5" ", %8, %9\n\t" "v_and_b32 " "%6" ", " "%6" ", %8\n\tv_perm_b32 " "%6" ", " "%6" ", %8, %9\n\t" "v_and_b32 " "%7" ", " "%7" ", %8\n\tv_perm_b32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};
struct v_and_x3_then_v_perm {
    static constexpr const char* name = "v_and_x3_then_v_perm";
    __device__ __forceinline__ static void round(uint32_t (&a)[8], uint32_t b, uint32_t ks) {
        asm volatile("v_and_b32 " "%0" ", " "%0" ", %8\n\tv_or_b32 " "%0" ", " "%0" ", %8\n\tv_xor_b32 " "%0" ", " "%0" ", %8\n\tv_perm_b32 " "%0" ", " "%0" ", %8, %9\n\t" "v_and_b32 " "%1" ", " "%1" ", %8\n\tv_or_b32 " "%1" ", " "%1" ", %8\n\tv_xor_b32 " "%1" ", " "%1" ", %8\n\tv_perm_b32 " "%1" ", " "%1" ", %8, %9\n\t" "v_and_b32 " "%2" ", " "%2" ", %8\n\tv_or_b32 " "%2" ", " "%2" ", %8\n\tv_xor_b32 " "%2" ", " "%2" ", %8\n\tv_perm_b32 " "%2" ", " "%2" ", %8, %9\n\t" "v_and_b32 " "%3" ", " "%3" ", %8\n\tv_or_b32 " "%3" ", " "%3" ", %8\n\tv_xor_b32 " "%3" ", " "%3" ", %8\n\tv_perm_b32 " "%3" ", " "%3" ", %8, %9\n\t" "v_and_b32 " "%4" ", " "%4" ", %8\n\tv_or_b32 " "%4" ", " "%4" ", %8\n\tv_xor_b32 " "%4" ", " "%4" ", %8\n\tv_perm_b32 " "%4" ", " "%4" ", %8, %9\n\t" "v_and_b32 " "%5" ", " "%5" ", %8\n\tv_or_b32 " "%5" ", " "%5" ", %8\n\tv_xor_b32 " "%5" ", " "%5" ", %8\n\tv_perm_b32 " "%5" ", " "%5" ", %8, %9\n\t" "v_and_b32 " "%6" ", " "%6" ", %8\n\tv_or_b32 " "%6" ", " "%6" ", %8\n\tv_xor_b32 " "%6" ", " "%6" ", %8\n\tv_perm_b32 " "%6" ", " "%6" ", %8, %9\n\t" "v_and_b32 " "%7" ", " "%7" ", %8\n\tv_or_b32 " "%7" ", " "%7" ", %8\n\tv_xor_b32 " "%7" ", " "%7" ", %8\n\tv_perm_b32 " "%7" ", " "%7" ", %8, %9\n\t"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                     : "v"(b), "s"(ks)
                     : "s20", "s21", "s22", "vcc");
    }
};

template <typename OP>
__global__ __launch_bounds__(1024) void k_op(uint64_t* cyc, uint32_t* sink, int iters) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * (2 * i + 3) + blockIdx.x;
    const uint32_t b = threadIdx.x ^ 0x5a5a5a5au, ks = 0x07070707u;
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory", "s20", "s21", "s22", "vcc");
    for (int it = 0; it < iters; ++it) {
        OP::round(a, b, ks);
        OP::round(a, b, ks);
        OP::round(a, b, ks);
        OP::round(a, b, ks);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= a[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// The window block of K1 as it is written in vk_count.h: eight carry-outs parked in SGPR pairs, then
// eight {s_mov_b64 exec; ds_add_u32}.  MODE 0: the block alone; 1: plus the address VALU (alignbit,
// and, sdwa-and, add) that feeds it; 2: ds_add with exec untouched (all lanes), no s_mov.
template <int MODE>
__global__ __launch_bounds__(1024) void k_window(uint64_t* cyc, uint32_t* sink, int iters) {
    __shared__ uint32_t h[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) h[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t w = s * 747796405u + 2891336453u;
    uint32_t a[8];
    const uint32_t one = 1u;
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        w = (w << 16) ^ (s >> 3);  // ~half of the carries set
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t x = __builtin_amdgcn_alignbit(s, w, 3 + 2 * j);
                a[2 * j] = (x & 0xFFFCu) + 0u;
                a[2 * j + 1] = ((x >> 16) & 0xFFFCu) + 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = ((s >> j) ^ (s << (j + 3))) & 0xFFFCu;
        }
        if (MODE == 2) {
            asm volatile(
                "ds_add_u32 %0, %8\n\tds_add_u32 %1, %8\n\tds_add_u32 %2, %8\n\tds_add_u32 %3, %8\n\t"
                "ds_add_u32 %4, %8\n\tds_add_u32 %5, %8\n\tds_add_u32 %6, %8\n\tds_add_u32 %7, %8"
                :
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(one)
                : "memory");
        } else {
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    __syncthreads();
    uint32_t t = w;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) t += h[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static uint64_t* d_cyc;
static uint32_t* d_sink;
static float g_last_ms;  // wall time of the last timed launch (calibrates the s_memtime tick)

template <typename F>
double median_cycles(F launch, int nwaves) {
    launch(4);
    hipDeviceSynchronize();
    const int iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&g_last_ms, e0, e1);
    std::vector<uint64_t> c(nwaves);
    hipMemcpy(c.data(), d_cyc, nwaves * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return static_cast<double>(c[nwaves / 2]) / iters;
}

template <typename OP>
void run_op() {
    printf("%-26s", OP::name);
    for (int threads : {256, 512, 1024, 2048}) {  // 1, 2, 4, 8 waves per SIMD (2048 = two 1024-thread WGs per CU)
        const int wg = threads > 1024 ? 1024 : threads, grid = 256 * (threads / wg);
        const double cyc = median_cycles(
            [&](int it) { hipLaunchKernelGGL(k_op<OP>, dim3(grid), dim3(wg), 0, 0, d_cyc, d_sink, it); }, grid * wg / 64);
        const double wps = threads / 256.0;
        printf("  %dw/SIMD %5.2f", static_cast<int>(wps), cyc / (32.0 * wps));
        if (threads == 2048) printf("  [tick = %.3f ns]", g_last_ms * 1e6 / (cyc * 2048));
    }
    printf("   cycles per instruction per SIMD\n");
}

template <int MODE>
void run_window(const char* what) {
    printf("%-26s", what);
    for (int threads : {256, 512, 1024}) {
        const double cyc = median_cycles(
            [&](int it) { hipLaunchKernelGGL(k_window<MODE>, dim3(256), dim3(threads), 0, 0, d_cyc, d_sink, it); },
            256 * threads / 64);
        printf("  %2d waves/CU %6.1f cyc/block/wave = %5.2f cyc/ds_add/CU", threads / 64, cyc, cyc / 8.0 / (threads / 64));
    }
    printf("\n");
}

int main() {
    hipMalloc(&d_cyc, 8192 * sizeof(uint64_t));
    hipMalloc(&d_sink, 2048 * 256 * 4);
    run_op<v_and_b32>();
    run_op<v_add_u32>();
    run_op<v_lshl_add_u32>();
    run_op<v_lshl_or_b32>();
    run_op<v_perm_b32>();
    run_op<v_alignbit_b32>();
    run_op<v_and_or_b32>();
    run_op<v_bitop3_b32>();
    run_op<v_xad_u32>();
    run_op<v_or3_b32>();
    run_op<v_bfe_u32>();
    run_op<v_bcnt_u32_b32>();
    run_op<v_ffbl_b32>();
    run_op<v_min3_u32>();
    run_op<v_mul_u32_u24>();
    run_op<v_mul_lo_u32>();
    run_op<v_add_u32_dpp_row_shr1>();
    run_op<v_mov_b32_dpp_wave_shr1>();
    run_op<v_and_b32_sdwa_word1>();
    run_op<v_add_co_u32_sgpr_carry>();
    run_op<v_add_co_u32_vcc>();
    run_op<v_readlane_b32>();
    run_op<s_mov_then_v_and>();
    run_op<v_and_then_s_nop>();
    run_op<v_and_b32_sgpr>();
    run_op<v_and_b32_literal>();
    run_op<v_and_b32_e64>();
    run_op<v_add_u32_literal>();
    run_op<v_lshlrev_b32_imm>();
    run_op<v_lshrrev_b32_imm>();
    run_op<v_or_b32>();
    run_op<v_xor_b32>();
    run_op<v_mov_b32>();
    run_op<v_not_b32>();
    run_op<v_sub_u32>();
    run_op<v_max_u32>();
    run_op<v_bfi_b32>();
    run_op<v_add3_u32>();
    run_op<v_dot4_u32_u8>();
    run_op<v_dot4_u32_u8_acc0>();
    run_op<v_mbcnt_lo>();
    run_op<v_cmp_e64_sgpr>();
    run_op<v_cndmask_e64>();
    run_op<v_and_then_v_perm>();
    run_op<v_and_x3_then_v_perm>();
    run_window<0>("window block (8 pos)");
    run_window<1>("window block + addr VALU");
    run_window<2>("8 ds_add, all lanes");
    return 0;
}
