// Microbenchmark (design input for k_probe_mz, round 2): issue rate of the VALU / LDS-crossbar instructions the kernel is made
// of, in wave-instructions per cycle per SIMD.  8 independent chains per lane, 4 waves per SIMD, inline asm so that the
// compiler neither folds nor reorders them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define CHAIN8(INS)                                                                                                  \
    asm volatile(INS(%0) "\n\t" INS(%1) "\n\t" INS(%2) "\n\t" INS(%3) "\n\t" INS(%4) "\n\t" INS(%5) "\n\t" INS(%6) "\n\t" INS(%7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc")

#define I_ADD(r) "v_add_u32 " #r ", " #r ", %8"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, " #r
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %8"
#define I_PKMIN(r) "v_pk_min_u16 " #r ", " #r ", %8"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8"
#define I_BFREV(r) "v_bfrev_b32 " #r ", " #r
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, " #r
#define I_BCNT(r) "v_bcnt_u32_b32 " #r ", " #r ", %8"
#define I_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define I_DPP(r) "v_mov_b32_dpp " #r ", " #r " wave_shr:1 row_mask:0xf bank_mask:0xf"
#define I_BPERM(r) "ds_bpermute_b32 " #r ", %8, " #r "\n\ts_waitcnt lgkmcnt(0)"
#define I_MAD_U64(r) "v_mad_u64_u32 %9, vcc, " #r ", %8, %9"
#define I_MULLO_I(r) "v_mul_lo_u32 " #r ", " #r ", 0x9e3779b1"

template <int WHICH>
__global__ __launch_bounds__(256) void k(uint32_t *out, int rounds) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t c = 0x9E3779B1u + blockIdx.x;
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (WHICH == 0) CHAIN8(I_ADD);
            if (WHICH == 1) CHAIN8(I_MULLO);
            if (WHICH == 2) CHAIN8(I_MULHI);
            if (WHICH == 3) CHAIN8(I_MUL24);
            if (WHICH == 4) CHAIN8(I_MAD24);
            if (WHICH == 5) CHAIN8(I_LSHLADD);
            if (WHICH == 6) CHAIN8(I_PKMIN);
            if (WHICH == 7) CHAIN8(I_XOR);
            if (WHICH == 8) CHAIN8(I_BFREV);
            if (WHICH == 9) CHAIN8(I_PERM);
            if (WHICH == 10) CHAIN8(I_BCNT);
            if (WHICH == 11) CHAIN8(I_CNDMASK);
            if (WHICH == 12) CHAIN8(I_DPP);
            if (WHICH == 13) CHAIN8(I_BPERM);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// 64-bit shifts / compares as the compiler emits them
template <int WHICH>
__global__ __launch_bounds__(256) void k64(uint64_t *out, int rounds) {
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t sh = (threadIdx.x & 31) + 1;
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (WHICH == 0) {
                asm volatile("v_lshlrev_b64 %0, %4, %0\n\tv_lshlrev_b64 %1, %4, %1\n\tv_lshlrev_b64 %2, %4, %2\n\tv_lshlrev_b64 %3, %4, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));
            } else {
                asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cmp_lt_u64 vcc, %1, %2\n\tv_cmp_lt_u64 vcc, %2, %3\n\tv_cmp_lt_u64 vcc, %3, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh) : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int WHICH>
int run(const char *name, uint32_t *out) {
    const int rounds = 2000, blocks = 256 * 4;  // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<WHICH>), dim3(blocks), dim3(256), 0, 0, out, rounds);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double inst = (double)blocks * 4 * rounds * 64;  // wave-instructions
    printf("%-18s %8.1f G wave-inst/s  = %.3f per SIMD-cycle at 2.4 GHz (1024 SIMDs)\n", name, inst / ms / 1e6, inst / ms / 1e6 / (2.4 * 1024));
    return 0;
}
template <int WHICH>
int run64(const char *name, uint64_t *out) {
    const int rounds = 2000, blocks = 256 * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k64<WHICH>), dim3(blocks), dim3(256), 0, 0, out, rounds);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double inst = (double)blocks * 4 * rounds * 64;
    printf("%-18s %8.1f G wave-inst/s  = %.3f per SIMD-cycle at 2.4 GHz (1024 SIMDs)\n", name, inst / ms / 1e6, inst / ms / 1e6 / (2.4 * 1024));
    return 0;
}

int main() {
    uint32_t *out; CK(hipMalloc(&out, 256 * 4 * 256 * 8));
    run<0>("v_add_u32", out); run<1>("v_mul_lo_u32", out); run<2>("v_mul_hi_u32", out); run<3>("v_mul_u32_u24", out);
    run<4>("v_mad_u32_u24", out); run<5>("v_lshl_add_u32", out); run<6>("v_pk_min_u16", out); run<7>("v_xor_b32", out);
    run<8>("v_bfrev_b32", out); run<9>("v_perm_b32", out); run<10>("v_bcnt_u32_b32", out); run<11>("v_cndmask_b32", out);
    run<12>("v_mov_b32_dpp", out); run<13>("ds_bpermute_b32", out);
    run64<0>("v_lshlrev_b64", (uint64_t *)out); run64<1>("v_cmp_lt_u64", (uint64_t *)out);
    return 0;
}
