// Microbenchmark (design input): issue cost of the integer VALU ops k_probe_part leans on, wave64 on gfx950.
// 256 CUs x 8 blocks x 256 threads; every thread runs N iterations of 8 independent chains of one op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int N = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int sh) {
    uint32_t a[8];
    uint64_t b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 8 + i; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u); }
    for (int it = 0; it < N; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = (a[i] * a[(i + 1) & 7]) ^ 0x9E3779B1u;        // v_mul_lo_u32 (+xor), data-dependent
            if (OP == 1) a[i] = __umul24(a[i], a[(i + 1) & 7]) ^ 0x9E3779B1u;  // v_mul_u32_u24 (+xor), data-dependent
            if (OP == 2) b[i] = (b[i] << (sh & 63)) + 1;                     // v_lshlrev_b64 variable
            if (OP == 3) a[i] = (a[i] << (sh & 31)) + 1u;                    // v_lshlrev_b32 variable (+add)
            if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], sh) ^ 1u;  // v_alignbit_b32
            if (OP == 5) a[i] = __brev(a[i]) + 1u;                           // v_bfrev_b32
            if (OP == 6) b[i] = (b[i] < b[(i + 1) & 7] ? b[i] : b[(i + 1) & 7]) + 3;  // 64-bit min
            if (OP == 7) a[i] = __builtin_amdgcn_readlane(a[i], 5) + a[i];   // v_readlane
            if (OP == 8) a[i] = (uint32_t)__popcll(__ballot(a[i] & 1)) + a[i];  // ballot + s_bcnt
            if (OP == 9) a[i] = (a[i] ^ (a[i] >> 15)) + 1u;                  // shift+xor (+add)
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)b[i];
    if (r == 0x1234567) out[0] = r;
}

template <int OP>
int run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<OP>), dim3(256 * 8), dim3(256), 0, 0, out, 12345u, 3);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // wave-instructions of the op per SIMD: 2048 blocks * 4 waves / 1024 SIMDs = 8 waves per SIMD
    const double ops_per_simd = 8.0 * N * 8;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-op per SIMD (~%.1f cycles at 2.3 GHz, incl. the companion add)\n", name, ms,
           ms * 1e6 / ops_per_simd, ms * 1e6 / ops_per_simd * 2.3);
    return 0;
}

int main() {
    uint32_t *out; CK(hipMalloc(&out, 4));
    run<0>("v_mul_lo_u32 + xor", out); run<1>("v_mul_u32_u24 + xor", out); run<2>("v_lshlrev_b64 var + add64", out);
    run<3>("v_lshlrev_b32 var + add", out); run<4>("v_alignbit_b32 + xor", out); run<5>("v_bfrev_b32 + add", out);
    run<6>("min u64 + add64", out); run<7>("v_readlane + add", out); run<8>("ballot + bcnt + add", out); run<9>("shift, xor, add", out);
    return 0;
}
