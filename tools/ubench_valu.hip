// ubench_valu.hip -- issue cost of the VALU instructions the node step is made of, on the GPU at hand.
// One kernel per instruction: 64 copies of it on independent registers per loop trip, ITER trips, W waves per SIMD on every
// CU; prints cycles per wave-instruction per SIMD (s_memtime of one wave over the loop, and wall time over all waves).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 4096
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, ASM)                                                                                      \
    __global__ void __launch_bounds__(256) NAME(float* out, float a, float b) {                               \
        float r[16];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; i++) r[i] = a + (float)(threadIdx.x + i);                    \
        uint32_t s = __float_as_uint(b);                                                                        \
        for (int it = 0; it < ITER; it++) {                                                                    \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                    \
                _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(s)); \
            }                                                                                                   \
        }                                                                                                       \
        float acc = 0;                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 16; i++) acc += r[i];                                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                       \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %1")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]")
KERNEL(k_fma_mix_hi, "v_fma_mix_f32 %0, %0, %1, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_bfi, "v_bfi_b32 %0, %2, %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %2")
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %1")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
KERNEL(k_cmp_cndmask, "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmp_addc, "v_cmp_le_f32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 5, 3")
KERNEL(k_lshlrev, "v_lshlrev_b32 %0, %2, %0")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %2")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %2")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %2, %2")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %2")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL(k_or, "v_or_b32 %0, %0, %2")
KERNEL(k_xor, "v_xor_b32 %0, %0, %2")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %2")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_lshrrev, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_cmp, "v_cmp_le_f32 vcc, %0, %1")
KERNEL(k_cmp_sgpr, "v_cmp_le_f32 s[20:21], %0, %1")
KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %0")
KERNEL(k_and_e64, "v_and_b32_e64 %0, %0, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %1")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %1")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")

typedef void (*kern_t)(float*, float, float);
struct Entry { const char* name; kern_t k; int per; };

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out; (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    Entry es[] = { { "v_fma_f32", k_fma, 1 }, { "v_fma_mix_f32 (lo)", k_fma_mix, 1 }, { "v_fma_mix_f32 (hi)", k_fma_mix_hi, 1 }, { "v_perm_b32", k_perm, 1 },
                   { "v_cvt_f32_ubyte1", k_cvt_ub, 1 }, { "v_bfi_b32", k_bfi, 1 }, { "v_max3_f32", k_max3, 1 }, { "v_max_f32", k_max, 1 }, { "v_and_b32", k_and, 1 },
                   { "v_cmp + v_cndmask", k_cmp_cndmask, 2 }, { "v_cmp + v_addc_co", k_cmp_addc, 2 }, { "v_bfe_u32", k_bfe, 1 }, { "v_lshlrev_b32", k_lshlrev, 1 },
                   { "v_or3_b32", k_or3, 1 }, { "v_ldexp_f32", k_ldexp, 1 }, { "v_mul_u32_u24", k_mul_u24, 1 }, { "v_mad_u32_u24", k_mad_u24, 1 },
                   { "v_bcnt_u32_b32", k_bcnt, 1 }, { "v_ffbh_u32", k_ffbh, 1 },
                   { "v_lshl_or_b32", k_lshl_or, 1 }, { "v_and_or_b32", k_and_or, 1 }, { "v_pk_fma_f16", k_pk_fma_f16, 1 }, { "v_pk_max_f16", k_pk_max_f16, 1 }, { "v_pk_add_f16", k_pk_add_f16, 1 },
                   { "v_or_b32", k_or, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_add_u32", k_add_u32, 1 }, { "v_add_f32", k_add_f32, 1 }, { "v_mul_f32", k_mul_f32, 1 }, { "v_min_f32", k_min_f32, 1 },
                   { "v_mov_b32", k_mov, 1 }, { "v_cndmask_b32", k_cndmask, 1 }, { "v_lshrrev_b32", k_lshrrev, 1 }, { "v_cmp_le_f32 vcc", k_cmp, 1 }, { "v_cmp_le_f32 sgpr", k_cmp_sgpr, 1 },
                   { "v_cvt_f32_f16", k_cvt_f16, 1 }, { "v_and_b32_e64", k_and_e64, 1 }, { "v_fmac_f32", k_fmac, 1 }, { "v_min3_f32", k_min3, 1 }, { "v_bitop3_b32", k_bitop3, 1 } };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%d CUs, clock %d kHz\n", cus, p.clockRate);
    printf("%-22s %14s %14s %14s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "5 waves/SIMD  (cycles per wave-instruction per SIMD at the reported clock)");
    for (auto& e : es) {
        printf("%-22s", e.name);
        for (int blocks_per_cu : { 1, 2, 5 }) {   // 256 threads = 4 waves = one per SIMD
            const int grid = cus * blocks_per_cu;
            e.k<<<grid, 256>>>(out, 1.0f, 2.0f);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            e.k<<<grid, 256>>>(out, 1.0f, 2.0f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)ITER * 64.0 * e.per * blocks_per_cu;   // per SIMD
            const double cyc = ms * 1e-3 * (double)p.clockRate * 1e3 / insts;
            printf(" %14.2f", cyc);
        }
        printf("\n");
    }
    return 0;
}
