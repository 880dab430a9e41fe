// tools/microbench/valu_rates.hip -- issue cost of the FP64 / conversion instructions that the gather and intersection
// kernels are made of, relative to v_fma_f64, on every SIMD at once (8 waves per SIMD, 8 independent chains per wave).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITER = 2000, CHAINS = 8;

#define KERNEL_DD(name, ASM)                                                                             \
__global__ void __launch_bounds__(256) name(double *out, double seed) {                                  \
    double v[CHAINS];                                                                                    \
    for (int c = 0; c < CHAINS; c++) v[c] = seed + c + threadIdx.x * 1e-3;                               \
    float vf[CHAINS];                                                                                    \
    for (int c = 0; c < CHAINS; c++) vf[c] = (float)v[c];                                                \
    double k = seed * 0.999; float kf = (float)k;                                                        \
    for (int i = 0; i < ITER; i++) {                                                                     \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(v[c]), "+v"(vf[c]) : "v"(k), "v"(kf)); \
    }                                                                                                    \
    double s = 0; for (int c = 0; c < CHAINS; c++) s += v[c] + vf[c];                                    \
    if (s == 12345.678) out[threadIdx.x] = s;                                                            \
}
// double -> 32-bit and back (two instructions per step where a single one would change type)
KERNEL_DD(k_fma_f64, "v_fma_f64 %0, %0, %2, %2")
KERNEL_DD(k_mul_f64, "v_mul_f64 %0, %0, %2")
KERNEL_DD(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL_DD(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_DD(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_DD(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_DD(k_trunc_f64, "v_trunc_f64 %0, %0")
KERNEL_DD(k_floor_f64, "v_floor_f64 %0, %0")
KERNEL_DD(k_ldexp_f64, "v_ldexp_f64 %0, %0, %3")
KERNEL_DD(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %2, %2")
KERNEL_DD(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %2, %2")
KERNEL_DD(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %2, %0")
KERNEL_DD(k_min_f64, "v_min_f64 %0, %0, %2")
KERNEL_DD(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %2")
KERNEL_DD(k_mov_b64, "v_mov_b64 %0, %2")
KERNEL_DD(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %2")
KERNEL_DD(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %3")
KERNEL_DD(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %2")
KERNEL_DD(k_cvt_u32_f64, "v_cvt_u32_f64 %1, %2")
KERNEL_DD(k_cvt_i32_f64, "v_cvt_i32_f64 %1, %2")
KERNEL_DD(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %3")
KERNEL_DD(k_fma_f32, "v_fma_f32 %1, %1, %3, %3")
KERNEL_DD(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %2")
KERNEL_DD(k_add_u32, "v_add_u32 %1, %1, %3")
KERNEL_DD(k_rcp_f32, "v_rcp_f32 %1, %1")
KERNEL_DD(k_sqrt_f32, "v_sqrt_f32 %1, %1")
KERNEL_DD(k_lshrrev_b64, "v_lshrrev_b64 %0, 1, %0")
KERNEL_DD(k_mul_lo_u32, "v_mul_lo_u32 %1, %1, %3")
KERNEL_DD(k_bfe_u32, "v_bfe_u32 %1, %1, 8, 8")
KERNEL_DD(k_cndmask, "v_cndmask_b32 %1, %1, %3, vcc")
KERNEL_DD(k_cndmask_e32, "v_cndmask_b32_e32 %1, %1, %3, vcc")
KERNEL_DD(k_cndmask_sgpr, "v_cndmask_b32_e64 %1, %1, %3, s[10:11]")
KERNEL_DD(k_cndmask_nodep, "v_cndmask_b32_e32 %1, %3, %3, vcc")
KERNEL_DD(k_mov_b32, "v_mov_b32 %1, %3")
KERNEL_DD(k_and_b32, "v_and_b32 %1, %1, %3")
KERNEL_DD(k_max_f64, "v_max_f64 %0, %0, %2")
KERNEL_DD(k_cmp_cnd, "v_cmp_lt_f64 vcc, %0, %2\n v_cndmask_b32_e32 %1, %1, %3, vcc")

struct Entry { const char *name; void (*fn)(double *, double); };

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * 8;        // 8 blocks x 4 waves = 8 waves per SIMD
    double *out; CK(hipMalloc(&out, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<Entry> es = {
#define E(n) {#n, n}
        E(k_fma_f64), E(k_mul_f64), E(k_add_f64), E(k_min_f64), E(k_cmp_f64), E(k_mov_b64), E(k_lshl_add_u64), E(k_rcp_f64), E(k_rsq_f64),
        E(k_sqrt_f64), E(k_trunc_f64), E(k_floor_f64), E(k_ldexp_f64), E(k_div_scale_f64), E(k_div_fmas_f64), E(k_div_fixup_f64),
        E(k_cvt_f64_f32), E(k_cvt_f32_f64), E(k_cvt_u32_f64), E(k_cvt_i32_f64), E(k_cvt_f64_u32), E(k_fma_f32), E(k_pk_fma_f32),
        E(k_add_u32), E(k_rcp_f32), E(k_sqrt_f32), E(k_lshrrev_b64), E(k_mul_lo_u32), E(k_bfe_u32), E(k_cndmask), E(k_cndmask_e32), E(k_cndmask_sgpr), E(k_cndmask_nodep), E(k_mov_b32), E(k_and_b32), E(k_max_f64), E(k_cmp_cnd)};
    double base = 0;
    printf("%d CUs, %d blocks of 256; %d x %d instructions per wave\n", cus, blocks, ITER, CHAINS);
    for (auto &e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.25);       // warm
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.25);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        // wave-instructions per SIMD = 8 waves x ITER x CHAINS; ns per wave-instruction per SIMD
        double ns = (double)ms / 5 * 1e6 / (8.0 * ITER * CHAINS);
        if (base == 0) base = ns;
        printf("%-18s %7.3f ns per wave-instruction per SIMD  = %5.2f x v_fma_f64  (%.1f cycles at 2.4 GHz)\n", e.name + 2, ns, ns / base, ns * 2.4);
    }
    CK(hipFree(out));
    return 0;
}
