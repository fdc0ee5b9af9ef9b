// instruction issue rates on gfx950: cycles per wave64 instruction, one wave per SIMD and four (256-thread block)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(double* out, float* fout, unsigned long long* cycles, int iters)
{
    double a0 = threadIdx.x * 1.0 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = threadIdx.x + 0.5f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const double c = out[0];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 2) { REP8(asm volatile("v_min_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_min_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 3) { REP8(asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));) }
        if (OP == 4) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 5) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0));) }
        if (OP == 6) { REP8(asm volatile("v_cmp_le_f64 vcc, %0, %8\n v_cmp_le_f64 vcc, %1, %8\n v_cmp_le_f64 vcc, %2, %8\n v_cmp_le_f64 vcc, %3, %8\n v_cmp_le_f64 vcc, %4, %8\n v_cmp_le_f64 vcc, %5, %8\n v_cmp_le_f64 vcc, %6, %8\n v_cmp_le_f64 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
        if (OP == 7) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (OP == 8) { REP8(asm volatile("v_max3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    fout[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
template <int OP> void run(const char* name, int threads)
{
    double* out; float* fout; unsigned long long* cyc;
    hipMalloc(&out, 8 * 70000); hipMalloc(&fout, 4 * 70000); hipMalloc(&cyc, 8);
    hipMemset(out, 0, 8 * 70000);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<1, threads>>>(out, fout, cyc, 10);
    hipEventRecord(e0);
    k<OP><<<1, threads>>>(out, fout, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // one workgroup on one CU: `threads`/64 waves over 4 SIMDs
    printf("%-14s threads %4d: %.2f ns per wave-instruction per SIMD (counter: %.2f ticks)\n", name, threads,
           ms * 1e6 / (iters * 64.0) / ((threads / 64 + 3) / 4), (double)c / (iters * 64.0) / ((threads / 64 + 3) / 4));
    hipFree(out); hipFree(fout); hipFree(cyc);
}
int main()
{
    for (int threads : {64, 256, 512}) {
        run<0>("v_add_f64", threads); run<1>("v_mul_f64", threads); run<2>("v_min/max_f64", threads); run<3>("v_cvt_f64_f32", threads);
        run<4>("v_fma_f64", threads); run<5>("v_add_f32", threads); run<6>("v_cmp_le_f64", threads); run<7>("v_pk_fma_f32", threads); run<8>("v_min3/max3_f32", threads);
    }
    return 0;
}
