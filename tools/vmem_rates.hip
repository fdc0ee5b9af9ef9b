// Vector-memory instruction rates on gfx950: time per wave64 global_load instruction on one CU, by the
// number of active lanes and by how the lanes' addresses are spread (all L1 hits: the buffer is 8 KB).
// One workgroup (one CU), eight independent loads per loop iteration and wave.
//   hipcc --offload-arch=gfx950 -O3 tools/vmem_rates.hip -o /tmp/vmem_rates && /tmp/vmem_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WIDTH>        // dwords per lane and load: 1 or 4
__global__ void k(const float* __restrict__ buf, float* out, int iters, int active, int stride_bytes)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    if (lane < active) {
        const char* p = reinterpret_cast<const char*>(buf) + (size_t)lane * stride_bytes;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (WIDTH == 4) {
                    f4 v = *reinterpret_cast<const volatile f4*>(p + ((u * 16) & 63));
                    acc += v.x + v.w;
                } else {
                    acc += *reinterpret_cast<const volatile float*>(p + ((u * 16) & 63));
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WIDTH>
void run(int threads, int active, int stride, const char* what)
{
    float *buf, *out;
    hipMalloc(&buf, 16384); hipMalloc(&out, 4 * 2048);
    hipMemset(buf, 0, 16384);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<WIDTH><<<1, threads>>>(buf, out, 10, active, stride);
    hipEventRecord(e0);
    k<WIDTH><<<1, threads>>>(buf, out, iters, active, stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 8.0 * (threads / 64);
    printf("dwordx%d  %4d threads  %2d active lanes  %-28s %7.2f ns per wave-instruction on the CU (%.1f cycles at 2.4 GHz)\n",
           WIDTH, threads, active, what, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
    hipFree(buf); hipFree(out);
}

int main()
{
    for (int threads : {256, 1024}) {
        for (int active : {64, 32, 16, 4, 1}) {
            run<4>(threads, active, 16, "consecutive 16-byte pieces");
            run<4>(threads, active, 128, "one 128-byte line per lane");
            run<4>(threads, active, 0, "one address");
            run<1>(threads, active, 128, "one 128-byte line per lane");
        }
    }
    return 0;
}
