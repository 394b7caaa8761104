// Dependent-issue latency of VALU instructions on gfx950: one chain per wave, 1 / 2 / 4 waves per SIMD.
// cycles per instruction per wave = what a wave that has only ONE instruction ready at a time costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN 512
__device__ inline unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double b, int ib) {
    double a = (double)threadIdx.x * 1.0001;
    int x = threadIdx.x * 7 + 3;
    const int lim = 1 << 30;
    const unsigned long long t0 = now();
#pragma unroll
    for (int i = 0; i < CHAIN; i++) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
        else if (OP == 1) asm volatile("v_add_u32 %0, %0, %1\n\tv_med3_i32 %0, %0, 0, %2" : "+v"(x) : "v"(ib), "v"(lim));
        else if (OP == 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(ib));
        else if (OP == 3) asm volatile("v_add_f64 %0, %0, %2\n\tv_add_u32 %1, %1, %3\n\tv_med3_i32 %1, %1, 0, %4" : "+v"(a), "+v"(x) : "v"(b), "v"(ib), "v"(lim));
        else if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
        else if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %0, %0, 1" : "+v"(x) : "v"(ib));
        else if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(ib));
        else if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(ib), "v"(lim));
    }
    const unsigned long long t1 = now();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char* name, int instr) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * 256 * 1024 * 8); hipMalloc(&cyc, 8 * 4 * 8192);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1e-9, 1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1e-9, 1);
        hipDeviceSynchronize();
        static unsigned long long h[4 * 8192];
        hipMemcpy(h, cyc, 8 * 4 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 4 * blocks; i++) s += h[i];
        // s_memtime counts at 100 MHz on gfx9 (constant-frequency counter): convert with the wall clock instead
        printf("%-28s waves/SIMD %d: %.1f memtime cycles per chain step (%d instr per step)\n", name, wps, s / (4 * blocks) / CHAIN, instr);
    }
}
__global__ void kt(unsigned long long* o) {   // ticks of s_memtime vs s_memrealtime
    const unsigned long long a0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = threadIdx.x;
    for (int i = 0; i < 20000; i++) a = a * 1.000001 + 0.5;
    const unsigned long long a1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[0] = a1 - a0; o[1] = r1 - r0; o[2] = (unsigned long long)a; }
}
int main() {
    unsigned long long* o; hipMalloc(&o, 64); hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, o); hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("s_memtime %llu ticks over %llu s_memrealtime ticks (100 MHz) -> %.1f MHz\n", h[0], h[1], 100.0 * h[0] / h[1]);
    run<0>("v_add_f64", 1); run<4>("v_fma_f64", 1); run<1>("v_add_u32 + v_med3_i32", 2); run<2>("v_mov_dpp + v_add_u32", 2);
    run<3>("f64 add || add+med3", 3); run<5>("v_mul_lo + add", 2); run<6>("v_add_u32", 1); run<7>("v_perm_b32", 1);
    return 0;
}
