// Does a wave64 VALU instruction whose upper 32 lanes are inactive issue faster on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void k(T* out, int active_lanes, int iters) {
    const int lane = threadIdx.x & 63;
    if (lane >= active_lanes) return;
    T a = (T)threadIdx.x * (T)1.0001, b = (T)1.0000001, c = (T)0.5, d = (T)0.25;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { a = a * b + c; c = c * b + d; d = d * b + a; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d;
}
template <typename T>
void run(const char* name) {
    T* out; hipMalloc(&out, sizeof(T) * 256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {1, 4}) for (int al : {64, 32, 16}) {
        const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block) -> waves_per_simd per SIMD
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, al, 100);
        hipEventRecord(e0); hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, al, 4000); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s waves/SIMD %d active lanes %2d: %.3f ms  (%.2f cycles per VALU instr per wave at 2.4 GHz)\n", name, waves_per_simd, al, ms,
               ms * 1e-3 * 2.4e9 / (4000.0 * 48 * waves_per_simd));
    }
}
int main() { run<float>("f32"); run<double>("f64"); return 0; }
