// What a launch costs when it does nothing: back-to-back empty kernels as the nodes of one HIP graph (the way bench.py
// replays its steps), by grid size, static LDS per block and register budget.  The step kernel's launch is 1,024 blocks
// of 256 threads, ~38 KB of LDS per block, 128 VGPRs.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_floor tools/ubench/launch_floor.hip && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int VGPRS>
__global__ __launch_bounds__(256) void empty_kernel(int* sink, int never) {
    if constexpr (LDS_BYTES > 0) {
        __shared__ int lds[LDS_BYTES / 4];
        if (never) { lds[threadIdx.x] = never; __syncthreads(); sink[threadIdx.x] = lds[(threadIdx.x * 7) % (LDS_BYTES / 4)]; }
    }
    if constexpr (VGPRS > 32) {   // hold VGPRS registers live across a (never taken) branch so the kernel is allocated them
        if (never) {
            int v[VGPRS - 16];
#pragma unroll
            for (int i = 0; i < VGPRS - 16; i++) v[i] = sink[i + threadIdx.x];
            __syncthreads();
            int s = 0;
#pragma unroll
            for (int i = 0; i < VGPRS - 16; i++) s += v[i] * (i + never);
            sink[threadIdx.x] = s;
        }
    }
}

template <int LDS_BYTES, int VGPRS>
double run(int blocks, int nodes, int* sink) {
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nodes; i++) hipLaunchKernelGGL((empty_kernel<LDS_BYTES, VGPRS>), dim3(blocks), dim3(256), 0, st, sink, 0);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    hipStreamDestroy(st);
    return best * 1e3 / nodes;
}

int main() {
    int* sink;
    hipMalloc(&sink, 1 << 20);
    const int nodes = 400;
    printf("us per empty launch, %d nodes of one graph, 256 threads per block (best of 5 replays)\n", nodes);
    printf("%8s %12s %12s %12s %12s\n", "blocks", "0 LDS", "16 KB LDS", "38 KB LDS", "38 KB+128 VGPR");
    for (int blocks : {1, 256, 512, 1024, 2048, 4096}) {
        printf("%8d %12.2f %12.2f %12.2f %12.2f\n", blocks, run<0, 0>(blocks, nodes, sink), run<16384, 0>(blocks, nodes, sink),
               run<38912, 0>(blocks, nodes, sink), run<38912, 128>(blocks, nodes, sink));
    }
    return 0;
}
