// Developer microbenchmark: what does a grid-wide barrier cost on MI355X inside a persistent kernel (all workgroups resident)?
// It decides whether the builder's level loop (two launches per level, ~4.5 us floor each) could live in one kernel.
// hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: __threadfence() + relaxed atomics   MODE 1: release / acquire atomics at agent scope, no separate fence
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, unsigned n_bar, float* data, unsigned dirty_per_thread) {
    const unsigned G = gridDim.x;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned b = 0; b < n_bar; b++) {
        for (unsigned j = 0; j < dirty_per_thread; j++) data[(size_t)j * G * 256 + tid] = (float)b;   // work of the "level"
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (b + 1) * G;
            if (MODE == 0) {
                __threadfence();
                atomicAdd(ctr, 1u);
                while (atomicAdd(ctr, 0u) < target) __builtin_amdgcn_s_sleep(1);
                __threadfence();
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1]++; }

// what does a small dependent kernel cost as a function of its grid and of the dependent global round trips inside it?
__global__ __launch_bounds__(256) void k_chain(const unsigned* __restrict__ tab, unsigned mask, int depth, int atomics,
                                               unsigned* sink) {
    unsigned x = tab[(blockIdx.x * 256 + threadIdx.x) & mask];
    for (int d = 0; d < depth; d++) x = tab[(x + threadIdx.x) & mask];
    for (int a = 0; a < atomics; a++) atomicAdd(&sink[16 + ((x + a) & 1023)], 1u);
    if (x == 0xFFFFFFFFu) sink[2] = x;
}

int main() {
    unsigned* ctr; float* data;
    hipMalloc(&ctr, 64); hipMalloc(&data, (size_t)2048 * 256 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned NB = 200;
    for (unsigned G : {64u, 256u, 512u, 1024u}) {
        for (unsigned dirty : {0u, 4u}) {
            for (int mode = 0; mode < 2; mode++) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    hipMemset(ctr, 0, 64);
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(G), dim3(256), 0, 0, ctr, NB, data, dirty);
                    else hipLaunchKernelGGL(k_bar<1>, dim3(G), dim3(256), 0, 0, ctr, NB, data, dirty);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("G=%4u blocks  dirty %u x 4 B/thread  mode %d: %.2f us per barrier\n", G, dirty, mode, best * 1e3f / NB);
            }
        }
    }
    // for comparison: dependent tiny launches on one stream
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (unsigned i = 0; i < NB; i++) hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, 0, ctr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("dependent 256-block launches: %.2f us each\n", best * 1e3f / NB);
    {
        unsigned* tab; unsigned* sink;
        const unsigned M = 1u << 20;
        hipMalloc(&tab, M * 4); hipMalloc(&sink, 8192);
        std::vector<unsigned> h(M);
        for (unsigned i = 0; i < M; i++) h[i] = (i * 2654435761u) >> 8;
        hipMemcpy(tab, h.data(), M * 4, hipMemcpyHostToDevice); hipMemset(sink, 0, 8192);
        for (unsigned G : {1u, 256u, 700u, 2048u}) for (int depth : {0, 1, 2, 4}) for (int at : {0, 1}) {
            float bestc = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                for (unsigned i = 0; i < NB; i++) hipLaunchKernelGGL(k_chain, dim3(G), dim3(256), 0, 0, tab, M - 1, depth, at, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < bestc) bestc = ms;
            }
            printf("k_chain G=%4u depth %d atomics %d: %.2f us per dependent launch\n", G, depth, at, bestc * 1e3f / NB);
        }
    }
    return 0;
}
