// Developer microbenchmark: does a hipGraph shorten the gap between DEPENDENT kernels on MI355X?  The step of bench.py is a chain of 16
// dependent launches with ≈ 2.5 µs between the end of one kernel and the start of the next (rocprofv3 kernel trace).  The same chain of
// small kernels (grid G workgroups, each a dependent load + an atomic) is timed (a) launched on a stream, (b) captured once and replayed
// with hipGraphLaunch; once with empty kernels (the stream case is then bound by the host's launch rate) and once with 8 µs of work in every
// kernel (the host runs ahead: what remains is the GPU's own gap between dependent dispatches).   hipcc --offload-arch=gfx950 -O3 -o graphgap graphgap.hip && ./graphgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_link(const unsigned* __restrict__ tab, unsigned mask, unsigned* sink, unsigned link, unsigned spin) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);   // (100 MHz ticks: the kernel's own duration, so that the host runs ahead)
    unsigned x = tab[(blockIdx.x * 256 + threadIdx.x + link) & mask];
    x = tab[(x + threadIdx.x) & mask];
    if (threadIdx.x == 0) atomicAdd(&sink[16 + ((x + link) & 1023)], 1u);
}

int main() {
    const unsigned mask = (1u << 20) - 1;
    unsigned *tab, *sink;
    hipMalloc(&tab, (size_t)(mask + 1) * 4); hipMalloc(&sink, 8192);
    std::vector<unsigned> h(mask + 1);
    for (unsigned i = 0; i <= mask; i++) h[i] = (i * 2654435761u) & mask;
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(sink, 0, 8192);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int CHAIN = 16, REPS = 200;
    for (unsigned spin : {0u, 800u})
    for (unsigned G : {1u, 256u, 1024u}) {
        // (a) stream
        float best_s = 1e9f, best_g = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0, st);
            for (int r = 0; r < REPS; r++)
                for (int k = 0; k < CHAIN; k++) hipLaunchKernelGGL(k_link, dim3(G), dim3(256), 0, st, tab, mask, sink, (unsigned)k, spin);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_s) best_s = ms;
        }
        // (b) graph of one chain, launched REPS times
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < CHAIN; k++) hipLaunchKernelGGL(k_link, dim3(G), dim3(256), 0, st, tab, mask, sink, (unsigned)k, spin);
        hipStreamEndCapture(st, &g);
        hipError_t rc = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (rc != hipSuccess) { printf("graph instantiate failed: %s\n", hipGetErrorString(rc)); return 1; }
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0, st);
            for (int r = 0; r < REPS; r++) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_g) best_g = ms;
        }
        printf("grid %4u x 256, %u us of work per kernel: chain of %d dependent kernels: stream %.2f us per kernel, graph %.2f us per kernel\n", G, spin / 100u, CHAIN,
               best_s * 1e3f / (REPS * CHAIN), best_g * 1e3f / (REPS * CHAIN));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
