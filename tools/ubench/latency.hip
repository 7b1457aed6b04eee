// Developer microbenchmark: dependent-load latency of the gather pattern traversal uses.
// hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

struct __attribute__((aligned(32))) Node { float a[3]; unsigned nxt; float b[3]; unsigned w; };

template <int MODE>  // 0: global chase, 1: LDS chase
__global__ void chase(const Node* __restrict__ nodes, unsigned n, unsigned steps, unsigned stride_lanes, unsigned* out,
                      long long* cycles) {
    __shared__ float4 lds_lo[4096], lds_hi[4096];
    if (MODE == 1) {
        for (unsigned q = threadIdx.x; q < 4096; q += blockDim.x) {
            const float4* p = reinterpret_cast<const float4*>(nodes + q);
            lds_lo[q] = p[0]; lds_hi[q] = p[1];
        }
        __syncthreads();
    }
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned i = (tid * stride_lanes * 977u) % n;
    float acc = 0;
    long long t0 = clock64();
    for (unsigned s = 0; s < steps; s++) {
        float4 lo, hi;
        if (MODE == 0) { const float4* p = reinterpret_cast<const float4*>(nodes + i); lo = p[0]; hi = p[1]; }
        else { lo = lds_lo[i & 4095]; hi = lds_hi[i & 4095]; }
        acc += lo.x + hi.y;
        i = __float_as_uint(lo.w);
    }
    long long t1 = clock64();
    out[tid] = i + (unsigned)acc;
    if (tid == 0) *cycles = t1 - t0;
}

int main() {
    for (unsigned n : {512u, 4096u, 16384u, 262144u, 4u << 20}) {
        std::vector<Node> h(n);
        std::mt19937 rng(1);
        std::vector<unsigned> perm(n);
        for (unsigned i = 0; i < n; i++) perm[i] = i;
        for (unsigned i = n - 1; i > 0; i--) std::swap(perm[i], perm[rng() % (i + 1)]);
        for (unsigned i = 0; i < n; i++) { h[perm[i]].nxt = perm[(i + 1) % n]; h[perm[i]].a[0] = 1.f; h[perm[i]].b[1] = 2.f; }
        Node* d; unsigned* out; long long* cyc;
        hipMalloc(&d, n * sizeof(Node)); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
        hipMemcpy(d, h.data(), n * sizeof(Node), hipMemcpyHostToDevice);
        const unsigned steps = 2000;
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 1 && n != 4096) continue;
            for (int waves_per_cu : {1, 4, 16, 32}) {
                for (unsigned spread : {0u, 1u}) {   // 0: all lanes follow the same chain, 1: every lane its own
                    int blocks = 256 * waves_per_cu / 4;   // 256-thread blocks
                    if (blocks < 1) blocks = 1;
                    dim3 g(waves_per_cu == 1 ? 256 : blocks), b(waves_per_cu == 1 ? 64 : 256);
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    for (int rep = 0; rep < 2; rep++) {
                        hipEventRecord(e0);
                        if (mode == 0) hipLaunchKernelGGL(chase<0>, g, b, 0, 0, d, n, steps, spread, out, cyc);
                        else hipLaunchKernelGGL(chase<1>, g, b, 0, 0, d, n, steps, spread, out, cyc);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                    }
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                    printf("n=%8u (%6.1f KB) mode=%s waves/CU=%2d %s: %7.1f ns/step (event)  %6.0f clk/step\n", n,
                           n * 32 / 1024.0, mode ? "LDS " : "glob", waves_per_cu, spread ? "lanes-diverge" : "lanes-same   ",
                           ms * 1e6 / steps, (double)c / steps);
                }
            }
        }
        hipFree(d); hipFree(out); hipFree(cyc);
    }
    return 0;
}
