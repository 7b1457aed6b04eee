// Developer microbenchmark: a persistent kernel whose workgroups all sit on ONE XCD (one L2) — what does a grid barrier cost there,
// and which release / acquire is enough for the workgroups (other CUs, same L2) to see each other's plain stores?
// Background (DESIGN.md §4 "Why three builder tiers"): a chip-wide grid barrier needs agent-scope fences (L2 write-back + invalidate:
// the XCD L2s are not coherent with each other) and costs 7-14 us for 256 workgroups — more than a kernel boundary.  Inside one XCD
// the L2 is shared, so only the per-CU L1 has to be bypassed or invalidated.
// hipcc --offload-arch=gfx950 -O3 -o xcdbar xcdbar.hip && ./xcdbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

// PROTO 0: agent-scope release / acquire fences (the textbook protocol)
// PROTO 1: release = s_waitcnt vmcnt(0) (stores have reached the L2), acquire = buffer_inv sc1 (L1 + non-coherent L2 lines)
// PROTO 2: release = s_waitcnt vmcnt(0), acquire = buffer_inv sc0
// PROTO 3: release = s_waitcnt vmcnt(0), no invalidate, data loads carry sc1 (agent-coherent load: misses the L1)
// PROTO 4: release = s_waitcnt vmcnt(0), no invalidate, data loads carry sc0
// PROTO 5: release = s_waitcnt vmcnt(0), no invalidate, plain loads (expected: stale reads — the control)
// ATOM 0: every workgroup polls the arrival counter, 1: the last one to arrive publishes the round number on a line of its own and the
// others poll that (workgroup-scope atomics were tried for the counter: other CUs never see them — the barrier times out)
template <int PROTO> __device__ __forceinline__ void release_side() {
    if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int PROTO> __device__ __forceinline__ void acquire_side() {
    if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (PROTO == 1) asm volatile("buffer_inv sc1" ::: "memory");
    else if (PROTO == 2) asm volatile("buffer_inv sc0" ::: "memory");
    else asm volatile("" ::: "memory");
}
template <int PROTO> __device__ __forceinline__ unsigned data_load(const unsigned* p) {
    unsigned v;
    if (PROTO == 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (PROTO == 4) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Ctl { unsigned members, error, stale, pad0[29]; unsigned arrive, pad1[31]; unsigned go, pad2[31]; };   // arrive and go on cache lines of their own

template <int PROTO, int ATOM>
__global__ __launch_bounds__(256) void k_xcd(Ctl* c, unsigned* data, unsigned slots, unsigned W, unsigned rounds, unsigned target, unsigned* where) {
    __shared__ unsigned s_rank, s_ok;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) {
        if (where) where[blockIdx.x] = x;
        s_rank = 0xFFFFFFFFu;
        if (x == target) s_rank = atomicAdd(&c->members, 1u);
    }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= W) return;   // not on the target XCD (or a surplus workgroup)
    unsigned stale = 0;
    for (unsigned r = 0; r < rounds; r++) {
        unsigned* buf = data + (size_t)(r % 3u) * slots;
        buf[rank * 256u + threadIdx.x] = r * 1000003u + rank * 256u + threadIdx.x;   // the "level's" result, plain stores
        release_side<PROTO>();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned goal = (r + 1u) * W;
            unsigned spins = 0, seen;
            if (ATOM == 0) {
                __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((seen = __hip_atomic_load(&c->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < goal && ++spins < 2000000u) __builtin_amdgcn_s_sleep(2);
            } else {
                const unsigned mine = __hip_atomic_fetch_add(&c->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mine + 1u == goal) __hip_atomic_store(&c->go, r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else while (__hip_atomic_load(&c->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r + 1u && ++spins < 2000000u) __builtin_amdgcn_s_sleep(2);
                seen = spins < 2000000u ? goal : 0u;
            }
            s_ok = seen >= goal ? 1u : 0u;
            if (seen < goal) c->error = 1u;
        }
        __syncthreads();
        if (!s_ok) return;
        acquire_side<PROTO>();
        const unsigned other = (rank + 1u + r % (W > 1u ? W - 1u : 1u)) % W;
        const unsigned got = data_load<PROTO>(&buf[other * 256u + threadIdx.x]);
        if (got != r * 1000003u + other * 256u + threadIdx.x) stale++;
    }
    if (stale) atomicAdd(&c->stale, stale);
}

template <int PROTO, int ATOM> static void run(const char* name, Ctl* c, unsigned* data, unsigned slots, unsigned W, unsigned* where) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned NB = 300;
    float best = 1e9f; Ctl h{};
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(c, 0, sizeof(Ctl));
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_xcd<PROTO, ATOM>), dim3(8 * W), dim3(256), 0, 0, c, data, slots, W, NB, 0u, rep == 0 ? where : nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
        if (h.error) break;
    }
    printf("W=%3u  %-46s barrier %-8s: %6.2f us per round, members %u, stale reads %u of %u%s\n", W, name, ATOM ? "go flag" : "counter", best * 1e3f / NB,
           h.members, h.stale, W * 256u * NB, h.error ? "  ** barrier timed out **" : "");
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    Ctl* c; unsigned* data; unsigned* where;
    const unsigned slots = 256u * 256u;
    hipMalloc(&c, sizeof(Ctl)); hipMalloc(&data, (size_t)3 * slots * 4); hipMalloc(&where, 4096 * 4);
    hipMemset(data, 0, (size_t)3 * slots * 4);
    // placement: which XCD does block b land on?
    {
        hipMemset(c, 0, sizeof(Ctl)); hipMemset(where, 0xFF, 4096 * 4);
        hipLaunchKernelGGL((k_xcd<0, 0>), dim3(2048), dim3(256), 0, 0, c, data, slots, 0u, 0u, 0u, where);
        hipDeviceSynchronize();
        std::vector<unsigned> w(2048); hipMemcpy(w.data(), where, 2048 * 4, hipMemcpyDeviceToHost);
        unsigned agree = 0, hist[16] = {0};
        for (unsigned b = 0; b < 2048; b++) { if (w[b] == b % 8u) agree++; if (w[b] < 16) hist[w[b]]++; }
        printf("placement: %u of 2048 blocks on XCD b %% 8; per XCD:", agree);
        for (int i = 0; i < 8; i++) printf(" %u", hist[i]);
        printf("\n");
    }
    for (unsigned W : {32u, 64u, 128u, 256u}) {
        run<0, 0>("agent-scope release / acquire fences", c, data, slots, W, where);
        run<0, 1>("agent-scope release / acquire fences", c, data, slots, W, where);
        run<1, 0>("waitcnt + buffer_inv sc1", c, data, slots, W, where);
        run<1, 1>("waitcnt + buffer_inv sc1", c, data, slots, W, where);
        run<2, 1>("waitcnt + buffer_inv sc0", c, data, slots, W, where);
        run<3, 1>("waitcnt, no invalidate, sc1 loads", c, data, slots, W, where);
        run<4, 1>("waitcnt, no invalidate, sc0 loads", c, data, slots, W, where);
        run<5, 1>("waitcnt, no invalidate, plain loads (control)", c, data, slots, W, where);
    }
    return 0;
}
