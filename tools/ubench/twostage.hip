// Developer microbenchmark (VERDICT r4 #1-i): what does a grid barrier cost inside a persistent kernel over ALL 256 CUs when it is built
// in TWO STAGES — one arrival counter per XCD (its 32 workgroups meet on a line of their own: xcdbar.hip measured 1.7 us for that), then
// one 8-way counter the last arriver of every XCD adds to — instead of the single contended counter of gridbar.hip (7-14 us at 256
// workgroups)?  And the number the per-XCD level tier needs: eight INDEPENDENT XCD-local barriers running side by side.
// Every round moves data like a builder level would: each workgroup publishes 256 words, waits, reads another workgroup's 256 words
// (a cross-XCD partner for the chip-wide variants, a same-XCD partner for the local one) and counts stale reads.
//   release = s_waitcnt vmcnt(0) after device-scope (sc1, write-through) stores; no fence, no invalidate; data loads carry sc1.
// Go/no-go for the persistent level tier: <= 3.5 us per round at 256 workgroups (a launch boundary costs ~4.5 us with its cold misses).
// hipcc --offload-arch=gfx950 -O3 -o twostage twostage.hip && ./twostage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
struct Line { unsigned v, pad[31]; };   // 128 B: every synchronisation word on a cache line of its own
struct Ctl {
    Line members[8];   // workgroups that registered on XCD x
    Line total;        // ... on the chip
    Line arrive[8];    // stage 1: arrivals per XCD
    Line garrive;      // stage 2: XCDs that are complete / variant 0: all arrivals
    Line go[8];        // release word per XCD
    Line gogo;         // release word, one for the chip
    Line error, stale;
};
__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned SPIN_MAX = 4000000u;
// VAR 0: one counter, everybody polls it (gridbar.hip's barrier, the control)
// VAR 1: one counter, the last arriver publishes the round on ONE release word everybody polls
// VAR 2: two stages, ONE release word
// VAR 3: two stages, one release word per XCD (written by the last XCD's last arriver)
// VAR 4: eight independent XCD-local barriers (no second stage): the per-XCD level tier's barrier, all XCDs busy at once
template <int VAR, int SLEEP>
__global__ __launch_bounds__(256) void k_bar(Ctl* c, unsigned* data, unsigned slots, unsigned rounds) {
    __shared__ unsigned s_rank, s_lrank, s_ok, s_nx, s_nt;
    const unsigned x = xcc_id() & 7u;
    if (threadIdx.x == 0) {
        s_lrank = add(&c->members[x].v, 1u);
        s_rank = add(&c->total.v, 1u);
        // registration barrier (not timed apart: once per launch): everybody must know the member counts
        unsigned spins = 0;
        while (ld(&c->total.v) < gridDim.x && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(4);
        s_ok = spins < SPIN_MAX;
        s_nx = ld(&c->members[x].v);
        s_nt = gridDim.x;
    }
    __syncthreads();
    if (!s_ok) { if (threadIdx.x == 0) st(&c->error.v, 1u); return; }
    const unsigned rank = s_rank, lrank = s_lrank, nx = s_nx, nt = s_nt;
    // my slot: chip-wide variants index by global rank; the local variant by (x, lrank)
    const unsigned myslot = VAR == 4 ? x * 64u + lrank : rank;
    unsigned stale = 0;
    for (unsigned r = 0; r < rounds; r++) {
        unsigned* buf = data + (size_t)(r % 3u) * slots;
        st(&buf[myslot * 256u + threadIdx.x], r * 1000003u + myslot * 256u + threadIdx.x);   // device-scope store (write-through)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            bool ok = true;
            if (VAR == 0) {
                add(&c->garrive.v, 1u);
                while (ld(&c->garrive.v) < (r + 1u) * nt && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
            } else if (VAR == 1) {
                const unsigned mine = add(&c->garrive.v, 1u);
                if (mine + 1u == (r + 1u) * nt) st(&c->gogo.v, r + 1u);
                else while (ld(&c->gogo.v) < r + 1u && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
            } else if (VAR == 2 || VAR == 3) {
                const unsigned mine = add(&c->arrive[x].v, 1u);
                bool released = false;
                if (mine + 1u == (r + 1u) * nx) {               // last of this XCD
                    const unsigned g = add(&c->garrive.v, 1u);
                    if (g + 1u == (r + 1u) * 8u) {              // last XCD
                        if (VAR == 2) st(&c->gogo.v, r + 1u);
                        else for (int i = 0; i < 8; i++) st(&c->go[i].v, r + 1u);
                        released = true;
                    }
                }
                if (!released) {
                    const unsigned* w = VAR == 2 ? &c->gogo.v : &c->go[x].v;
                    while (ld(w) < r + 1u && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
                }
            } else {
                const unsigned mine = add(&c->arrive[x].v, 1u);
                if (mine + 1u == (r + 1u) * nx) st(&c->go[x].v, r + 1u);
                else while (ld(&c->go[x].v) < r + 1u && ++spins < SPIN_MAX) __builtin_amdgcn_s_sleep(SLEEP);
            }
            ok = spins < SPIN_MAX;
            s_ok = ok ? 1u : 0u;
            if (!ok) st(&c->error.v, 1u);
        }
        __syncthreads();
        if (!s_ok) return;
        unsigned other;
        if (VAR == 4) other = x * 64u + (lrank + 1u + r % (nx > 1u ? nx - 1u : 1u)) % nx;
        else other = (rank + 1u + r % (nt - 1u)) % nt;   // walks over every other workgroup, most of them on other XCDs
        const unsigned got = ld(&buf[other * 256u + threadIdx.x]);
        if (got != r * 1000003u + other * 256u + threadIdx.x) stale++;
    }
    if (stale) add(&c->stale.v, stale);
}

template <int VAR, int SLEEP> static void run(const char* name, Ctl* c, unsigned* data, unsigned slots, unsigned grid) {
    const unsigned sleep = SLEEP;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned NB = 300;
    float best = 1e9f; Ctl h{};
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(c, 0, sizeof(Ctl));
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_bar<VAR, SLEEP>), dim3(grid), dim3(256), 0, 0, c, data, slots, NB);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost);
        if (h.error.v) break;
    }
    unsigned mn = ~0u, mx = 0;
    for (int i = 0; i < 8; i++) { if (h.members[i].v < mn) mn = h.members[i].v; if (h.members[i].v > mx) mx = h.members[i].v; }
    printf("grid=%4u sleep=%u  %-58s %6.2f us per round   (per XCD %u..%u workgroups, stale %u of %u)%s\n", grid, sleep, name, best * 1e3f / NB, mn, mx,
           h.stale.v, grid * 256u * NB, h.error.v ? "  ** timed out **" : "");
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    Ctl* c; unsigned* data;
    const unsigned slots = 1024u * 256u;
    hipMalloc(&c, sizeof(Ctl)); hipMalloc(&data, (size_t)3 * slots * 4);
    hipMemset(data, 0, (size_t)3 * slots * 4);
    for (unsigned grid : {64u, 128u, 256u, 512u}) {
        run<0, 1>("one counter, everybody polls it (control)", c, data, slots, grid);
        run<1, 1>("one counter + one release word", c, data, slots, grid);
        run<2, 1>("two stages (8 XCD counters + 8-way), one release word", c, data, slots, grid);
        run<3, 1>("two stages, one release word per XCD", c, data, slots, grid);
        run<4, 1>("eight independent XCD-local barriers (no second stage)", c, data, slots, grid);
        run<2, 4>("two stages (8 XCD counters + 8-way), one release word", c, data, slots, grid);
        run<3, 4>("two stages, one release word per XCD", c, data, slots, grid);
        run<4, 4>("eight independent XCD-local barriers (no second stage)", c, data, slots, grid);
    }
    return 0;
}
