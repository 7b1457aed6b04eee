/* Plain C consumer of include/bvh_mi355x.h: build → flatten → traverse → fetch, printing everything a test needs
 * to compare with the CPU checker.  Proves the drop-in boundary is a C ABI (no Python, no C++ types).
 * usage: abi_roundtrip <n_cubes_per_axis> [ranks]   — boxes of the reference's generate_aligned_boxes pattern, extended to 3-D;
 * ranks: ctxs of the multi-GPU part (default: one per device) */
#include <stdio.h>
#include <stdlib.h>

#include "bvh_mi355x.h"

#define CHECK(x) do { int _rc = (x); if (_rc != BVHGPU_OK) { fprintf(stderr, "%s -> %d (%s): %s\n", #x, _rc, \
    bvhgpu_status_string(_rc), bvhgpu_last_error(ctx)); return 2; } } while (0)

int main(int argc, char** argv) {
    int m = argc > 1 ? atoi(argv[1]) : 5;
    size_t n = (size_t)m * m * m;
    float* aabbs = (float*)malloc(n * 6 * sizeof(float));
    size_t k = 0;
    for (int x = 0; x < m; x++) for (int y = 0; y < m; y++) for (int z = 0; z < m; z++, k++) {
        float c[3] = {(float)(2 * x), (float)(2 * y), (float)(2 * z)};
        for (int a = 0; a < 3; a++) { aabbs[6 * k + a] = c[a] + -0.5f; aabbs[6 * k + 3 + a] = c[a] + 0.5f; }
    }
    bvhgpu_ctx* ctx = NULL;
    int ndev = 0;
    bvhgpu_device_count(&ndev);
    if (ndev <= 0) { fprintf(stderr, "no HIP device\n"); return 3; }
    CHECK(bvhgpu_create(0, NULL, &ctx));
    bvhgpu_tree* tree = NULL;
    CHECK(bvhgpu_build_flat_f32(ctx, aabbs, n, BVHGPU_HOST, &tree));      /* FlatBvh::build */
    size_t ns, nn, nf; int dt;
    CHECK(bvhgpu_tree_info(tree, &dt, &ns, &nn, &nf));
    printf("shapes %zu nodes %zu flat %zu dtype %d\n", ns, nn, nf, dt);
    /* two rays: along the x axis through the row y = z = 0, and a diagonal */
    float o[6] = {-5.f, 0.f, 0.f, -3.f, -3.f, -3.f};
    float d[6] = {1.f, 0.f, 0.f, 1.f, 1.f, 1.f};
    bvhgpu_ray_f32 rays[2];
    CHECK(bvhgpu_rays_new_f32(ctx, o, d, 2, BVHGPU_HOST, rays, BVHGPU_HOST));   /* Ray::new */
    bvhgpu_hits* hits = NULL;
    CHECK(bvhgpu_traverse_f32(tree, rays, 2, BVHGPU_HOST, BVHGPU_TRAVERSE_STATS, &hits));
    uint64_t total; bvhgpu_traverse_stats st; size_t nr;
    CHECK(bvhgpu_hits_info(hits, &nr, &total, &st));
    uint32_t off[3];
    uint32_t* idx = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
    CHECK(bvhgpu_hits_fetch(hits, off, idx, NULL, BVHGPU_HOST));
    printf("total %llu visited %llu\n", (unsigned long long)total, (unsigned long long)st.visited);
    for (int r = 0; r < 2; r++) {
        printf("ray %d:", r);
        for (uint32_t j = off[r]; j < off[r + 1]; j++) printf(" %u", idx[j]);
        printf("\n");
    }
    /* nearest_to with the UnitBox distance */
    float q[3] = {2.2f, 0.1f, 3.9f};
    uint32_t ns_; float nd_;
    CHECK(bvhgpu_nearest_f32(tree, q, 1, BVHGPU_HOST, 0, &ns_, &nd_));
    printf("nearest %u %.6f\n", ns_, nd_);
    /* the multi-GPU exchange step: ONE process, one ctx per device (ncclCommInitAll) — every device this box has, i.e. one
     * here and eight on a node; `abi_roundtrip <m> <K>` makes K ctxs (ctx i on device i % ndev: more ctxs than devices needs the
     * tests' stand-in library, BVHGPU_RCCL_LIB + BVHGPU_RCCL_SHARED_DEVICE).  Rank 0 holds the tree; both forms of the
     * broadcast; then the asynchronous step of a frame loop: rebuild_flat_async on the root, bcast_known on every rank, every
     * rank walks its shard of ONE ray stream (rays [r T / K, (r + 1) T / K), generated in place) and is waited for once.
     * The per-rank hit counts and a checksum over the concatenated CSR are printed for the checker. */
    int K = argc > 2 ? atoi(argv[2]) : ndev;
    if (K < 1 || K > 64) { fprintf(stderr, "bad rank count\n"); return 3; }
    bvhgpu_ctx* ctxs[64];
    ctxs[0] = ctx;
    for (int i = 1; i < K; i++) CHECK(bvhgpu_create(i % ndev, NULL, &ctxs[i]));
    bvhgpu_comm* comm = NULL;
    CHECK(bvhgpu_comm_init_all(ctxs, K, &comm));
    int nranks = -1, first = -1, nlocal = -1;
    CHECK(bvhgpu_comm_info(comm, &nranks, &first, &nlocal));
    bvhgpu_tree* trees[64];
    for (int i = 0; i < K; i++) trees[i] = i == 0 ? tree : NULL;
    CHECK(bvhgpu_bcast(comm, trees, 0));
    CHECK(bvhgpu_bcast_known(comm, trees, 0, BVHGPU_F32, n, 0u));
    CHECK(bvhgpu_traverse_f32(trees[0], rays, 2, BVHGPU_HOST, 0u, &hits));
    uint64_t total2;
    CHECK(bvhgpu_hits_info(hits, &nr, &total2, NULL));
    printf("comm ranks %d first %d local %d; after bcast total %llu\n", nranks, first, nlocal, (unsigned long long)total2);
    {
        const size_t T = 60000;
        const float bounds[6] = {-3.f, -3.f, -3.f, 2.f * (float)m + 1.f, 2.f * (float)m + 1.f, 2.f * (float)m + 1.f};
        bvhgpu_hits* hs[64];
        unsigned long long sum = 0, count = 0, per_rank[64];
        void* ray_dev[64];   /* HBM through the ABI: this file has no HIP headers */
        for (int i = 0; i < K; i++) {
            size_t lo = (size_t)i * T / (size_t)K, hi = (size_t)(i + 1) * T / (size_t)K, cnt = hi - lo;
            CHECK(bvhgpu_device_alloc(ctxs[i], (cnt ? cnt : 1) * sizeof(bvhgpu_ray_f32), &ray_dev[i]));
            CHECK(bvhgpu_gen_rays_f32(ctxs[i], (uint64_t)lo, cnt, bounds, (bvhgpu_ray_f32*)ray_dev[i]));
            hs[i] = NULL;
        }
        for (int step = 0; step < 2; step++) {
            CHECK(bvhgpu_rebuild_flat_async_f32(trees[0], aabbs, n, BVHGPU_HOST));
            CHECK(bvhgpu_bcast_known(comm, trees, 0, BVHGPU_F32, n, 0u));
            for (int i = 0; i < K; i++) {
                size_t lo = (size_t)i * T / (size_t)K, hi = (size_t)(i + 1) * T / (size_t)K;
                CHECK(bvhgpu_traverse_async_f32(trees[i], (const bvhgpu_ray_f32*)ray_dev[i], hi - lo, BVHGPU_DEVICE, 0u, &hs[i]));
            }
            for (int i = 0; i < K; i++) CHECK(bvhgpu_hits_wait(hs[i]));
        }
        for (int i = 0; i < K; i++) {
            size_t lo = (size_t)i * T / (size_t)K, hi = (size_t)(i + 1) * T / (size_t)K, cnt = hi - lo;
            uint64_t tot; size_t nr2;
            CHECK(bvhgpu_hits_info(hs[i], &nr2, &tot, NULL));
            uint32_t* o2 = (uint32_t*)malloc((cnt + 1) * 4);
            uint32_t* i2 = (uint32_t*)malloc((tot ? tot : 1) * 4);
            CHECK(bvhgpu_hits_fetch(hs[i], o2, i2, NULL, BVHGPU_HOST));
            per_rank[i] = tot;
            for (size_t r = 0; r < cnt; r++)
                for (uint32_t j = o2[r]; j < o2[r + 1]; j++) { sum = sum * 1000003ull + (unsigned long long)(lo + r) * 31ull + i2[j]; count++; }
            free(o2); free(i2);
        }
        printf("shards %d rays %zu hits %llu checksum %llu per-rank", K, T, count, sum);
        for (int i = 0; i < K; i++) printf(" %llu", per_rank[i]);
        printf("\n");
        for (int i = 0; i < K; i++) { bvhgpu_hits_destroy(hs[i]); CHECK(bvhgpu_device_free(ctxs[i], ray_dev[i])); }
    }
    for (int i = 1; i < K; i++) bvhgpu_tree_destroy(trees[i]);
    bvhgpu_comm_destroy(comm);
    for (int i = 1; i < K; i++) bvhgpu_destroy(ctxs[i]);
    /* the process-per-GPU form with one rank */
    unsigned char id[BVHGPU_COMM_ID_BYTES];
    CHECK(bvhgpu_comm_unique_id(id));
    CHECK(bvhgpu_comm_init_rank(ctx, 1, 0, id, &comm));
    trees[0] = tree;
    CHECK(bvhgpu_bcast(comm, trees, 0));
    bvhgpu_comm_destroy(comm);
    bvhgpu_hits_destroy(hits);
    bvhgpu_tree_destroy(tree);
    bvhgpu_destroy(ctx);
    free(idx); free(aabbs);
    return 0;
}
