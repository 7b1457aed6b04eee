/* Plain C consumer of include/bvh_mi355x.h: build → flatten → traverse → fetch, printing everything a test needs
 * to compare with the CPU checker.  Proves the drop-in boundary is a C ABI (no Python, no C++ types).
 * usage: abi_roundtrip <n_cubes_per_axis>   — boxes of the reference's generate_aligned_boxes pattern, extended to 3-D */
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
    /* the multi-GPU exchange step with the one device this box has: a one-rank RCCL communicator (ncclCommInitAll),
     * the broadcast of the flattened tree (in place on the root), both forms; then the same two rays on the result */
    bvhgpu_comm* comm = NULL;
    bvhgpu_ctx* ctxs[1] = {ctx};
    CHECK(bvhgpu_comm_init_all(ctxs, 1, &comm));
    int nranks = -1, first = -1, nlocal = -1;
    CHECK(bvhgpu_comm_info(comm, &nranks, &first, &nlocal));
    bvhgpu_tree* trees[1] = {tree};
    CHECK(bvhgpu_bcast(comm, trees, 0));
    CHECK(bvhgpu_bcast_known(comm, trees, 0, BVHGPU_F32, n, 0u));
    CHECK(bvhgpu_traverse_f32(trees[0], rays, 2, BVHGPU_HOST, 0u, &hits));
    uint64_t total2;
    CHECK(bvhgpu_hits_info(hits, &nr, &total2, NULL));
    printf("comm ranks %d first %d local %d; after bcast total %llu\n", nranks, first, nlocal, (unsigned long long)total2);
    bvhgpu_comm_destroy(comm);
    /* the process-per-GPU form with one rank */
    unsigned char id[BVHGPU_COMM_ID_BYTES];
    CHECK(bvhgpu_comm_unique_id(id));
    CHECK(bvhgpu_comm_init_rank(ctx, 1, 0, id, &comm));
    CHECK(bvhgpu_bcast(comm, trees, 0));
    bvhgpu_comm_destroy(comm);
    bvhgpu_hits_destroy(hits);
    bvhgpu_tree_destroy(tree);
    bvhgpu_destroy(ctx);
    free(idx); free(aabbs);
    return 0;
}
