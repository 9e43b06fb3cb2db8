/* caller.c -- a plain C (C99, no C++) host of include/metheor_hip.h: proves the header is C-clean and that the drop-in
 * boundary works without Python or C++ on the host side.  The batch is the SoA of the reference's tests/test1.bam
 * (test1_soa.h, written by tests/test_c_caller.py from the oracle's decode); expected: pdr.rs:226-237 with min_depth 0,
 * min_cpgs 0, min_qual 10 -> sites 0,2,4,6 each pdr 0.875, n_concordant 2, n_discordant 14; lpmd.rs:219 -> 0.5 (48/48).
 * Exit status 0 = all checks passed. */
#include <stdio.h>
#include <string.h>

#include "metheor_hip.h"
#include "test1_soa.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MTH_OK) { fprintf(stderr, "%s -> %d (%s) %s\n", #call, rc_, mth_strerror(rc_), ctx ? mth_last_error(ctx) : ""); return 1; } } while (0)

int main(void) {
    mth_ctx_t *ctx = NULL;
    mth_batch_t b;
    mth_pdr_lpmd_params_t p;
    uint64_t n = 0;
    int32_t tid[8], pos[8];
    float pdr[8], lpmd = -1.0f;
    uint32_t nc[8], nd[8];
    int64_t g[4];
    int i, ndev = 0;

    if (mth_abi_version() != MTH_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    CHECK(mth_device_count(&ndev));
    CHECK(mth_ctx_create(0, &ctx));
    memset(&b, 0, sizeof b);
    b.tid = 0; b.region_beg = 0; b.region_end = 248956422; b.max_span = 8;
    b.n_reads = T1_N_READS; b.n_cpgs = T1_N_CPGS; b.mem = MTH_MEM_HOST;
    b.read_start = t1_start; b.read_end = t1_end; b.read_mapq = t1_mapq; b.read_fwd = t1_fwd;
    b.cpg_off = t1_off; b.cpg_pos = t1_pos; b.cpg_rel = t1_rel;
    memset(&p, 0, sizeof p);
    p.pdr_min_depth = 0; p.pdr_min_cpgs = 0; p.pdr_min_qual = 10; p.lpmd_min_qual = 10;
    p.want_pdr = 1; p.want_lpmd = 1; p.lpmd_min_distance = 2; p.lpmd_max_distance = 16;
    CHECK(mth_pdr_lpmd_accumulate(ctx, &b, &p));
    CHECK(mth_pdr_count(ctx, &n));
    if (n != 4) { fprintf(stderr, "expected 4 sites, got %llu\n", (unsigned long long)n); return 1; }
    CHECK(mth_pdr_fetch(ctx, tid, pos, pdr, nc, nd));
    for (i = 0; i < 4; ++i) {
        if (tid[i] != 0 || pos[i] != 2 * i || pdr[i] != 0.875f || nc[i] != 2 || nd[i] != 14) {
            fprintf(stderr, "row %d: tid %d pos %d pdr %g nc %u nd %u\n", i, tid[i], pos[i], (double)pdr[i], nc[i], nd[i]);
            return 1;
        }
    }
    CHECK(mth_lpmd_global(ctx, g, &lpmd));
    if (g[0] != 48 || g[1] != 48 || g[2] != 16 || g[3] != 16 || lpmd != 0.5f) {
        fprintf(stderr, "lpmd: %lld %lld %lld %lld %g\n", (long long)g[0], (long long)g[1], (long long)g[2], (long long)g[3], (double)lpmd);
        return 1;
    }
    /* the single-context form of the exchange step is the identity */
    CHECK(mth_allreduce_lpmd(&ctx, 1));
    CHECK(mth_lpmd_global(ctx, g, &lpmd));
    if (g[0] != 48 || lpmd != 0.5f) return 1;
    mth_ctx_destroy(ctx);
    printf("c caller ok: 4 sites, lpmd 0.5, %d device(s)\n", ndev);
    return 0;
}
