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
    /* ---- all seven measures from ONE prepared batch (mth_batch_prepare: one device copy, one read index): the reference's
     * known answers on test1.bam -- pdr.rs:226-237, lpmd.rs:219, mhl.rs:257-259 (0.1625 x 4), me.rs:146-151 (one quartet, depth 16,
     * ME 1.0), pm.rs:145 (0.9375), fdrp.rs:261-267 (1.0 x 4 with -q 0 -d 2 -D 40 -l 4), qfdrp.rs:367-373 (8/15) ---- */
    {
        mth_batch_t pb;
        mth_quartet_params_t qp;
        mth_mhl_params_t mp;
        mth_fdrp_params_t fp;
        int32_t qtid[4], qpos[16], mtid[8], mpos[8], ftid[8], fpos[8];
        uint32_t qcnt[64], mcov[8], fn[8];
        float me[4], pm[4], mhl[8], fd[8], qf[8];
        const float q815 = 8.0f / 15.0f;
        CHECK(mth_batch_prepare(ctx, &b, &pb));
        if (pb.mem != MTH_MEM_PREPARED || pb.n_reads != T1_N_READS) { fprintf(stderr, "prepared batch: mem %d n_reads %u\n", pb.mem, pb.n_reads); return 1; }
        CHECK(mth_reset(ctx));
        CHECK(mth_pdr_lpmd_accumulate(ctx, &pb, &p));
        CHECK(mth_pdr_count(ctx, &n));
        if (n != 4) { fprintf(stderr, "prepared: expected 4 sites, got %llu\n", (unsigned long long)n); return 1; }
        CHECK(mth_pdr_fetch(ctx, tid, pos, pdr, nc, nd));
        for (i = 0; i < 4; ++i) if (pos[i] != 2 * i || pdr[i] != 0.875f || nc[i] != 2 || nd[i] != 14) { fprintf(stderr, "prepared pdr row %d\n", i); return 1; }
        CHECK(mth_lpmd_global(ctx, g, &lpmd));
        if (g[0] != 48 || g[1] != 48 || lpmd != 0.5f) { fprintf(stderr, "prepared lpmd\n"); return 1; }
        memset(&qp, 0, sizeof qp); qp.min_qual = 10;
        CHECK(mth_quartet_accumulate(ctx, &pb, &qp));
        n = 0;
        CHECK(mth_quartet_fetch(ctx, 10, &n, NULL, NULL, NULL, NULL, NULL));
        if (n != 1) { fprintf(stderr, "prepared: expected 1 quartet, got %llu\n", (unsigned long long)n); return 1; }
        CHECK(mth_quartet_fetch(ctx, 10, &n, qtid, qpos, qcnt, me, pm));
        if (qpos[0] != 0 || qpos[1] != 2 || qpos[2] != 4 || qpos[3] != 6 || me[0] != 1.0f || pm[0] != 0.9375f) { fprintf(stderr, "prepared me/pm: %g %g\n", (double)me[0], (double)pm[0]); return 1; }
        memset(&mp, 0, sizeof mp); mp.min_depth = 10; mp.min_cpgs = 4; mp.min_qual = 10;
        CHECK(mth_mhl_accumulate(ctx, &pb, &mp));
        n = 0;
        CHECK(mth_mhl_fetch(ctx, &n, NULL, NULL, NULL, NULL));
        if (n != 4) { fprintf(stderr, "prepared: expected 4 mhl rows, got %llu\n", (unsigned long long)n); return 1; }
        CHECK(mth_mhl_fetch(ctx, &n, mtid, mpos, mhl, mcov));
        for (i = 0; i < 4; ++i) if (mpos[i] != 2 * i || mhl[i] != 0.1625f) { fprintf(stderr, "prepared mhl row %d: %d %g\n", i, mpos[i], (double)mhl[i]); return 1; }
        memset(&fp, 0, sizeof fp); fp.min_depth = 2; fp.max_depth = 40; fp.min_overlap = 4; fp.min_qual = 0;
        CHECK(mth_fdrp_accumulate(ctx, &pb, &fp));
        n = 0;
        CHECK(mth_fdrp_fetch(ctx, &n, NULL, NULL, NULL, NULL, NULL));
        if (n != 4) { fprintf(stderr, "prepared: expected 4 fdrp rows, got %llu\n", (unsigned long long)n); return 1; }
        CHECK(mth_fdrp_fetch(ctx, &n, ftid, fpos, fd, qf, fn));
        for (i = 0; i < 4; ++i) if (fpos[i] != 2 * i || fd[i] != 1.0f || qf[i] != q815 || fn[i] != 16) { fprintf(stderr, "prepared fdrp row %d: %g %g %u\n", i, (double)fd[i], (double)qf[i], fn[i]); return 1; }
        CHECK(mth_batch_release(ctx, &pb));
        if (mth_pdr_lpmd_accumulate(ctx, &pb, &p) == MTH_OK) { fprintf(stderr, "a released prepared batch was accepted\n"); return 1; }
    }
    mth_ctx_destroy(ctx);
    printf("c caller ok: 4 sites, lpmd 0.5, seven measures from one prepared batch, %d device(s)\n", ndev);
    return 0;
}
