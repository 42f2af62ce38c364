// fold.hip — the table kernel and the host side of fold.h: several folds of partial parameter gradients per launch.
#include "fold.h"
#include "foldk.h"
#include <cstdlib>

thread_local FoldBatch* g_fold = nullptr;

bool fold_batch_enabled(size_t ntokens) {
    static const int knob = [] { const char* v = getenv("NBSS_FOLD_BATCH"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();
    return knob >= 0 ? knob != 0 : ntokens <= FOLD_MAX_TOKENS;
}

// One 256-thread block = one block of one of the table's folds (foldk.h); the item is found by walking the (at most six) block offsets.
__global__ __launch_bounds__(256) void fold_table_kernel(FoldTable t) {
    NBSS_LDS(smem);
    float* red = reinterpret_cast<float*>(smem);
    const int b = blockIdx.x;
    int k = 0;
    while (k + 1 < t.n && b >= t.blk0[k + 1]) ++k;
    const FoldItem& it = t.it[k];
    const int lb = b - t.blk0[k], bx = lb % it.gx, by = lb / it.gx;
    switch (it.kind) {
        case FK_WGRAD_REDUCE: fk_wgrad_reduce(it.u.wr.a, it.u.wr.xb, it.u.wr.nt_major, bx, by, it.gy, red); break;
        case FK_AFF_SLICES: fk_affine_slices(it.u.af.part, it.u.af.nwg, it.u.af.naff, bx * 256 + (int)threadIdx.x, by, it.gy); break;
        case FK_AFF_FINAL: fk_affine_final(it.u.af.part, it.u.af.nwg, it.u.af.naff, it.u.af.nsl, it.u.af.segs, it.u.af.G, bx * 256 + (int)threadIdx.x); break;
        case FK_TAILW_FIN:
            fk_tailw_finalize(it.u.tw.part, it.u.tw.xb, it.u.tw.MTA, it.u.tw.ntot, it.u.tw.W, it.u.tw.gamma, it.u.tw.beta, it.u.tw.dW, it.u.tw.dbias, bx, red);
            break;
        case FK_TAILW_AFF: fk_tailw_affine(it.u.tw.part, it.u.tw.MTA, it.u.tw.dgamma, it.u.tw.dbeta); break;
        case FK_P16_SLICES: fk_p16_slices((const bf16_t*)it.u.p16.part16, it.u.p16.nrows, it.u.p16.slices, it.u.p16.p16, bx, by, it.gy); break;
        case FK_TCONV_FINAL:
            fk_tconv_final(it.u.p16.slices, it.u.p16.nsl, it.u.p16.G, it.u.p16.off[0], it.u.p16.off[1], it.u.p16.off[2], it.u.p16.off[3], it.u.p16.p16, bx);
            break;
        case FK_FCONV_FINAL: fk_fconv_final(it.u.p16.slices, it.u.p16.nsl, it.u.p16.G, bx); break;
        case FK_FULL_SQ: fk_full_sq_final(it.u.sq.tmp, it.u.sq.Ws, it.u.sq.gamma, it.u.sq.beta, it.u.sq.dWs, it.u.sq.dbs, it.u.sq.dgamma, it.u.sq.dbeta); break;
        default: break;
    }
}

int FoldBatch::flush() {
    for (int s = 0; s < FOLD_STAGES; ++s) {
        FoldTable& t = tab[s];
        if (t.n > 0) {
            const int nblk = t.blk0[t.n];
            NBSS_LAUNCH(fold_table_kernel, dim3(nblk), dim3(256), 2 * FK_RSL * 64 * sizeof(float), st, t);
            const int e = NBSS_CHECK_LAUNCH();
            if (e) return e;
        }
    }
    reset();
    return NBSS_OK;
}

int FoldBatch::add(int stage, const FoldItem& it) {
    if (stage < 1 || stage > FOLD_STAGES || it.nblk <= 0) return NBSS_EINVAL;
    FoldTable& t = tab[stage - 1];
    if (t.n == FOLD_MAX_ITEMS) {  // (does not happen with the sub-blocks as they are: at most four first passes)
        const int e = flush();
        if (e) return e;
    }
    t.it[t.n] = it;
    t.blk0[t.n + 1] = t.blk0[t.n] + it.nblk;
    ++t.n;
    return NBSS_OK;
}

void* FoldBatch::alloc(size_t bytes, int* err) {
    *err = NBSS_OK;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes > pool_bytes) return nullptr;
    if (used + bytes > pool_bytes && (*err = flush())) return nullptr;
    void* p = pool + used;
    used += bytes;
    return p;
}

FoldScope::FoldScope(hipStream_t st, void* pool, size_t pool_bytes, size_t ntokens) : prev(g_fold), open(false) {
    // NBSS_FOLD_POOL=<bytes>: test knob — a pool smaller than the sub-block's partial tiles exercises the flush-when-full and the larger-than-the-pool paths
    static const long cap = [] { const char* v = getenv("NBSS_FOLD_POOL"); return v ? atol(v) : 0L; }();
    if (cap > 0 && (size_t)cap < pool_bytes) pool_bytes = (size_t)cap;
    fb.st = st;
    fb.pool = (char*)pool;
    fb.pool_bytes = pool_bytes;
    fb.reset();
    if (fold_batch_enabled(ntokens) && g_fold == nullptr) {
        g_fold = &fb;
        open = true;
    }
}
int FoldScope::end() {
    if (!open) return NBSS_OK;
    open = false;
    g_fold = prev;
    return fb.flush();
}
FoldScope::~FoldScope() {
    if (open) {
        g_fold = prev;
        fb.flush();
    }
}
