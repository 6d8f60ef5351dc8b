/*
 * mm_oracle.c -- builds the CPU oracle (see mm_oracle.inc for the contract).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the kaolin boundary (kaolin v0.12.0 is not in /root/reference).
 * Two instantiations: *_f32 (parity oracle / CPU baseline) and *_f64 (finite-difference self-checks).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#define _GNU_SOURCE                                          /* MAP_ANONYMOUS under -std=c11 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Appendix-C switches (include/mm_render.h MM_OPT_*: the choices recalled from kaolin's sources that cannot be re-verified here),
 * mirrored bit for bit by the HIP path.  Process-wide on purpose: this is test infrastructure, set before a call, never concurrently. */
enum { MMO_OPT_CULL_STRICT = 1 << 4, MMO_OPT_SOFT_SKIP_CULLED = 1 << 5, MMO_OPT_BBOX_HALF_OPEN = 1 << 6, MMO_OPT_BARY_ONE_MINUS = 1 << 7,
       MMO_OPT_SH_ORDER_XYZ = 1 << 8, MMO_OPT_BBOX_MIN_CLOSED_MAX_OPEN = 1 << 9 };
/* scatter loops of the backward run as (image, band) tasks with private accumulators that are then added in band order (mm_oracle.inc) */
#define MMO_BANDS 8
#define MMO_TEX_BANDS 4
static int mmo_options = 0;

/* ---- scratch arena ------------------------------------------------------------------------------------------------------------------
 * The oracle materialises every intermediate like the reference does: ~0.5 GB of temporaries per step at B=48.  Taken from malloc / calloc
 * they are mmap'ed and unmapped on every call, and their first touch inside the OpenMP loops is 100 000 page faults under one address-space
 * lock: on the GPU box's 128 threads the step ran SLOWER than on 16 (thread scaling 2.2x).  Temporaries therefore come from a grow-only
 * arena that stays mapped between calls (reset when the outermost entry point is entered); zero-filling, where an accumulator needs it, is
 * a parallel loop.  Test infrastructure: one call at a time per process (allocations from inside parallel regions take a lock). */
#include <sys/mman.h>
#define MMO_MAX_CHUNKS 64
static struct { char* base; size_t cap, used; } mmo_chunks[MMO_MAX_CHUNKS];
static int mmo_nchunks = 0, mmo_depth = 0;
static void* mmo_alloc(size_t n) {
    void* out = NULL;
    n = (n + 255) & ~(size_t)255;
#pragma omp critical(mmo_arena)
    {
        for (int i = 0; i < mmo_nchunks && !out; ++i)
            if (mmo_chunks[i].cap - mmo_chunks[i].used >= n) { out = mmo_chunks[i].base + mmo_chunks[i].used; mmo_chunks[i].used += n; }
        if (!out && mmo_nchunks < MMO_MAX_CHUNKS) {
            size_t cap = n > ((size_t)256 << 20) ? n : ((size_t)256 << 20);
            void* m = mmap(NULL, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                mmo_chunks[mmo_nchunks].base = (char*)m; mmo_chunks[mmo_nchunks].cap = cap; mmo_chunks[mmo_nchunks].used = n;
                ++mmo_nchunks; out = m;
            }
        }
    }
    if (!out) abort();                                           /* (test infrastructure: out of address space is fatal) */
    return out;
}
static void* mmo_zalloc(size_t n) {
    char* p = (char*)mmo_alloc(n);
    const long long blocks = (long long)((n + ((size_t)1 << 20) - 1) >> 20);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < blocks; ++i) {
        const size_t o = (size_t)i << 20;
        memset(p + o, 0, n - o < ((size_t)1 << 20) ? n - o : ((size_t)1 << 20));
    }
    return p;
}
static void mmo_enter(void) { if (mmo_depth++ == 0) for (int i = 0; i < mmo_nchunks; ++i) mmo_chunks[i].used = 0; }
static void mmo_leave(void) { --mmo_depth; }
#define MMO_MALLOC(n) mmo_alloc(n)
#define MMO_CALLOC(n, sz) mmo_zalloc((size_t)(n) * (size_t)(sz))
#define MMO_FREE(p) ((void)(p))
/* which bbox borders are open (a centre exactly on them is outside): bit 0 the min border, bit 1 the max border -- include/mm_render.h */
static int mmo_box_mode(void) { return ((mmo_options & MMO_OPT_BBOX_HALF_OPEN) ? 3 : 0) | ((mmo_options & MMO_OPT_BBOX_MIN_CLOSED_MAX_OPEN) ? 2 : 0); }
static const uint8_t* mmo_soft_valid = NULL;   /* (B,F) faces the soft mask may use when MMO_OPT_SOFT_SKIP_CULLED is set (NULL: all) */
void mmo_set_options(int bits) { mmo_options = bits; }
int mmo_get_options(void) { return mmo_options; }
void mmo_set_soft_valid(const uint8_t* v) { mmo_soft_valid = v; }

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#define MMO_SQRT sqrtf
#define MMO_EXP expf
#define MMO_FLOOR floorf
#define MMO_FABS fabsf
#include "mm_oracle.inc"
#undef REAL
#undef FN
#undef MMO_SQRT
#undef MMO_EXP
#undef MMO_FLOOR
#undef MMO_FABS

#define REAL double
#define FN(name) CAT(name, _f64)
#define MMO_SQRT sqrt
#define MMO_EXP exp
#define MMO_FLOOR floor
#define MMO_FABS fabs
#include "mm_oracle.inc"

#ifdef _OPENMP
#include <omp.h>
int mmo_num_threads(void) { return omp_get_max_threads(); }
void mmo_set_threads(int n) { omp_set_num_threads(n); }
#else
int mmo_num_threads(void) { return 1; }
void mmo_set_threads(int n) { (void)n; }
#endif
