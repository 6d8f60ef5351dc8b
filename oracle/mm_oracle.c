/*
 * mm_oracle.c -- builds the CPU oracle (see mm_oracle.inc for the contract).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the kaolin boundary (kaolin v0.12.0 is not in /root/reference).
 * Two instantiations: *_f32 (parity oracle / CPU baseline) and *_f64 (finite-difference self-checks).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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
