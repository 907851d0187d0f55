/*
 * isls_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, never shipped as product; see isls_oracle_impl.h).
 * Builds liboracle_isls.so:  make -C oracle
 * Entry points mirror include/isls_hip.h one-to-one (same argument structs, HOST pointers, no stream):
 *   oracle_<name>_f64 / _f32  <->  isls_<name>_f64 / _f32
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/isls_hip.h"
#include "isls_oracle.h"

#define REAL double
#define FN(name) name##_f64
#define SQRT sqrt
#define FABS fabs
#define FMOD fmod
#define SIN sin
#define COS cos
#define ASIN asin
#include "isls_oracle_impl.h"
#undef REAL
#undef FN
#undef SQRT
#undef FABS
#undef FMOD
#undef SIN
#undef COS
#undef ASIN

#define REAL float
#define FN(name) name##_f32
#define SQRT sqrtf
#define FABS fabsf
#define FMOD fmodf
#define SIN sinf
#define COS cosf
#define ASIN asinf
#include "isls_oracle_impl.h"

int oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
