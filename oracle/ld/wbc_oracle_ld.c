/*
 * wbc_oracle_ld.c -- the CPU oracle (../wbc_oracle.c: test infrastructure, NOT product code) compiled in x87 extended precision
 * (long double: 64-bit mantissa).  Same source text, same algorithm, same literals; every `double` of the restatement becomes
 * `long double` and <tgmath.h> routes sin / cos / sqrt / atan2 / fabs ... to their long-double variants.  What it is for:
 * adjudicating disagreements between the HIP path and the double-precision oracle -- the QPs are strictly convex (eps2 > 0), so
 * the solution is unique and this build is ~2000x closer to it than either (tools/lab/truth.py, profiles/r03/truth.md).
 * The system headers are included BEFORE `double` is redefined, so their declarations stay what libm exports.
 */
#include <tgmath.h>
#undef I   /* <complex.h>'s imaginary unit: the restatement has struct fields of that name */
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#define double long double
#include "../wbc_oracle.h"
#include "../wbc_oracle.c"
