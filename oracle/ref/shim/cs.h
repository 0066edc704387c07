/* TEST INFRASTRUCTURE - stand-in for SuiteSparse's <cs.h> (CSparse is not installed in this image and not vendored by the reference: SURVEY.md §8c).
 * The interface (struct layouts, function names and contracts) is CSparse's, because the reference's g2o - solvers/linear_solver_csparse.h,
 * solvers/csparse_extension.cpp (which carries the NUMERIC Cholesky itself), core/marginal_covariance_cholesky.cpp - is compiled verbatim against it.
 * The bodies (oracle/ref/minics.cpp) are this repository's own restatement of the published algorithms (T. A. Davis, "Direct Methods for Sparse Linear
 * Systems", SIAM 2006): elimination tree, postorder, row-subtree reach, column counts by row-subtree traversal, symmetric permutation, triangular solves.
 * ONE deliberate difference: cs_amd returns a plain minimum-degree ordering (smallest index among equal degrees), not SuiteSparse's APPROXIMATE minimum
 * degree with its aggressive absorption - a fill-reducing ordering changes the order of the floating-point operations of the factorisation, not the
 * system that is solved.  Unpinned like the rest of the third-party arithmetic. */
#ifndef VDO_REF_MINI_CS_H_
#define VDO_REF_MINI_CS_H_
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct cs_sparse { int nzmax; int m; int n; int* p; int* i; double* x; int nz; } cs;      /* nz == -1: compressed columns */
typedef struct cs_symbolic { int* pinv; int* q; int* parent; int* cp; int* leftmost; int m2; double lnz; double unz; } css;
typedef struct cs_numeric { cs* L; cs* U; int* pinv; double* B; } csn;
#define CS_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define CS_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define CS_FLIP(i) (-(i)-2)
#define CS_UNFLIP(i) (((i) < 0) ? CS_FLIP(i) : (i))
#define CS_MARKED(w, j) (w[j] < 0)
#define CS_MARK(w, j) { w[j] = CS_FLIP(w[j]); }
#define CS_CSC(A) (A && (A->nz == -1))
#define CS_TRIPLET(A) (A && (A->nz >= 0))
void* cs_malloc(int n, size_t size);
void* cs_calloc(int n, size_t size);
void* cs_free(void* p);
cs* cs_spalloc(int m, int n, int nzmax, int values, int triplet);
cs* cs_spfree(cs* A);
csn* cs_nfree(csn* N);
css* cs_sfree(css* S);
csn* cs_ndone(csn* N, cs* C, void* w, void* x, int ok);
double cs_cumsum(int* p, int* c, int n);
int* cs_pinv(const int* p, int n);
int cs_pvec(const int* p, const double* b, double* x, int n);
int cs_ipvec(const int* p, const double* b, double* x, int n);
cs* cs_symperm(const cs* A, const int* pinv, int values);
int* cs_etree(const cs* A, int ata);
int* cs_post(const int* parent, int n);
int* cs_counts(const cs* A, const int* parent, const int* post, int ata);
int cs_ereach(const cs* A, int k, const int* parent, int* s, int* w);
int cs_lsolve(const cs* L, double* x);
int cs_ltsolve(const cs* L, double* x);
int* cs_amd(int order, const cs* A);
css* cs_schol(int order, const cs* A);
#ifdef __cplusplus
}
#endif
#endif
