// TEST INFRASTRUCTURE - bodies of oracle/ref/shim/cs.h: the part of CSparse's interface that the reference's g2o calls, restated from the published
// algorithms (Davis 2006); see the header for what is and is not the same as SuiteSparse.  Not a product file.
#include "cs.h"
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

extern "C" {
void* cs_malloc(int n, size_t size) { return std::malloc((size_t)(n > 1 ? n : 1) * size); }
void* cs_calloc(int n, size_t size) { return std::calloc((size_t)(n > 1 ? n : 1), size); }
void* cs_free(void* p) { if (p) std::free(p); return nullptr; }

cs* cs_spalloc(int m, int n, int nzmax, int values, int triplet) {
  cs* A = (cs*)cs_calloc(1, sizeof(cs));
  if (!A) return nullptr;
  A->m = m; A->n = n; A->nzmax = nzmax = nzmax > 1 ? nzmax : 1; A->nz = triplet ? 0 : -1;
  A->p = (int*)cs_malloc(triplet ? nzmax : n + 1, sizeof(int));
  A->i = (int*)cs_malloc(nzmax, sizeof(int));
  A->x = values ? (double*)cs_malloc(nzmax, sizeof(double)) : nullptr;
  if (!A->p || !A->i || (values && !A->x)) return cs_spfree(A);
  return A;
}
cs* cs_spfree(cs* A) { if (!A) return nullptr; cs_free(A->p); cs_free(A->i); cs_free(A->x); cs_free(A); return nullptr; }
csn* cs_nfree(csn* N) { if (!N) return nullptr; cs_spfree(N->L); cs_spfree(N->U); cs_free(N->pinv); cs_free(N->B); cs_free(N); return nullptr; }
css* cs_sfree(css* S) { if (!S) return nullptr; cs_free(S->pinv); cs_free(S->q); cs_free(S->parent); cs_free(S->cp); cs_free(S->leftmost); cs_free(S); return nullptr; }
csn* cs_ndone(csn* N, cs* C, void* w, void* x, int ok) { cs_spfree(C); cs_free(w); cs_free(x); return ok ? N : cs_nfree(N); }

// p = running sum of c (p[n] = total), c becomes a copy of p[0..n-1]; the total is returned as a double
double cs_cumsum(int* p, int* c, int n) {
  if (!p || !c) return -1;
  int run = 0; double total = 0;
  for (int i = 0; i < n; ++i) { p[i] = run; run += c[i]; total += c[i]; c[i] = p[i]; }
  p[n] = run;
  return total;
}
int* cs_pinv(const int* p, int n) {
  if (!p) return nullptr;
  int* pinv = (int*)cs_malloc(n, sizeof(int));
  if (!pinv) return nullptr;
  for (int k = 0; k < n; ++k) pinv[p[k]] = k;
  return pinv;
}
int cs_pvec(const int* p, const double* b, double* x, int n) { if (!x || !b) return 0; for (int k = 0; k < n; ++k) x[k] = b[p ? p[k] : k]; return 1; }
int cs_ipvec(const int* p, const double* b, double* x, int n) { if (!x || !b) return 0; for (int k = 0; k < n; ++k) x[p ? p[k] : k] = b[k]; return 1; }

// C = P A P' for a symmetric A of which only the upper triangle is read; C holds its upper triangle
cs* cs_symperm(const cs* A, const int* pinv, int values) {
  if (!CS_CSC(A)) return nullptr;
  const int n = A->n; const int* Ap = A->p; const int* Ai = A->i; const double* Ax = A->x;
  cs* C = cs_spalloc(n, n, Ap[n], values && Ax != nullptr, 0);
  int* w = (int*)cs_calloc(n, sizeof(int));
  if (!C || !w) { cs_free(w); return cs_spfree(C); }
  for (int j = 0; j < n; ++j) {                 // entries per column of C
    const int j2 = pinv ? pinv[j] : j;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      if (i > j) continue;
      const int i2 = pinv ? pinv[i] : i;
      ++w[i2 > j2 ? i2 : j2];
    }
  }
  cs_cumsum(C->p, w, n);
  for (int j = 0; j < n; ++j) {
    const int j2 = pinv ? pinv[j] : j;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      if (i > j) continue;
      const int i2 = pinv ? pinv[i] : i;
      const int q = w[i2 > j2 ? i2 : j2]++;
      C->i[q] = i2 < j2 ? i2 : j2;
      if (C->x) C->x[q] = Ax[p];
    }
  }
  cs_free(w);
  return C;
}

// elimination tree of A (upper triangle) or of A'A
int* cs_etree(const cs* A, int ata) {
  if (!CS_CSC(A)) return nullptr;
  const int m = A->m, n = A->n; const int* Ap = A->p; const int* Ai = A->i;
  int* parent = (int*)cs_malloc(n, sizeof(int));
  int* w = (int*)cs_malloc(n + (ata ? m : 0), sizeof(int));
  if (!parent || !w) { cs_free(w); cs_free(parent); return nullptr; }
  int* ancestor = w; int* prev = w + n;
  if (ata) for (int i = 0; i < m; ++i) prev[i] = -1;
  for (int k = 0; k < n; ++k) {
    parent[k] = -1; ancestor[k] = -1;
    for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
      int i = ata ? prev[Ai[p]] : Ai[p];
      while (i != -1 && i < k) {                // walk to the root of i's current tree, compressing the path onto k
        const int next = ancestor[i];
        ancestor[i] = k;
        if (next == -1) parent[i] = k;
        i = next;
      }
      if (ata) prev[Ai[p]] = k;
    }
  }
  cs_free(w);
  return parent;
}

// postorder of a forest (children visited in increasing index order)
int* cs_post(const int* parent, int n) {
  if (!parent) return nullptr;
  int* post = (int*)cs_malloc(n, sizeof(int));
  int* w = (int*)cs_malloc(3 * n, sizeof(int));
  if (!post || !w) { cs_free(w); cs_free(post); return nullptr; }
  int* head = w; int* next = w + n; int* stack = w + 2 * n;
  for (int j = 0; j < n; ++j) head[j] = -1;
  for (int j = n - 1; j >= 0; --j) { if (parent[j] == -1) continue; next[j] = head[parent[j]]; head[parent[j]] = j; }
  int k = 0;
  for (int root = 0; root < n; ++root) {
    if (parent[root] != -1) continue;
    int top = 0; stack[0] = root;
    while (top >= 0) {
      const int p = stack[top]; const int child = head[p];
      if (child == -1) { --top; post[k++] = p; }
      else { head[p] = next[child]; stack[++top] = child; }
    }
  }
  cs_free(w);
  return post;
}

// column counts of the Cholesky factor of A (upper triangle given): every row subtree is walked once (cost |L|; CSparse uses the skeleton-matrix
// algorithm, whose result is the same numbers).  ata != 0 is not needed by the callers compiled here.
int* cs_counts(const cs* A, const int* parent, const int* post, int ata) {
  (void)post;
  if (!CS_CSC(A) || !parent || ata) return nullptr;
  const int n = A->n; const int* Ap = A->p; const int* Ai = A->i;
  int* count = (int*)cs_malloc(n, sizeof(int));
  int* mark = (int*)cs_malloc(n, sizeof(int));
  if (!count || !mark) { cs_free(mark); cs_free(count); return nullptr; }
  for (int k = 0; k < n; ++k) { count[k] = 1; mark[k] = -1; }
  for (int k = 0; k < n; ++k) {
    mark[k] = k;
    for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
      int i = Ai[p];
      if (i >= k) continue;
      for (; i != -1 && mark[i] != k; i = parent[i]) { ++count[i]; mark[i] = k; }      // L(k, i) != 0 for every i on the way up
    }
  }
  cs_free(mark);
  return count;
}

// pattern of row k of L: the nodes of the k-th row subtree, left in s[top .. n-1] in an order the caller's sparse triangular solve can use
int cs_ereach(const cs* A, int k, const int* parent, int* s, int* w) {
  if (!CS_CSC(A) || !parent || !s || !w) return -1;
  const int n = A->n; const int* Ap = A->p; const int* Ai = A->i;
  int top = n;
  CS_MARK(w, k);
  for (int p = Ap[k]; p < Ap[k + 1]; ++p) {
    int i = Ai[p];
    if (i > k) continue;
    int len = 0;
    while (!CS_MARKED(w, i)) { s[len++] = i; CS_MARK(w, i); i = parent[i]; }
    while (len > 0) s[--top] = s[--len];
  }
  for (int p = top; p < n; ++p) CS_MARK(w, s[p]);
  CS_MARK(w, k);
  return top;
}

int cs_lsolve(const cs* L, double* x) {
  if (!CS_CSC(L) || !x) return 0;
  const int n = L->n; const int* Lp = L->p; const int* Li = L->i; const double* Lx = L->x;
  for (int j = 0; j < n; ++j) {
    x[j] /= Lx[Lp[j]];
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) x[Li[p]] -= Lx[p] * x[j];
  }
  return 1;
}
int cs_ltsolve(const cs* L, double* x) {
  if (!CS_CSC(L) || !x) return 0;
  const int n = L->n; const int* Lp = L->p; const int* Li = L->i; const double* Lx = L->x;
  for (int j = n - 1; j >= 0; --j) {
    for (int p = Lp[j] + 1; p < Lp[j + 1]; ++p) x[j] -= Lx[p] * x[Li[p]];
    x[j] /= Lx[Lp[j]];
  }
  return 1;
}

// minimum-degree ordering of the graph of A + A' (order 1; order 0 = natural).  Elimination graph kept explicitly as sorted adjacency sets; the node of
// smallest current degree (smallest index among equals) is eliminated and its neighbours made a clique.  See cs.h: this is NOT SuiteSparse's AMD.
int* cs_amd(int order, const cs* A) {
  if (!CS_CSC(A) || order < 0 || order > 3) return nullptr;
  const int n = A->n; const int* Ap = A->p; const int* Ai = A->i;
  int* P = (int*)cs_malloc(n + 1, sizeof(int));
  if (!P) return nullptr;
  if (order == 0) { for (int k = 0; k < n; ++k) P[k] = k; P[n] = n; return P; }
  std::vector<std::set<int> > adj((size_t)n);
  for (int j = 0; j < n; ++j) for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i != j && i >= 0 && i < n) { adj[(size_t)i].insert(j); adj[(size_t)j].insert(i); } }
  std::set<std::pair<int, int> > queue;                       // (degree, node)
  for (int j = 0; j < n; ++j) queue.insert(std::make_pair((int)adj[(size_t)j].size(), j));
  for (int k = 0; k < n; ++k) {
    const int v = queue.begin()->second; queue.erase(queue.begin());
    P[k] = v;
    std::vector<int> nb(adj[(size_t)v].begin(), adj[(size_t)v].end());
    for (int u : nb) { queue.erase(std::make_pair((int)adj[(size_t)u].size(), u)); adj[(size_t)u].erase(v); }
    for (size_t a = 0; a < nb.size(); ++a) for (size_t b = a + 1; b < nb.size(); ++b) { adj[(size_t)nb[a]].insert(nb[b]); adj[(size_t)nb[b]].insert(nb[a]); }
    for (int u : nb) queue.insert(std::make_pair((int)adj[(size_t)u].size(), u));
    std::set<int>().swap(adj[(size_t)v]);
  }
  P[n] = n;
  return P;
}

// symbolic analysis for a Cholesky factorisation: ordering, elimination tree, column pointers of L
css* cs_schol(int order, const cs* A) {
  if (!CS_CSC(A)) return nullptr;
  const int n = A->n;
  css* S = (css*)cs_calloc(1, sizeof(css));
  if (!S) return nullptr;
  int* P = cs_amd(order, A);
  S->pinv = cs_pinv(P, n);
  cs_free(P);
  if (order && !S->pinv) return cs_sfree(S);
  cs* C = cs_symperm(A, S->pinv, 0);
  S->parent = cs_etree(C, 0);
  int* post = cs_post(S->parent, n);
  int* c = cs_counts(C, S->parent, post, 0);
  cs_free(post); cs_spfree(C);
  S->cp = (int*)cs_malloc(n + 1, sizeof(int));
  S->unz = S->lnz = cs_cumsum(S->cp, c, n);
  cs_free(c);
  return (S->lnz >= 0) ? S : cs_sfree(S);
}
}  // extern "C"
