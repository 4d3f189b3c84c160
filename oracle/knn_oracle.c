/*
 * CPU ORACLE (test infrastructure, never shipped) -- exact brute-force k-NN.
 *
 * Restates what faiss IndexFlatL2 / IndexIVFFlat(nlist=1) compute for the reference at
 * manifold_gp/utils/nearest_neighbors.py:17-37: for each query the k database points with the
 * smallest squared Euclidean distance, ascending.  faiss itself is absent (un-pinned
 * third-party wheel), so the tie/rounding rule is DEFINED here and the HIP kernel must
 * reproduce it bit-for-bit:
 *
 *   d2(q, x) = sum_{j=0..d-1} ((double)q[j] - (double)x[j])^2      (fp64, j ascending,
 *              every operation individually rounded: compile with -ffp-contract=off)
 *   order    = ascending (d2, index)   -- the lower database index wins an exact tie
 *   D out    = (float)d2
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double d; int64_t i; } cand_t;

static inline int cand_less(double d1, int64_t i1, double d2, int64_t i2) {
    return (d1 < d2) || (d1 == d2 && i1 < i2);
}

/* max-heap on (d, i): root = current worst of the kept k */
static void sift_down(cand_t* h, int n, int p) {
    for (;;) {
        int l = 2 * p + 1, r = l + 1, m = p;
        if (l < n && cand_less(h[m].d, h[m].i, h[l].d, h[l].i)) m = l;
        if (r < n && cand_less(h[m].d, h[m].i, h[r].d, h[r].i)) m = r;
        if (m == p) return;
        cand_t t = h[p]; h[p] = h[m]; h[m] = t;
        p = m;
    }
}

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (cand_less(x->d, x->i, y->d, y->i)) return -1;
    if (cand_less(y->d, y->i, x->d, x->i)) return 1;
    return 0;
}

/* db [N,d] row-major f32, q [n,d] row-major f32 -> D [n,k] f32, I [n,k] i64.  returns 0. */
int oracle_knn_f64(const float* db, int64_t N, int d, const float* q, int64_t n, int k,
                   float* D, int64_t* I) {
    if (k <= 0 || k > N) return -1;
#pragma omp parallel
    {
        cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n; ++r) {
            const float* qr = q + r * (int64_t)d;
            int cnt = 0;
            for (int64_t c = 0; c < N; ++c) {
                const float* xr = db + c * (int64_t)d;
                double acc = 0.0;
                for (int j = 0; j < d; ++j) {
                    double df = (double)qr[j] - (double)xr[j];
                    double sq = df * df;
                    acc = acc + sq;
                }
                if (cnt < k) {
                    heap[cnt].d = acc; heap[cnt].i = c; ++cnt;
                    if (cnt == k) for (int p = k / 2 - 1; p >= 0; --p) sift_down(heap, k, p);
                } else if (cand_less(acc, c, heap[0].d, heap[0].i)) {
                    heap[0].d = acc; heap[0].i = c;
                    sift_down(heap, k, 0);
                }
            }
            qsort(heap, (size_t)k, sizeof(cand_t), cmp_cand);
            for (int t = 0; t < k; ++t) {
                D[r * (int64_t)k + t] = (float)heap[t].d;
                I[r * (int64_t)k + t] = heap[t].i;
            }
        }
        free(heap);
    }
    return 0;
}

/*
 * Reference-style CPU SpMV used ONLY as bench.py's cpu_baseline "port" leg when the torch
 * path is unavailable: y = diag*x - S x - S^T x over upper-triangular COO
 * (manifold_gp/operators/graph_laplacian_operator.py:117-119), single column, fp32.
 */
int oracle_coo_laplacian_mv(const int64_t* row, const int64_t* col, const float* s, int64_t M,
                            const float* diag, const float* x, int64_t N, float* y) {
    for (int64_t i = 0; i < N; ++i) y[i] = diag[i] * x[i];
    for (int64_t e = 0; e < M; ++e) {
        y[row[e]] -= s[e] * x[col[e]];
        y[col[e]] -= s[e] * x[row[e]];
    }
    return 0;
}
