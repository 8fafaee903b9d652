/*
 * pdwt_oracle.c -- CPU restatement of the reference's wavelet hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (pypwt_amd/) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the reported
 * CPU baseline.  The product fails loudly when its HIP library is missing.
 *
 * What is restated (all citations relative to /root/reference):
 *   sizes / levels       pdwt/src/utils.cu:14-27, pdwt/src/wt.cu:155-165
 *   DWT analysis         pdwt/src/separable.cu:91-131 (rows), :135-176 (columns)
 *   DWT synthesis        pdwt/src/separable.cu:246-289 (columns), :293-328 (rows)
 *   SWT analysis         pdwt/src/separable.cu:409-448, :452-493
 *   SWT synthesis        pdwt/src/separable.cu:553-589, :593-626
 *   level loops          pdwt/src/separable.cu:179-236, :332-395, :496-537, :629-672
 *   Haar butterflies     pdwt/src/haar.cu:10-58 (2D), :132-160 (1D)
 *   coefficient layout   pdwt/src/common.cu:399-445, pdwt/src/wt.cu:473-506
 *   thresholds / shrink  pdwt/src/common.cu:13-137, :219-371
 *   norms                pdwt/src/wt.cu:368-416
 *   non-separable 2D     pdwt/src/nonseparable.cu:114-225, :304-401
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here
 * against the pywt vectors in tests/golden/ (pywt is the reference's own
 * oracle, test/test_wavelets.py:230,301,372,438).
 *
 * The arithmetic type is REAL: float (default; what the reference computes in,
 * DTYPE at pdwt/src/filters.h:16-30) or double (-DORACLE_DOUBLE; used to pin
 * the index math against pywt's float64 without fp32 noise).  Data in and out
 * is DATA: float32, or float64 with -DORACLE_STORE_DOUBLE (the checker of the
 * product's fp64 build, libpypwt_amd_f64.so).
 *
 * The code is written as plain loops over output samples; it is not a copy of
 * the CUDA kernels: boundary handling is one true-modulo function, synthesis is
 * one polyphase formula, and level loops write each band once.
 */
#include <math.h>
#include <tgmath.h> /* fabs, fmax, copysign, sqrt resolve to the f-suffixed functions for float DATA */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_DOUBLE
typedef double REAL;
#else
typedef float REAL;
#endif
#ifdef ORACLE_STORE_DOUBLE
typedef double DATA;
#else
typedef float DATA;
#endif

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ sizes */

/* ceil-half: an odd length is first extended by one sample (utils.cu:24-27) */
API int oracle_div2(int n) { return (n + (n & 1)) / 2; }

/* floor(log2(i)) for i >= 1, 0 otherwise (utils.cu:14-20) */
API int oracle_ilog2(int i) {
    int l = 0;
    while (i > 1) { i >>= 1; ++l; }
    return l;
}

/* level clamp of the constructor (wt.cu:155-165): floor(log2(N/(hlen-1))) */
API int oracle_max_level(int N, int hlen) {
    if (hlen < 2) return 0;
    return oracle_ilog2(N / (hlen - 1));
}

API int oracle_real_is_double(void) { return sizeof(REAL) == 8; }
API int oracle_data_is_double(void) { return sizeof(DATA) == 8; }

/* Periodized source index for the analysis passes (separable.cu:114-121):
 * the signal of length N is extended to Np = N + (N odd) by repeating the
 * last sample, and that extended signal is periodic. */
static inline int per_src(int i, int N) {
    int Np = N + (N & 1);
    int m = i % Np;
    if (m < 0) m += Np;
    if (m >= N) m = N - 1; /* only reachable when N is odd: the virtual sample */
    return m;
}

static inline int mod_n(int i, int N) {
    int m = i % N;
    return m < 0 ? m + N : m;
}

/* analysis centre (separable.cu:98-107): even filters are shifted left */
static inline int ana_centre(int hlen) { return (hlen & 1) ? hlen / 2 : hlen / 2 - 1; }

/* ------------------------------------------------- decimated analysis (DWT) */

/* rows: in (Nr,Nc) -> outL,outH (Nr, ceil(Nc/2))   [separable.cu:91-131]
 * out[k] = sum_j x[per(2k - c + j)] * f[hlen-1-j]                          */
API void oracle_analysis_rows(const DATA *in, int Nr, int Nc, const DATA *lo, const DATA *hi,
                              int hlen, DATA *outL, DATA *outH) {
    const int Nc2 = oracle_div2(Nc), c = ana_centre(hlen);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++) {
        const DATA *row = in + (size_t)y * Nc;
        for (int k = 0; k < Nc2; k++) {
            REAL aL = 0, aH = 0;
            for (int j = 0; j < hlen; j++) {
                REAL v = row[per_src(2 * k - c + j, Nc)];
                aL += v * (REAL)lo[hlen - 1 - j];
                aH += v * (REAL)hi[hlen - 1 - j];
            }
            outL[(size_t)y * Nc2 + k] = (DATA)aL;
            outH[(size_t)y * Nc2 + k] = (DATA)aH;
        }
    }
}

/* columns: in (Nr,Nc) -> outL,outH (ceil(Nr/2), Nc)   [separable.cu:135-176] */
API void oracle_analysis_cols(const DATA *in, int Nr, int Nc, const DATA *lo, const DATA *hi,
                              int hlen, DATA *outL, DATA *outH) {
    const int Nr2 = oracle_div2(Nr), c = ana_centre(hlen);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < Nr2; k++) {
        for (int x = 0; x < Nc; x++) {
            REAL aL = 0, aH = 0;
            for (int j = 0; j < hlen; j++) {
                REAL v = in[(size_t)per_src(2 * k - c + j, Nr) * Nc + x];
                aL += v * (REAL)lo[hlen - 1 - j];
                aH += v * (REAL)hi[hlen - 1 - j];
            }
            outL[(size_t)k * Nc + x] = (DATA)aL;
            outH[(size_t)k * Nc + x] = (DATA)aH;
        }
    }
}

/* ------------------------------------------------ decimated synthesis (IDWT)
 * One polyphase formula for both parities of hlen/2 (separable.cu:250-287 and
 * :297-326).  With h2 = hlen/2, c = h2/2 and p = g + (h2 even ? 1 : 0):
 *   out[g] = sum_{j=0}^{h2-1}  a[(p/2 - c + j) mod Nin] * rlo[t_j]
 *                            + d[(p/2 - c + j) mod Nin] * rhi[t_j],
 *   t_j = hlen - 1 - (2j + 1 - (p & 1)).
 * Nin = number of coefficients, Nout <= 2*Nin = number of samples produced
 * (an odd original length drops the virtual last sample, separable.cu:296).  */
static inline void syn_params(int hlen, int *h2, int *c, int *shift) {
    *h2 = hlen / 2;
    *c = *h2 / 2;
    *shift = ((*h2) & 1) ? 0 : 1;
}

API void oracle_synthesis_cols(const DATA *a, const DATA *d, int Nin, int Nc, int Nout,
                               const DATA *rlo, const DATA *rhi, int hlen, DATA *out) {
    int h2, c, s;
    syn_params(hlen, &h2, &c, &s);
#pragma omp parallel for schedule(static)
    for (int g = 0; g < Nout; g++) {
        const int p = g + s, base = p / 2 - c, par = 1 - (p & 1);
        for (int x = 0; x < Nc; x++) {
            REAL ra = 0, rd = 0;
            for (int j = 0; j < h2; j++) {
                const int t = hlen - 1 - (2 * j + par);
                if (t < 0) continue; /* odd custom filter lengths */
                const size_t src = (size_t)mod_n(base + j, Nin) * Nc + x;
                ra += (REAL)a[src] * (REAL)rlo[t];
                rd += (REAL)d[src] * (REAL)rhi[t];
            }
            out[(size_t)g * Nc + x] = (DATA)(ra + rd);
        }
    }
}

API void oracle_synthesis_rows(const DATA *a, const DATA *d, int Nr, int Nin, int Nout,
                               const DATA *rlo, const DATA *rhi, int hlen, DATA *out) {
    int h2, c, s;
    syn_params(hlen, &h2, &c, &s);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++) {
        const DATA *ar = a + (size_t)y * Nin, *dr = d + (size_t)y * Nin;
        for (int g = 0; g < Nout; g++) {
            const int p = g + s, base = p / 2 - c, par = 1 - (p & 1);
            REAL ra = 0, rd = 0;
            for (int j = 0; j < h2; j++) {
                const int t = hlen - 1 - (2 * j + par);
                if (t < 0) continue;
                const int src = mod_n(base + j, Nin);
                ra += (REAL)ar[src] * (REAL)rlo[t];
                rd += (REAL)dr[src] * (REAL)rhi[t];
            }
            out[(size_t)y * Nout + g] = (DATA)(ra + rd);
        }
    }
}

/* ------------------------------------------------- undecimated analysis (SWT)
 * out[g] = sum_j x[(g + (j - c) f) mod N] * filt[hlen-1-j],  f = 2^(level-1)
 * (separable.cu:409-448 rows, :452-493 columns)                              */
API void oracle_swt_analysis_rows(const DATA *in, int Nr, int Nc, const DATA *lo, const DATA *hi,
                                  int hlen, int level, DATA *outL, DATA *outH) {
    const int f = 1 << (level - 1), c = ana_centre(hlen);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++) {
        const DATA *row = in + (size_t)y * Nc;
        for (int g = 0; g < Nc; g++) {
            REAL aL = 0, aH = 0;
            for (int j = 0; j < hlen; j++) {
                REAL v = row[mod_n(g + (j - c) * f, Nc)];
                aL += v * (REAL)lo[hlen - 1 - j];
                aH += v * (REAL)hi[hlen - 1 - j];
            }
            outL[(size_t)y * Nc + g] = (DATA)aL;
            outH[(size_t)y * Nc + g] = (DATA)aH;
        }
    }
}

API void oracle_swt_analysis_cols(const DATA *in, int Nr, int Nc, const DATA *lo, const DATA *hi,
                                  int hlen, int level, DATA *outL, DATA *outH) {
    const int f = 1 << (level - 1), c = ana_centre(hlen);
#pragma omp parallel for schedule(static)
    for (int g = 0; g < Nr; g++) {
        for (int x = 0; x < Nc; x++) {
            REAL aL = 0, aH = 0;
            for (int j = 0; j < hlen; j++) {
                REAL v = in[(size_t)mod_n(g + (j - c) * f, Nr) * Nc + x];
                aL += v * (REAL)lo[hlen - 1 - j];
                aH += v * (REAL)hi[hlen - 1 - j];
            }
            outL[(size_t)g * Nc + x] = (DATA)aL;
            outH[(size_t)g * Nc + x] = (DATA)aH;
        }
    }
}

/* ------------------------------------------------ undecimated synthesis (ISWT)
 * out[g] = 1/2 sum_j ( a[(g - c f + j f) mod N] rlo[hlen-1-j]
 *                    + d[(g - c f + j f) mod N] rhi[hlen-1-j] ),  c = hlen/2
 * (separable.cu:553-589 columns, :593-626 rows; the 1/2 sits inside the MAC
 *  at :581-584 and :621-622)                                                  */
API void oracle_swt_synthesis_cols(const DATA *a, const DATA *d, int Nr, int Nc, const DATA *rlo,
                                   const DATA *rhi, int hlen, int level, DATA *out) {
    const int f = 1 << (level - 1), c = hlen / 2;
#pragma omp parallel for schedule(static)
    for (int g = 0; g < Nr; g++) {
        for (int x = 0; x < Nc; x++) {
            REAL ra = 0, rd = 0;
            for (int j = 0; j < hlen; j++) {
                const size_t src = (size_t)mod_n(g - c * f + j * f, Nr) * Nc + x;
                ra += (REAL)a[src] * (REAL)rlo[hlen - 1 - j] / 2;
                rd += (REAL)d[src] * (REAL)rhi[hlen - 1 - j] / 2;
            }
            out[(size_t)g * Nc + x] = (DATA)(ra + rd);
        }
    }
}

API void oracle_swt_synthesis_rows(const DATA *a, const DATA *d, int Nr, int Nc, const DATA *rlo,
                                   const DATA *rhi, int hlen, int level, DATA *out) {
    const int f = 1 << (level - 1), c = hlen / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++) {
        const DATA *ar = a + (size_t)y * Nc, *dr = d + (size_t)y * Nc;
        for (int g = 0; g < Nc; g++) {
            REAL ra = 0, rd = 0;
            for (int j = 0; j < hlen; j++) {
                const int src = mod_n(g - c * f + j * f, Nc);
                ra += (REAL)ar[src] * (REAL)rlo[hlen - 1 - j] / 2;
                rd += (REAL)dr[src] * (REAL)rhi[hlen - 1 - j] / 2;
            }
            out[(size_t)y * Nc + g] = (DATA)(ra + rd);
        }
    }
}

/* ------------------------------------------------------- Haar (haar.cu:10-160)
 * Selected by the reference when hlen == 2 and the transform is decimated
 * (wt.cu:248,255,282,289). */
API void oracle_haar2d_fwd(const DATA *img, int Nr, int Nc, DATA *A, DATA *H, DATA *V, DATA *D) {
    const int Nr2 = oracle_div2(Nr), Nc2 = oracle_div2(Nc);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr2; y++) {
        int y0 = 2 * y, y1 = 2 * y + 1;
        if (y1 == Nr) y1--; /* odd size: repeat the last row (haar.cu:19-25) */
        for (int x = 0; x < Nc2; x++) {
            int x0 = 2 * x, x1 = 2 * x + 1;
            if (x1 == Nc) x1--;
            REAL a = img[(size_t)y0 * Nc + x0], b = img[(size_t)y0 * Nc + x1];
            REAL c = img[(size_t)y1 * Nc + x0], d = img[(size_t)y1 * Nc + x1];
            const size_t o = (size_t)y * Nc2 + x;
            A[o] = (DATA)((REAL)0.5 * ((a + c) + (b + d)));
            V[o] = (DATA)((REAL)0.5 * ((a + c) - (b + d)));
            H[o] = (DATA)((REAL)0.5 * ((a - c) + (b - d)));
            D[o] = (DATA)((REAL)0.5 * ((a - c) - (b - d)));
        }
    }
}

/* coefficients (Nrc,Ncc) -> image (Nr,Nc), Nr <= 2 Nrc  (haar.cu:41-58) */
API void oracle_haar2d_inv(DATA *img, const DATA *A, const DATA *H, const DATA *V, const DATA *D,
                           int Nrc, int Ncc, int Nr, int Nc) {
    (void)Nrc;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++) {
        for (int x = 0; x < Nc; x++) {
            const size_t s = (size_t)(y / 2) * Ncc + (x / 2);
            REAL a = A[s], b = V[s], c = H[s], d = D[s], r;
            const int gx = x & 1, gy = y & 1;
            if (!gx && !gy) r = (REAL)0.5 * ((a + c) + (b + d));
            else if (gx && !gy) r = (REAL)0.5 * ((a + c) - (b + d));
            else if (!gx && gy) r = (REAL)0.5 * ((a - c) + (b - d));
            else r = (REAL)0.5 * ((a - c) - (b - d));
            img[(size_t)y * Nc + x] = (DATA)r;
        }
    }
}

#define ORACLE_ONE_SQRT2 0.70710678118654746

API void oracle_haar1d_fwd(const DATA *img, int Nr, int Nc, DATA *A, DATA *D) {
    const int Nc2 = oracle_div2(Nc);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++)
        for (int x = 0; x < Nc2; x++) {
            int x1 = 2 * x + 1;
            if (x1 == Nc) x1--;
            REAL a = img[(size_t)y * Nc + 2 * x], b = img[(size_t)y * Nc + x1];
            A[(size_t)y * Nc2 + x] = (DATA)((REAL)ORACLE_ONE_SQRT2 * (a + b));
            D[(size_t)y * Nc2 + x] = (DATA)((REAL)ORACLE_ONE_SQRT2 * (a - b));
        }
}

API void oracle_haar1d_inv(DATA *img, const DATA *A, const DATA *D, int Nr, int Ncc, int Nc) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr; y++)
        for (int x = 0; x < Nc; x++) {
            REAL a = A[(size_t)y * Ncc + x / 2], b = D[(size_t)y * Ncc + x / 2];
            img[(size_t)y * Nc + x] = (DATA)((REAL)ORACLE_ONE_SQRT2 * ((x & 1) ? (a - b) : (a + b)));
        }
}

/* ------------------------------------------------------ coefficient layout
 * Flat buffer in the reference's band order (common.cu:399-445, wt.cu:473-506):
 *   2D: [A_L | H1 V1 D1 | H2 V2 D2 | ... ]   level 1 = finest
 *   1D: [A_L | D1 | D2 | ... ]  every band (Nr, Nc_l); batched rows independent
 * SWT: every band is full size.                                              */
API void oracle_level_shape(int Nr, int Nc, int ndim, int do_swt, int level, int *r, int *c) {
    for (int i = 0; i < level && !do_swt; i++) {
        if (ndim == 2) Nr = oracle_div2(Nr);
        Nc = oracle_div2(Nc);
    }
    *r = Nr;
    *c = Nc;
}

API int oracle_num_bands(int ndim, int levels) { return ndim == 2 ? 3 * levels + 1 : levels + 1; }

/* element offset of band `num` in the flat buffer, and its shape */
API size_t oracle_band_offset(int Nr, int Nc, int ndim, int do_swt, int levels, int num, int *r, int *c) {
    size_t off = 0;
    int br, bc;
    oracle_level_shape(Nr, Nc, ndim, do_swt, levels, &br, &bc);
    if (num == 0) { *r = br; *c = bc; return 0; }
    off += (size_t)br * bc;
    const int per = ndim == 2 ? 3 : 1;
    for (int n = 1; n <= num; n++) {
        const int lvl = (n - 1) / per + 1;
        oracle_level_shape(Nr, Nc, ndim, do_swt, lvl, &br, &bc);
        if (n == num) { *r = br; *c = bc; return off; }
        off += (size_t)br * bc;
    }
    return off;
}

API size_t oracle_coeff_count(int Nr, int Nc, int ndim, int do_swt, int levels) {
    int r, c;
    const int nb = oracle_num_bands(ndim, levels);
    size_t off = oracle_band_offset(Nr, Nc, ndim, do_swt, levels, nb - 1, &r, &c);
    return off + (size_t)r * c;
}

/* ------------------------------------------------------------- level loops */

/* filters = { dec_lo, dec_hi, rec_lo, rec_hi }.  Returns 0, or <0 on bad args. */
API int oracle_forward(const DATA *img, int Nr, int Nc, int ndim, int do_swt, int levels,
                       const DATA *dec_lo, const DATA *dec_hi, int hlen, DATA *coeffs) {
    if (levels < 1 || hlen < 1 || Nr < 1 || Nc < 1 || (ndim != 1 && ndim != 2)) return -1;
    const size_t n = (size_t)Nr * Nc;
    DATA *cur = (DATA *)malloc(n * sizeof(DATA));
    DATA *nxt = (DATA *)malloc(n * sizeof(DATA));
    DATA *t1 = (DATA *)malloc(n * sizeof(DATA));
    DATA *t2 = (DATA *)malloc(n * sizeof(DATA));
    if (!cur || !nxt || !t1 || !t2) { free(cur); free(nxt); free(t1); free(t2); return -2; }
    memcpy(cur, img, n * sizeof(DATA));
    int r = Nr, c = Nc;
    const int per = ndim == 2 ? 3 : 1;
    const int haar = (hlen == 2 && !do_swt); /* wt.cu:248,255 */
    for (int l = 1; l <= levels; l++) {
        int br, bc, dummy_r, dummy_c;
        const size_t off = oracle_band_offset(Nr, Nc, ndim, do_swt, levels, per * (l - 1) + 1, &br, &bc);
        (void)dummy_r; (void)dummy_c;
        DATA *b0 = coeffs + off, *b1 = b0 + (size_t)br * bc, *b2 = b1 + (size_t)br * bc;
        if (ndim == 2) {
            if (haar) {
                oracle_haar2d_fwd(cur, r, c, nxt, b0, b1, b2);
            } else if (!do_swt) {
                /* separable.cu:196-207: rows first, then columns;
                 * A = L(tmpL), H = H(tmpL), V = L(tmpH), D = H(tmpH) (:165-174) */
                oracle_analysis_rows(cur, r, c, dec_lo, dec_hi, hlen, t1, t2);
                oracle_analysis_cols(t1, r, bc, dec_lo, dec_hi, hlen, nxt, b0);
                oracle_analysis_cols(t2, r, bc, dec_lo, dec_hi, hlen, b1, b2);
            } else {
                oracle_swt_analysis_rows(cur, r, c, dec_lo, dec_hi, hlen, l, t1, t2);
                oracle_swt_analysis_cols(t1, r, c, dec_lo, dec_hi, hlen, l, nxt, b0);
                oracle_swt_analysis_cols(t2, r, c, dec_lo, dec_hi, hlen, l, b1, b2);
            }
        } else {
            if (haar) oracle_haar1d_fwd(cur, r, c, nxt, b0);
            else if (!do_swt) oracle_analysis_rows(cur, r, c, dec_lo, dec_hi, hlen, nxt, b0);
            else oracle_swt_analysis_rows(cur, r, c, dec_lo, dec_hi, hlen, l, nxt, b0);
        }
        r = br; c = bc;
        DATA *sw = cur; cur = nxt; nxt = sw;
    }
    memcpy(coeffs, cur, (size_t)r * c * sizeof(DATA));
    free(cur); free(nxt); free(t1); free(t2);
    return 0;
}

API int oracle_inverse(const DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels,
                       const DATA *rec_lo, const DATA *rec_hi, int hlen, DATA *img) {
    if (levels < 1 || hlen < 1 || Nr < 1 || Nc < 1 || (ndim != 1 && ndim != 2)) return -1;
    const size_t n = (size_t)Nr * Nc;
    DATA *cur = (DATA *)malloc(n * sizeof(DATA));
    DATA *nxt = (DATA *)malloc(n * sizeof(DATA));
    DATA *t1 = (DATA *)malloc(n * sizeof(DATA));
    DATA *t2 = (DATA *)malloc(n * sizeof(DATA));
    if (!cur || !nxt || !t1 || !t2) { free(cur); free(nxt); free(t1); free(t2); return -2; }
    int r, c;
    oracle_level_shape(Nr, Nc, ndim, do_swt, levels, &r, &c);
    memcpy(cur, coeffs, (size_t)r * c * sizeof(DATA));
    const int per = ndim == 2 ? 3 : 1;
    const int haar = (hlen == 2 && !do_swt);
    for (int l = levels; l >= 1; l--) {
        int br, bc, orr, oc;
        const size_t off = oracle_band_offset(Nr, Nc, ndim, do_swt, levels, per * (l - 1) + 1, &br, &bc);
        oracle_level_shape(Nr, Nc, ndim, do_swt, l - 1, &orr, &oc);
        const DATA *b0 = coeffs + off, *b1 = b0 + (size_t)br * bc, *b2 = b1 + (size_t)br * bc;
        if (ndim == 2) {
            if (haar) {
                oracle_haar2d_inv(nxt, cur, b0, b1, b2, br, bc, orr, oc);
            } else if (!do_swt) {
                /* separable.cu:351-361: columns (A,H)->tmp1, (V,D)->tmp2, then rows */
                oracle_synthesis_cols(cur, b0, br, bc, orr, rec_lo, rec_hi, hlen, t1);
                oracle_synthesis_cols(b1, b2, br, bc, orr, rec_lo, rec_hi, hlen, t2);
                oracle_synthesis_rows(t1, t2, orr, bc, oc, rec_lo, rec_hi, hlen, nxt);
            } else {
                oracle_swt_synthesis_cols(cur, b0, br, bc, rec_lo, rec_hi, hlen, l, t1);
                oracle_swt_synthesis_cols(b1, b2, br, bc, rec_lo, rec_hi, hlen, l, t2);
                oracle_swt_synthesis_rows(t1, t2, br, bc, rec_lo, rec_hi, hlen, l, nxt);
            }
        } else {
            if (haar) oracle_haar1d_inv(nxt, cur, b0, br, bc, oc);
            else if (!do_swt) oracle_synthesis_rows(cur, b0, br, bc, oc, rec_lo, rec_hi, hlen, nxt);
            else oracle_swt_synthesis_rows(cur, b0, br, bc, rec_lo, rec_hi, hlen, l, nxt);
        }
        DATA *sw = cur; cur = nxt; nxt = sw;
    }
    memcpy(img, cur, n * sizeof(DATA));
    free(cur); free(nxt); free(t1); free(t2);
    return 0;
}

/* --------------------------------------------------- coefficient operators */

static inline DATA soft1(DATA x, DATA b) { return copysign(fmax(fabs(x) - b, (DATA)0), x); }
static inline DATA hard1(DATA x, DATA b) { return (fabs(x) - b > (DATA)0) ? x : (DATA)0; }
static inline DATA linf1(DATA x, DATA b) { return copysign(fmin(fabs(x), b), x); }

/* beta for the approximation band when `normalize` (common.cu:229-236):
 * beta / sqrt(2)^levels computed as a shift plus one optional 1/sqrt(2) */
static DATA app_beta(DATA beta, int levels, int normalize) {
    if (normalize > 0) {
        const int n2 = levels / 2;
        beta /= (DATA)(1 << n2);
        if (n2 * 2 != levels) beta = (DATA)(beta / 1.4142135623730951); /* SQRT_2 is a double, common.cu:8 */
    }
    return beta;
}

/* op: 0 soft (common.cu:13-52,219-249), 1 hard (:57-97,252-282; the reference
 * passes beta instead of beta2 to the approximation band at :270 -- restated
 * here WITH the normalised beta2, the documented intent), 2 proj_linf
 * (:101-137,285-308; no normalize) */
API void oracle_threshold(DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels, int op,
                          DATA beta, int do_app, int normalize) {
    int r, c;
    if (do_app) {
        oracle_band_offset(Nr, Nc, ndim, do_swt, levels, 0, &r, &c);
        const DATA b2 = (op == 2) ? beta : app_beta(beta, levels, normalize);
        for (size_t i = 0; i < (size_t)r * c; i++)
            coeffs[i] = op == 0 ? soft1(coeffs[i], b2) : op == 1 ? hard1(coeffs[i], b2) : linf1(coeffs[i], b2);
    }
    const int per = ndim == 2 ? 3 : 1;
    for (int l = 1; l <= levels; l++) {
        if (normalize > 0 && op != 2) beta = (DATA)(beta / 1.4142135623730951); /* common.cu:244 */
        for (int k = 0; k < per; k++) {
            const size_t off = oracle_band_offset(Nr, Nc, ndim, do_swt, levels, per * (l - 1) + 1 + k, &r, &c);
            DATA *b = coeffs + off;
            for (size_t i = 0; i < (size_t)r * c; i++)
                b[i] = op == 0 ? soft1(b[i], beta) : op == 1 ? hard1(b[i], beta) : linf1(b[i], beta);
        }
    }
}

/* group soft threshold (common.cu:145-198, 311-341): per pixel, the detail
 * bands of one level (plus A at the last level when do_app) shrink by
 * max(1 - beta/||.||_2, 0) */
API void oracle_group_soft_threshold(DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels,
                                     DATA beta, int do_app, int normalize) {
    const int per = ndim == 2 ? 3 : 1;
    int r, c;
    for (int l = 1; l <= levels; l++) {
        if (normalize > 0) beta = (DATA)(beta / 1.4142135623730951);
        const size_t off = oracle_band_offset(Nr, Nc, ndim, do_swt, levels, per * (l - 1) + 1, &r, &c);
        const size_t n = (size_t)r * c;
        DATA *b = coeffs + off;
        DATA *a = (do_app && l == levels) ? coeffs : NULL;
        for (size_t i = 0; i < n; i++) {
            DATA nrm = 0;
            for (int k = 0; k < per; k++) nrm += b[k * n + i] * b[k * n + i];
            if (a) nrm += a[i] * a[i];
            nrm = sqrt(nrm);
            const DATA res = (nrm == 0) ? (DATA)0 : fmax(1.0f - beta / nrm, (DATA)0);
            for (int k = 0; k < per; k++) b[k * n + i] *= res;
            if (a) a[i] *= res;
        }
    }
}

/* shrink: x / (1 + beta) on every detail band, and on A when do_app
 * (common.cu:347-371) */
API void oracle_shrink(DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels, DATA beta,
                       int do_app) {
    const size_t total = oracle_coeff_count(Nr, Nc, ndim, do_swt, levels);
    int r, c;
    oracle_band_offset(Nr, Nc, ndim, do_swt, levels, 0, &r, &c);
    const DATA s = 1.0f / (1.0f + beta);
    for (size_t i = do_app ? 0 : (size_t)r * c; i < total; i++) coeffs[i] *= s;
}

/* norms over ALL bands (wt.cu:368-416).  The reference's 1D norm2sq sums
 * |x| instead of x^2 for detail bands (wt.cu:387); restated as the documented
 * squared L2 norm. */
API double oracle_norm1(const DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels) {
    const size_t total = oracle_coeff_count(Nr, Nc, ndim, do_swt, levels);
    double s = 0;
    for (size_t i = 0; i < total; i++) s += fabs((double)coeffs[i]);
    return s;
}

API double oracle_norm2sq(const DATA *coeffs, int Nr, int Nc, int ndim, int do_swt, int levels) {
    const size_t total = oracle_coeff_count(Nr, Nc, ndim, do_swt, levels);
    double s = 0;
    for (size_t i = 0; i < total; i++) s += (double)coeffs[i] * coeffs[i];
    return s;
}

/* circular shift (common.cu:202-211, 378-396): out[y,x] = in[(y-sr) mod Nr, (x-sc) mod Nc] */
API void oracle_circshift(const DATA *in, DATA *out, int Nr, int Nc, int sr, int sc) {
    for (int y = 0; y < Nr; y++)
        for (int x = 0; x < Nc; x++)
            out[(size_t)y * Nc + x] = in[(size_t)mod_n(y - sr, Nr) * Nc + mod_n(x - sc, Nc)];
}

/* ------------------------------------------------ non-separable 2D (one level)
 * nonseparable.cu:114-170: four hlen x hlen filters built as outer products
 * (w_outer(f_row..): LL = lo x lo, LH = lo x hi, HL = hi x lo, HH = hi x hi,
 * nonseparable.cu:70-74), indexed [jy][jx]; A,H,V,D use LL,LH,HL,HH with
 * out = sum_{jy,jx} x[per(2y-c+jy), per(2x-c+jx)] * F[(hlen-1-jy)*hlen + (hlen-1-jx)].
 * Filters are passed explicitly so user-supplied non-separable banks work too. */
API void oracle_nonsep_fwd_level(const DATA *in, int Nr, int Nc, const DATA *FA, const DATA *FH,
                                 const DATA *FV, const DATA *FD, int hlen, int do_swt, int level,
                                 DATA *A, DATA *H, DATA *V, DATA *D) {
    const int c = ana_centre(hlen);
    const int f = do_swt ? (1 << (level - 1)) : 1;
    const int Nr2 = do_swt ? Nr : oracle_div2(Nr), Nc2 = do_swt ? Nc : oracle_div2(Nc);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Nr2; y++)
        for (int x = 0; x < Nc2; x++) {
            REAL ra = 0, rh = 0, rv = 0, rd = 0;
            for (int jy = 0; jy < hlen; jy++) {
                const int sy = do_swt ? mod_n(y + (jy - c) * f, Nr) : per_src(2 * y - c + jy, Nr);
                for (int jx = 0; jx < hlen; jx++) {
                    const int sx = do_swt ? mod_n(x + (jx - c) * f, Nc) : per_src(2 * x - c + jx, Nc);
                    const REAL v = in[(size_t)sy * Nc + sx];
                    const int t = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                    ra += v * (REAL)FA[t]; rh += v * (REAL)FH[t];
                    rv += v * (REAL)FV[t]; rd += v * (REAL)FD[t];
                }
            }
            const size_t o = (size_t)y * Nc2 + x;
            A[o] = (DATA)ra; H[o] = (DATA)rh; V[o] = (DATA)rv; D[o] = (DATA)rd;
        }
}

/* Non-separable synthesis, one level.
 * DWT (nonseparable.cu:176-225): one thread per output pixel (gy,gx); h2 = hlen/2; for an odd h2 the
 * centre is c = h2/2, for an even h2 it is c = h2/2 and the pixel index is shifted by one ("virtual id",
 * :193-194, result written back at (g-1), :223); coefficient (g/2 - c + j) (periodic, :197-208) meets
 * tap hlen-1-(2j + off), off = 1 - (g & 1) (:202-203, :213-216); the four bands are summed (:222-223).
 * SWT (nonseparable.cu:360-401): c = hlen/2 times the dilation 2^(level-1), coefficient
 * (g - c + j f) (periodic), tap hlen-1-j, every product divided by 4 (:393-396).
 * Coefficient planes are (Nrc, Ncc); the output is (Nr, Nc) with Nr <= 2 Nrc (DWT) or == (SWT). */
API void oracle_nonsep_inv_level(const DATA *A, const DATA *H, const DATA *V, const DATA *D, int Nrc, int Ncc,
                                 int Nr, int Nc, const DATA *FA, const DATA *FH, const DATA *FV, const DATA *FD,
                                 int hlen, int do_swt, int level, DATA *out) {
#pragma omp parallel for schedule(static)
    for (int gy = 0; gy < Nr; gy++)
        for (int gx = 0; gx < Nc; gx++) {
            REAL r = 0;
            if (!do_swt) {
                const int h2 = hlen / 2, c = h2 / 2, shift = (h2 & 1) ? 0 : 1;
                const int vy = gy + shift, vx = gx + shift;
                const int offy = 1 - (vy & 1), offx = 1 - (vx & 1);
                for (int jy = 0; jy < h2; jy++) {
                    const int ty = hlen - 1 - (2 * jy + offy);
                    const int iy = mod_n(vy / 2 - c + jy, Nrc);
                    if (ty < 0) continue;
                    for (int jx = 0; jx < h2; jx++) {
                        const int tx = hlen - 1 - (2 * jx + offx);
                        const int ix = mod_n(vx / 2 - c + jx, Ncc);
                        if (tx < 0) continue;
                        const size_t o = (size_t)iy * Ncc + ix;
                        const int t = ty * hlen + tx;
                        r += (REAL)A[o] * (REAL)FA[t] + (REAL)H[o] * (REAL)FH[t] + (REAL)V[o] * (REAL)FV[t] +
                             (REAL)D[o] * (REAL)FD[t];
                    }
                }
            } else {
                const int f = 1 << (level - 1), c = (hlen / 2) * f;
                for (int jy = 0; jy < hlen; jy++) {
                    const int iy = mod_n(gy - c + jy * f, Nr);
                    for (int jx = 0; jx < hlen; jx++) {
                        const int ix = mod_n(gx - c + jx * f, Nc);
                        const size_t o = (size_t)iy * Nc + ix;
                        const int t = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                        r += ((REAL)A[o] * (REAL)FA[t] + (REAL)H[o] * (REAL)FH[t] + (REAL)V[o] * (REAL)FV[t] +
                              (REAL)D[o] * (REAL)FD[t]) / 4;
                    }
                }
            }
            out[(size_t)gy * Nc + gx] = (DATA)r;
        }
}

/* ------------------------------------------------------------ test inputs */

/* Counter-based generator shared with tests/golden/make_golden.py:hash_input
 * and the HIP fill kernel: lowbias32(i ^ seed) >> 8, scaled to [0, scale). */
API void oracle_fill_hash(DATA *x, size_t n, uint32_t seed, DATA scale) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint32_t h = (uint32_t)i ^ seed;
        h ^= h >> 16; h *= 0x7FEB352Du;
        h ^= h >> 15; h *= 0x846CA68Bu;
        h ^= h >> 16;
        x[i] = (DATA)((double)(h >> 8) * (1.0 / 16777216.0) * (double)scale);
    }
}

/* same generator with an index offset (image b of a batch: offset = b * Nr * Nc); the product's
 * pdwt_fill_image_hash takes the same parameter */
API void oracle_fill_hash_off(DATA *x, size_t n, uint32_t seed, DATA scale, long long index_offset) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint32_t h = (uint32_t)((long long)i + index_offset) ^ seed;
        h ^= h >> 16; h *= 0x7FEB352Du;
        h ^= h >> 15; h *= 0x846CA68Bu;
        h ^= h >> 16;
        x[i] = (DATA)((double)(h >> 8) * (1.0 / 16777216.0) * (double)scale);
    }
}

API int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
