/* Plain-C user of the drop-in boundary (include/pypwt_amd.h): no C++, no Python, no torch.
 *   gcc -std=c99 -I include tests/c_abi/roundtrip.c -L pypwt_amd -lpypwt_amd -Wl,-rpath,$PWD/pypwt_amd -lm
 * Exit code 0 = forward / coefficient read-back / soft threshold / inverse behaved, 77 = no GPU. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pypwt_amd.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, pdwt_last_error()); fails++; } } while (0)

int main(void) {
    const int Nr = 96, Nc = 120, levels = 2;
    float* img = (float*)malloc(sizeof(float) * Nr * Nc);
    float* rec = (float*)malloc(sizeof(float) * Nr * Nc);
    for (int i = 0; i < Nr * Nc; i++) img[i] = (float)((i * 2654435761u) >> 20 & 255);

    pdwt_handle h = NULL;
    int rc = pdwt_create(img, Nr, Nc, "nope", levels, 1, 1, 0, 0, 2, &h);
    EXPECT(rc == PDWT_ERR_WAVELET && h == NULL);
    rc = pdwt_create(img, Nr, Nc, "db3", levels, 1, 1, 0, 0, 2, &h);
    if (rc == PDWT_ERR_HIP) { printf("no HIP device: %s\n", pdwt_last_error()); return 77; }
    EXPECT(rc == PDWT_OK && h != NULL);

    pdwt_info info; int sep, cyc, state, batch;
    EXPECT(pdwt_get_info(h, &info, &sep, &cyc, &state, &batch) == PDWT_OK);
    EXPECT(info.nlevels == levels && info.hlen == 6 && info.ndims == 2 && state == PDWT_INIT && batch == 1);

    EXPECT(pdwt_forward(h) == PDWT_OK);
    int r, c;
    long long n = pdwt_coeff_count(h, 0, &r, &c);
    EXPECT(n == 24 * 30 && r == 24 && c == 30);
    float* A = (float*)malloc(sizeof(float) * (size_t)n);
    EXPECT(pdwt_get_coeff(h, A, 0) == n);
    /* energy of an orthogonal transform is preserved: ||coeffs||^2 == ||img||^2 */
    float n2 = 0.f; double e = 0.0;
    EXPECT(pdwt_norm2sq(h, &n2) == PDWT_OK);
    for (int i = 0; i < Nr * Nc; i++) e += (double)img[i] * img[i];
    EXPECT(fabs((double)n2 - e) < 1e-4 * e);

    EXPECT(pdwt_soft_threshold(h, 0.0f, 0, 0) == PDWT_OK); /* beta = 0: identity */
    EXPECT(pdwt_inverse(h) == PDWT_OK);
    EXPECT(pdwt_inverse(h) == PDWT_ERR_STATE);              /* twice in a row: refused (wt.cu:272-275) */
    EXPECT(pdwt_get_coeff(h, A, 0) == 0);                   /* refused after inverse (wt.cu:474-477) */
    EXPECT(pdwt_get_image(h, rec) == (long long)Nr * Nc);
    float maxerr = 0.f;
    for (int i = 0; i < Nr * Nc; i++) { float d = fabsf(rec[i] - img[i]); if (d > maxerr) maxerr = d; }
    printf("round-trip max error %.3g\n", maxerr);
    EXPECT(maxerr < 7e-4f);                                 /* idwt2 tolerance, test_wavelets.py:545 */
    EXPECT(pdwt_image_ptr(h) != 0 && pdwt_coeff_ptr(h, 1) != 0);
    EXPECT(pdwt_print_info(h) == PDWT_OK);
    EXPECT(pdwt_destroy(h) == PDWT_OK);
    free(img); free(rec); free(A);
    printf(fails ? "C ABI round trip: %d failure(s)\n" : "C ABI round trip: ok\n", fails);
    return fails ? 1 : 0;
}
