/* CPU oracle (TEST INFRASTRUCTURE, never linked into the product): plain-C restatement of the reference's
 * transform path, float64 arithmetic on float32 windowed frames exactly like the reference.
 *
 *   oracle_mdct4       <- models/mdct.py:392-425  (MDCT4.forward: pad, frame, window (float32), cosine transform)
 *   oracle_imdct4      <- models/mdct.py:457-489  (IMDCT4.forward: transform, window, fold overlap-add, 4/N, crop)
 *   oracle_normalize   <- models/pix2pixHD_model.py:96-123 (arcsinh branch + range norm, --abs_norm constants)
 *   oracle_denormalize <- models/pix2pixHD_model.py:127-133
 *
 * The reference evaluates twiddle * FFT * twiddle in complex128 (mdct.py:387-390, 421-423); the real part of that
 * product is the cosine sum below.  Pinned against tests/golden (captured from the reference) by
 * tests/test_oracle_golden.py::test_c_oracle.  Built by `python -m mdctgan_amd.build` (gcc -O2 -fopenmp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* x [B][T] float32 -> spec [B][F][N/2] float64 (+ frames [B][F][N] float32 if non-NULL); hop = N/2, centre padding */
int oracle_mdct4(const float* x, int B, int T, int N, const float* window, double* spec, float* frames) {
    const int hop = N / 2;
    const int tail = (T % hop) ? hop - T % hop : 0;
    const int F = (T + tail) / hop + 1;
    double* C = (double*)malloc(sizeof(double) * N * hop);
    if (!C) return -1;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < hop; ++k) C[n * hop + k] = cos((2.0 * M_PI / N) * (n + 0.5 + N / 4.0) * (k + 0.5));
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f) {
            float z[4096];
            for (int n = 0; n < N; ++n) {
                const int t = f * hop + n - hop;
                const float v = (t >= 0 && t < T) ? x[(size_t)b * T + t] : 0.0f;
                z[n] = v * window[n]; /* float32 multiply, mdct.py:410 */
                if (frames) frames[((size_t)b * F + f) * N + n] = z[n];
            }
            for (int k = 0; k < hop; ++k) {
                double s = 0.0;
                for (int n = 0; n < N; ++n) s += (double)z[n] * C[n * hop + k];
                spec[((size_t)b * F + f) * hop + k] = s;
            }
        }
    free(C);
    return F;
}

/* spec [B][F][N/2] float64 -> audio [B][(F-1)*hop] float64 */
int oracle_imdct4(const double* spec, int B, int F, int N, const float* window, double* audio) {
    const int hop = N / 2, out_len = (F - 1) * hop + N, T = (F - 1) * hop;
    double* C = (double*)malloc(sizeof(double) * N * hop);
    double* full = (double*)calloc((size_t)B * out_len, sizeof(double));
    if (!C || !full) return -1;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < hop; ++k) C[n * hop + k] = cos((2.0 * M_PI / N) * (n + 0.5 + N / 4.0) * (k + 0.5));
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            for (int n = 0; n < N; ++n) {
                double s = 0.0;
                const double* X = spec + ((size_t)b * F + f) * hop;
                for (int k = 0; k < hop; ++k) s += X[k] * C[n * hop + k];
                full[(size_t)b * out_len + f * hop + n] += s * (double)window[n];   /* fold == overlap-add */
            }
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) audio[(size_t)b * T + t] = full[(size_t)b * out_len + t + N / 2] * 4.0 / N;
    free(C);
    free(full);
    return T;
}

/* L = asinh(gain * X) / float32(ln 10); out = (L - mn)/(mx - mn) * (nr1 - nr0) + nr0, rounded to float32 */
void oracle_normalize(const double* X, long long n, double gain, double mn, double mx, double nr0, double nr1,
                      float* out) {
    const double ln10 = (double)(float)log(10.0);
#pragma omp parallel for
    for (long long i = 0; i < n; ++i) out[i] = (float)((asinh(gain * X[i]) / ln10 - mn) / (mx - mn) * (nr1 - nr0) + nr0);
}

void oracle_denormalize(const float* v, long long n, double gain, double mn, double mx, double nr0, double nr1,
                        double* X) {
    const double ln10 = (double)(float)log(10.0);
#pragma omp parallel for
    for (long long i = 0; i < n; ++i)
        X[i] = sinh((((double)v[i] - nr0) / (nr1 - nr0) * (mx - mn) + mn) * ln10) / gain;
}
