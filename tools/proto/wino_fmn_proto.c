/* Numerics prototype (NOT product code, NOT the oracle): Winograd F(m x m, 3 x 3) in float32 with the arithmetic a matrix-core kernel
 * would have -- input transform V = B^T d B in float32 (fixed order: rows first, zero coefficients skipped, fmaf chains in index
 * order), per frequency one SEQUENTIAL fmaf chain over the input channels (what v_mfma_f32_32x32x2_f32 does with its k's,
 * tools/mfma_order.hip), output transform Y = A^T M A in float32 (fixed order).  The transformed weights U = G g G^T come from the host
 * (computed in double, rounded once), as pmx_api.hip::pack_wino does for F(2x2, 3x3).
 * tools/wino_f4_check.py drives it: per-layer error against a float64 convolution next to the direct fp32 chain and F(2x2, 3x3).
 * n = m + 2 points; BT is n x n, AT is m x n (row-major); tf64 = 1: the input transform is computed in double and rounded once. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void wino_fmn_proto(const float* x, const float* U, const double* BT, const double* AT, int m, int cin, int H, int W, int cout, int relu,
                    const float* bias, float* y, int tf64)
{
    const int n = m + 2, nf = n * n;
    const int ty = (H + m - 1) / m, tx = (W + m - 1) / m;
#pragma omp parallel
    {
        float* V = (float*)malloc(sizeof(float) * (size_t)nf * cin);
        float* M = (float*)malloc(sizeof(float) * (size_t)nf);
#pragma omp for collapse(2) schedule(dynamic)
        for (int by = 0; by < ty; ++by)
            for (int bx = 0; bx < tx; ++bx) {
                const int y0 = by * m - 1, x0 = bx * m - 1;
                for (int c = 0; c < cin; ++c) {
                    float d[8][8];
                    for (int i = 0; i < n; ++i)
                        for (int j = 0; j < n; ++j) {
                            const int gy = y0 + i, gx = x0 + j;
                            d[i][j] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)c * H + gy) * W + gx] : 0.f;
                        }
                    if (tf64) {
                        double t[8][8], v;
                        for (int i = 0; i < n; ++i)
                            for (int j = 0; j < n; ++j) { v = 0; for (int k = 0; k < n; ++k) v += BT[i * n + k] * (double)d[k][j]; t[i][j] = v; }
                        for (int i = 0; i < n; ++i)
                            for (int j = 0; j < n; ++j) { v = 0; for (int k = 0; k < n; ++k) v += t[i][k] * BT[j * n + k]; V[(size_t)(i * n + j) * cin + c] = (float)v; }
                    } else {
                        float t[8][8], v;
                        for (int i = 0; i < n; ++i)
                            for (int j = 0; j < n; ++j) {
                                v = 0.f; int first = 1;
                                for (int k = 0; k < n; ++k) { const float b = (float)BT[i * n + k]; if (b == 0.f) continue; v = first ? b * d[k][j] : fmaf(b, d[k][j], v); first = 0; }
                                t[i][j] = v;
                            }
                        for (int i = 0; i < n; ++i)
                            for (int j = 0; j < n; ++j) {
                                v = 0.f; int first = 1;
                                for (int k = 0; k < n; ++k) { const float b = (float)BT[j * n + k]; if (b == 0.f) continue; v = first ? t[i][k] * b : fmaf(t[i][k], b, v); first = 0; }
                                V[(size_t)(i * n + j) * cin + c] = v;
                            }
                    }
                }
                for (int co = 0; co < cout; ++co) {
                    for (int f = 0; f < nf; ++f) {
                        const float* u = U + ((size_t)f * cout + co) * cin;
                        const float* v = V + (size_t)f * cin;
                        float acc = 0.f;
                        for (int c = 0; c < cin; ++c) acc = fmaf(v[c], u[c], acc);
                        M[f] = acc;
                    }
                    float t[8][8];
                    for (int i = 0; i < m; ++i)
                        for (int j = 0; j < n; ++j) {
                            float v = 0.f; int first = 1;
                            for (int k = 0; k < n; ++k) { const float a = (float)AT[i * n + k]; if (a == 0.f) continue; v = first ? a * M[k * n + j] : fmaf(a, M[k * n + j], v); first = 0; }
                            t[i][j] = v;
                        }
                    for (int i = 0; i < m; ++i)
                        for (int j = 0; j < m; ++j) {
                            float v = 0.f; int first = 1;
                            for (int k = 0; k < n; ++k) { const float a = (float)AT[j * n + k]; if (a == 0.f) continue; v = first ? t[i][k] * a : fmaf(t[i][k], a, v); first = 0; }
                            const int gy = by * m + i, gx = bx * m + j;
                            if (gy < H && gx < W) {
                                v += bias ? bias[co] : 0.f;
                                if (relu && v < 0.f) v = 0.f;
                                y[((size_t)co * H + gy) * W + gx] = v;
                            }
                        }
                }
            }
        free(V); free(M);
    }
}
