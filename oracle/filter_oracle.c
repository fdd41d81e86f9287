/*
 * filter_oracle.c -- CPU restatement of Heuristic::filterPoints (heuristic.cpp:55-176): outlier / redundancy filter of
 * the point cloud after every iteration (recon.cpp:125).
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED (FLANN's randomised KD-tree and cv::sortIdx are not in
 * the reference tree).
 *
 * Followed literally, quirks included:
 *   - radius = alpha / 4 (heuristic.cpp:63) is compared with SQUARED distances: cvflann::L2_Simple returns squared
 *     distances and the reference uses them as distances (heuristic.cpp:81-89, SURVEY A-13); weight = 1 - d2/radius
 *   - only neighbours with a SMALLER index are stored ("to ensure symmetry", heuristic.cpp:86); the greedy pass
 *     therefore lowers the score of lower-index neighbours only (heuristic.cpp:152-154)
 *   - clamped (<= 2), L1-normalised power iteration until the mean squared change <= 1e-6 or 200 rounds (103-136)
 *   - points are visited by descending density; a point is kept if its remaining score >= 0.7 (139-163)
 * Where the reference is silent: FLANN's approximate search is replaced by the exact neighbourhood (brute force here),
 * neighbours are kept in ascending index order, and the descending-density order is a stable sort (ties by index).
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int idx;
    float w;
} Nb;

static const float *g_density;
static int cmp_desc(const void *a, const void *b)
{
    const int ia = *(const int *)a, ib = *(const int *)b;
    if (g_density[ia] > g_density[ib]) return -1;
    if (g_density[ia] < g_density[ib]) return 1;
    return ia - ib;
}

int orc_filter_points(const float *points4, int N, float alpha, int32_t *keep_out, float *density_out /* nullable, N */)
{
    if (N <= 0) return 0;
    float *p3 = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    for (int i = 0; i < N; i++)
        for (int c = 0; c < 3; c++) p3[3 * i + c] = points4[4 * i + c] / points4[4 * i + 3]; /* dehomogenize, util.cpp:16-29 */
    const float radius = alpha / 4.f;
    int *block = (int *)malloc(sizeof(int) * ((size_t)N + 1));
    size_t cap = 1024, cnt = 0;
    Nb *nb = (Nb *)malloc(sizeof(Nb) * cap);
    for (int i = 0; i < N; i++) {
        block[i] = (int)cnt;
        for (int j = 0; j < i; j++) {
            const float dx = p3[3 * i] - p3[3 * j], dy = p3[3 * i + 1] - p3[3 * j + 1], dz = p3[3 * i + 2] - p3[3 * j + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 <= radius) {
                if (cnt == cap) {
                    cap *= 2;
                    nb = (Nb *)realloc(nb, sizeof(Nb) * cap);
                }
                nb[cnt].idx = j;
                nb[cnt].w = (float)(1. - d2 / radius); /* densityFn, heuristic.cpp:49-52 */
                cnt++;
            }
        }
    }
    block[N] = (int)cnt;

    float *density = (float *)malloc(sizeof(float) * (size_t)N), *score = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; i++) density[i] = 1.f;
    double change;
    int it = 0;
    do {
        for (int i = 0; i < N; i++) score[i] = 0.f;
        double sum = 0.;
        for (int i = 0; i < N; i++) {
            float densityTemp = 0.f;
            for (int k = block[i]; k < block[i + 1]; k++) {
                densityTemp += density[nb[k].idx] * nb[k].w;
                score[nb[k].idx] += density[i] * nb[k].w;
                sum += (density[i] + density[nb[k].idx]) * nb[k].w;
            }
            score[i] += densityTemp;
        }
        const float normalizer = (float)(N / sum);
        change = 0.;
        for (int i = 0; i < N; i++) {
            float nd = score[i] * normalizer;
            if (nd > 2.f) nd = 2.f;
            const float df = density[i] - nd;
            change += df * df;
            density[i] = nd;
        }
        change /= N;
        it++;
    } while (change > 1e-6 && it < 200);
    if (density_out) memcpy(density_out, density, sizeof(float) * (size_t)N);

    int *order = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; i++) order[i] = i;
    g_density = density;
    qsort(order, (size_t)N, sizeof(int), cmp_desc);
    const float densityLimit = .7f;
    int kept = 0;
    uint8_t *keep = (uint8_t *)calloc((size_t)N, 1);
    for (int i = 0; i < N; i++) {
        const int ord = order[i];
        if (score[ord] < densityLimit) continue;
        const double localDensity = density[ord];
        for (int k = block[ord]; k < block[ord + 1]; k++) score[nb[k].idx] = (float)(score[nb[k].idx] - localDensity * nb[k].w);
        keep[ord] = 1;
        kept++;
    }
    int m = 0;
    for (int i = 0; i < N; i++)
        if (keep[i]) keep_out[m++] = i; /* std::sort(order...) + in-place compaction, heuristic.cpp:166-175 */
    free(keep);
    free(order);
    free(score);
    free(density);
    free(nb);
    free(block);
    free(p3);
    return kept;
}
