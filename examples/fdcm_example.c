/* fdcm_example.c -- the C ABI of libfdcm_hip.so from plain C (C99): the sequence a binding of the
 * reference would make for one frame, with the host-only tail.  Build:
 *     gcc -std=c99 -I include examples/fdcm_example.c -o fdcm_example -L openfdcm_amd -lfdcm_hip \
 *         -Wl,-rpath,$PWD/openfdcm_amd -lm
 * Without a HIP device the compute calls fail with FDCM_EHIP and the program says so (there is no
 * CPU fallback); the host-only entry points (filter, penalize, sort) run everywhere. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fdcm.h"

/* Config 1 of BASELINE.json from the reference's own asset files: `fdcm_example <dir>` reads <dir>/scene_0.scene and
 * <dir>/template_0.tmpl, template_1.tmpl, .. (as many as there are) with fdcm_lines_read, builds the feature map
 * (depth 30, coeff 5, padding 1.0, L2), searches with DefaultSearch(4, 4) + BatchOptimize(10) and prints the match
 * count and the three best matches after ExponentialPenalty(1.5) + sort -- the notebook's sequence. */
static int run_assets(const char* dir) {
    char path[1024];
    float* scene = NULL;
    int64_t n_scene = 0;
    snprintf(path, sizeof(path), "%s/scene_0.scene", dir);
    if (fdcm_lines_read(path, &scene, &n_scene) != FDCM_OK) { printf("%s\n", fdcm_last_error()); return 2; }
    float* all = NULL;
    int64_t cap = 0, n_all = 0, T = 0;
    int64_t* offsets = (int64_t*)malloc(sizeof(int64_t));
    offsets[0] = 0;
    for (;;) {
        float* t = NULL;
        int64_t n = 0;
        snprintf(path, sizeof(path), "%s/template_%lld.tmpl", dir, (long long)T);
        if (fdcm_lines_read(path, &t, &n) != FDCM_OK) break;  /* the first missing number ends the list */
        if (n_all + n > cap) { cap = 2 * (n_all + n); all = (float*)realloc(all, (size_t)cap * 4 * sizeof(float)); }
        for (int64_t i = 0; i < 4 * n; ++i) all[4 * n_all + i] = t[i];
        fdcm_lines_free(t);
        n_all += n;
        ++T;
        offsets = (int64_t*)realloc(offsets, (size_t)(T + 1) * sizeof(int64_t));
        offsets[T] = n_all;
    }
    printf("assets: %lld scene lines, %lld templates, %lld template lines\n", (long long)n_scene, (long long)T, (long long)n_all);
    /* write + read back: the writer makes files the reader (and the reference) takes */
    snprintf(path, sizeof(path), "%s", "/tmp/fdcm_example_roundtrip.lines");
    float* back = NULL;
    int64_t n_back = 0;
    if (fdcm_lines_write(path, scene, n_scene) != FDCM_OK || fdcm_lines_read(path, &back, &n_back) != FDCM_OK || n_back != n_scene) {
        printf("round trip failed: %s\n", fdcm_last_error());
        return 2;
    }
    for (int64_t i = 0; i < 4 * n_scene; ++i) if (back[i] != scene[i]) { printf("round trip differs\n"); return 2; }
    fdcm_lines_free(back);
    remove(path);
    fdcm_featuremap* fm = NULL;
    fdcm_templates* ts = NULL;
    int ndev = 0;
    if (fdcm_device_count(&ndev) != FDCM_OK || ndev == 0 || fdcm_featuremap_build(scene, n_scene, 30, 5.f, 1.0f, FDCM_L2, &fm) != FDCM_OK) {
        printf("no HIP device: %s\n", fdcm_last_error());
        return 0;
    }
    fdcm_match* out = NULL;
    int64_t n = 0;
    if (fdcm_templates_create(all, offsets, T, &ts) != FDCM_OK ||
        fdcm_search(fm, ts, scene, n_scene, 4, 4, FDCM_BATCH_OPTIMIZE, 10, 0, &out, &n) != FDCM_OK) {
        printf("%s\n", fdcm_last_error());
        return 2;
    }
    fdcm_featuremap_info info;
    fdcm_featuremap_get_info(fm, &info);
    printf("config 1: feature size %lld x %lld x %lld, %lld raw matches\n", (long long)info.width, (long long)info.height,
           (long long)info.depth, (long long)n);
    fdcm_match* best = NULL;
    int64_t nb = 0;
    if (fdcm_topk(fm, ts, NULL, 0, 0, FDCM_EXPONENTIAL_PENALTY, 1.5f, 3, &best, &nb) != FDCM_OK) { printf("%s\n", fdcm_last_error()); return 2; }
    for (int64_t i = 0; i < nb; ++i)
        printf("  best %lld: template %d score %.9g\n", (long long)i, best[i].tmpl_idx, best[i].score);
    fdcm_matches_free(best);
    fdcm_matches_free(out);
    fdcm_templates_free(ts);
    fdcm_featuremap_free(fm);
    fdcm_lines_free(scene);
    free(all);
    free(offsets);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) return run_assets(argv[1]);
    /* a 4-line scene and one 3-line template, x1 y1 x2 y2 per line (the reference's LineArray columns) */
    const float scene[] = {0, 0, 40, 0, 40, 0, 40, 30, 40, 30, 0, 30, 0, 30, 0, 0};
    const float tmpl[] = {2, 2, 22, 2, 22, 2, 22, 17, 22, 17, 2, 17};
    const int64_t offsets[] = {0, 3};
    printf("%s\n", fdcm_version());

    /* host-only: ConcentricRangeStrategy's filter, penalize, sort_matches */
    int64_t idx[4], n_in = 0;
    const float center[2] = {20, 15};
    if (fdcm_filter_in_range(scene, 4, center, 10.f, 18.f, idx, &n_in) != FDCM_OK) return 2;
    printf("lines with centre in [10, 18) of (20, 15): %lld\n", (long long)n_in);
    fdcm_match m[3] = {{0, 30.f, {1, 0, 0, 0, 1, 0}}, {0, 10.f, {1, 0, 0, 0, 1, 0}}, {0, 20.f, {1, 0, 0, 0, 1, 0}}};
    const float len[1] = {55.f};
    if (fdcm_penalize(FDCM_EXPONENTIAL_PENALTY, 1.5f, m, 3, len, 1) != FDCM_OK) return 2;
    if (fdcm_sort_matches(m, 3) != FDCM_OK) return 2;
    printf("best penalised score %.6f (expected %.6f)\n", m[0].score, 10.f / powf(55.f, 1.5f));

    /* device: build + search + device tail */
    int ndev = 0;
    fdcm_featuremap* fm = NULL;
    fdcm_templates* ts = NULL;
    if (fdcm_device_count(&ndev) != FDCM_OK || ndev == 0 ||
        fdcm_featuremap_build(scene, 4, 30, 5.f, 2.2f, FDCM_L2, &fm) != FDCM_OK) {
        printf("no HIP device: %s\n", fdcm_last_error());
        return 0;
    }
    if (fdcm_templates_create(tmpl, offsets, 1, &ts) != FDCM_OK) { printf("%s\n", fdcm_last_error()); return 2; }
    fdcm_match* out = NULL;
    int64_t n = 0;
    if (fdcm_search(fm, ts, scene, 4, 4, 4, FDCM_BATCH_OPTIMIZE, 10, 0, &out, &n) != FDCM_OK) {
        printf("%s\n", fdcm_last_error());
        return 2;
    }
    fdcm_featuremap_info info;
    fdcm_featuremap_get_info(fm, &info);
    printf("feature size %lld x %lld x %lld, %lld raw matches\n", (long long)info.width, (long long)info.height,
           (long long)info.depth, (long long)n);
    fdcm_match* best = NULL;
    int64_t nb = 0;
    if (fdcm_topk(fm, ts, NULL, 0, 0, FDCM_DEFAULT_PENALTY, 1.f, 3, &best, &nb) == FDCM_OK) {
        for (int64_t i = 0; i < nb; ++i)
            printf("  #%lld score %.6f  t = (%.3f, %.3f)\n", (long long)i, best[i].score, best[i].transform[2], best[i].transform[5]);
        fdcm_matches_free(best);
    }
    /* the feature-map seam (featuremap.h:27-52): what an optimiser that lives outside the library calls */
    {
        const float align_vec[2] = {1.f, 0.f};
        float lim[2] = {0, 0};
        const float moved[6] = {0, 0, 3, 0, -2, 1};  /* three translations of the template */
        const int64_t toff[] = {0, 3};
        float score[3] = {0, 0, 0};
        if (fdcm_featuremap_minmax_translation(fm, tmpl, 3, align_vec, lim) != FDCM_OK ||
            fdcm_featuremap_evaluate(fm, tmpl, offsets, 1, moved, toff, score) != FDCM_OK) {
            printf("%s\n", fdcm_last_error());
            return 2;
        }
        printf("seam: multipliers of (1, 0) in [%.0f, %.0f]; scores %.4f %.4f %.4f\n", lim[0], lim[1], score[0], score[1], score[2]);
    }
    fdcm_matches_free(out);
    fdcm_templates_free(ts);
    fdcm_featuremap_free(fm);
    return 0;
}
