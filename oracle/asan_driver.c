/* Memory-safety driver for the oracle (CPU AddressSanitizer + UBSan; GPU ASan is not available
 * on the pool).  Runs odd sizes, K = 3, dmin > 0, tiny images whose windows wrap several times
 * (SURVEY.md Appendix C.6).  Test infrastructure only. */
#include <stdio.h>
#include <stdlib.h>
#include "stereo_oracle.h"

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 255); }

int main(void) {
    const int cases[][5] = {  /* H, W, K, min, max */
        {17, 23, 2, 0, 9}, {21, 31, 3, 0, 11}, {12, 18, 1, 0, 5}, {40, 24, 1, 0, 31},
        {37, 53, 2, 6, 25}, {33, 47, 4, 0, 15}, {9, 11, 2, 0, 3}, {64, 96, 2, 75, 131}};
    unsigned seed = 7;
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
        so_config cfg;
        so_default_config(&cfg);
        cfg.height = cases[c][0]; cfg.width = cases[c][1]; cfg.downscale_factor = cases[c][2];
        cfg.min_disparity = cases[c][3]; cfg.max_disparity = cases[c][4];
        const size_t HW = (size_t)cfg.height * cfg.width;
        float *l = malloc(3 * HW * sizeof(float)), *r = malloc(3 * HW * sizeof(float)), *o = malloc(HW * sizeof(float));
        for (size_t i = 0; i < 3 * HW; ++i) { l[i] = frand(&seed); r[i] = frand(&seed); }
        if (so_run_gray(&cfg, l, r, o, NULL) || so_run_rgb(&cfg, l, r, o, NULL)) { printf("case %u failed\n", c); return 1; }
        unsigned char *md = malloc(HW), *mf = malloc(HW);
        so_validity_masks(&cfg, md, mf);
        free(md); free(mf); free(l); free(r); free(o);
    }
    printf("asan driver ok\n");
    return 0;
}
