/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * CPU restatement of STRique's flank alignment `align_raw<float,float>::semiglobal`
 *   reference: src/align_raw.h:106-158   (align<true,false>, view positions)
 *              src/score_distance.h:115-122 (score = max(off - (float)pow(|h-v|,1.2), dmin))
 *              src/score_distance.h:140-226 (constant gap scores)
 *              src/pyalign.cpp:47-61     (binding: align_raw<float,float>, align_overlap)
 *
 * The DP itself lives in SeqAn 2 (`globalAlignment(..., AlignConfig<true,false,false,true>,
 * AffineGaps())`, called at src/align_raw.h:134-135).  SeqAn is an un-vendored git submodule
 * (.gitmodules:4-6, no pinned SHA in the mounted tree; `submodules/seqan/` is empty) so the
 * reference cannot be compiled here.  This file restates SeqAn's published affine-gap
 * (Gotoh) semi-global algorithm as laid out in SURVEY.md Appendix A.1.
 *
 * PARITY STATUS: "parity unpinned" at the SeqAn boundary -- the reference holds no golden
 * vector for align_overlap's raw score / index lists.  Integer geometry is pinned indirectly
 * (docs/installation/test.md:15-16: offset 1633, ticks 40758; scripts/STRique_test.py counts).
 * Tie-break rules are isolated in the TIE_* macros below.
 *
 * Full (N+1)x(M+1) matrix, float32 cells, one trace byte per cell, one double pow per cell:
 * the same work the reference does per alignment, so this is also the timed CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* tie rules: H/V "extend" wins over "open"; H wins over V; diagonal wins over gap.
 * oracle/residual_probe.py recompiles this file with the alternatives (-DTIE_...=...) and with the
 * ORACLE_* switches below to show which enumerated semantics move the documented known answer. */
#ifndef TIE_EXT
#define TIE_EXT(ext, opn)   ((ext) >= (opn))
#endif
#ifndef TIE_H_OVER_V
#define TIE_H_OVER_V(h, v)  ((h) >= (v))
#endif
#ifndef TIE_D_OVER_G
#define TIE_D_OVER_G(d, g)  ((d) >= (g))
#endif
/* probe switches (all off in the oracle proper):
 *   ORACLE_GAP_OPEN_PLUS_EXT  a gap of n costs open + n*ext instead of SeqAn's open + (n-1)*ext
 *   ORACLE_POWF               powf in float instead of pow in double
 *   ORACLE_LINEAR             |h-v| without the exponent (the commented-out line src/score_distance.h:119)
 *   ORACLE_EXPONENT           the exponent (default 1.2) */
#ifndef ORACLE_EXPONENT
#define ORACLE_EXPONENT 1.2
#endif

#define TR_DIR_MASK 3u   /* 0 = diagonal, 1 = from H, 2 = from V */
#define TR_HEXT 4u
#define TR_VEXT 8u

typedef struct {
    float open_h, ext_h, open_v, ext_v, dist_offset, dist_min;
} strq_oracle_params;

/* src/score_distance.h:117-122 -- TValue=float: difference in float, pow in double, cast back */
static inline float cell_score(const strq_oracle_params *p, float h, float v)
{
    float d = h > v ? h - v : v - h;
#if defined(ORACLE_LINEAR)
    float s = p->dist_offset - d;
#elif defined(ORACLE_POWF)
    float s = p->dist_offset - powf(d, (float)ORACLE_EXPONENT);
#else
    float s = p->dist_offset - (float)pow((double)d, ORACLE_EXPONENT);
#endif
    return s > p->dist_min ? s : p->dist_min;
}

/* exported so that tests / LUT checks can evaluate single cells */
float strq_oracle_cell_score(const float params[6], float h, float v)
{
    strq_oracle_params p = { params[0], params[1], params[2], params[3], params[4], params[5] };
    return cell_score(&p, h, v);
}

/*
 * a: read signal (horizontal, n), b: flank (vertical, m).
 * params = {open_h, ext_h, open_v, ext_v, dist_offset, dist_min}.
 * outputs:
 *   score   best score (row m, leftmost maximum over columns 0..n)
 *   j_end   DP column of that maximum, j0 = DP column where the path leaves row 0
 *   rec[m]  per flank row k: (j << 1) | is_vertical
 *             diagonal: j = DP column (b[k] aligned to a[j-1]);
 *             vertical: j = number of samples of `a` consumed before b[k]
 *   a_idx[n], b_idx[m] (nullable): view positions exactly as src/align_raw.h:141-146 returns
 *   use_lut: 0 = per-cell double pow (reference arithmetic / cost),
 *            1 = memoise cell_score over distinct (a,b) values (same bits, faster; "LUT CPU variant")
 * returns 0 on success.
 */
int strq_oracle_align(const float *a, int64_t n, const float *b, int64_t m,
                      const float params[6],
                      float *score, int64_t *j_end_out, int64_t *j0_out,
                      int32_t *rec, uint64_t *a_idx, uint64_t *b_idx, int use_lut)
{
    if (n < 0 || m < 1) return 1;
    strq_oracle_params p = { params[0], params[1], params[2], params[3], params[4], params[5] };
#ifdef ORACLE_GAP_OPEN_PLUS_EXT
    p.open_h += p.ext_h; p.open_v += p.ext_v;
#endif
    const float NINF = -FLT_MAX / 2;
    const int64_t rows = m + 1;
    uint8_t *trace = (uint8_t *)malloc((size_t)(n + 1) * rows);
    float *S0 = (float *)malloc(sizeof(float) * rows), *S1 = (float *)malloc(sizeof(float) * rows);
    float *H0 = (float *)malloc(sizeof(float) * rows), *H1 = (float *)malloc(sizeof(float) * rows);
    float *V = (float *)malloc(sizeof(float) * rows);
    /* optional memo: dictionary-encode a and b */
    int32_t *acode = NULL, *bcode = NULL; float *lut = NULL; int64_t na = 0, nb = 0;
    if (!trace || !S0 || !S1 || !H0 || !H1 || !V) return 2;
    if (use_lut) {
        /* simple O(n * distinct) encode is too slow for big n; sort-based encode instead */
        acode = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
        bcode = (int32_t *)malloc(sizeof(int32_t) * m);
        float *av = (float *)malloc(sizeof(float) * (n ? n : 1)), *bv = (float *)malloc(sizeof(float) * m);
        /* open addressing on the float bit pattern */
        int64_t cap = 1; while (cap < 4 * (n + m) + 16) cap <<= 1;
        int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * cap);
        uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * cap);
        for (int pass = 0; pass < 2; ++pass) {
            const float *src = pass ? b : a; int64_t len = pass ? m : n;
            int32_t *code = pass ? bcode : acode; float *vals = pass ? bv : av; int64_t cnt = 0;
            memset(slot, 0xff, sizeof(int32_t) * cap);
            for (int64_t i = 0; i < len; ++i) {
                uint32_t u; memcpy(&u, &src[i], 4);
                uint64_t h = (u * 2654435761u) & (cap - 1);
                while (slot[h] >= 0 && key[h] != u) h = (h + 1) & (cap - 1);
                if (slot[h] < 0) { slot[h] = (int32_t)cnt; key[h] = u; vals[cnt++] = src[i]; }
                code[i] = slot[h];
            }
            if (pass) nb = cnt; else na = cnt;
        }
        free(slot); free(key);
        if (na * nb > (int64_t)1 << 28) { use_lut = 0; }
        else {
            lut = (float *)malloc(sizeof(float) * ((na * nb) != 0 ? na * nb : 1));
            for (int64_t x = 0; x < na; ++x)
                for (int64_t y = 0; y < nb; ++y) lut[x * nb + y] = cell_score(&p, av[x], bv[y]);
        }
        free(av); free(bv);
    }

    /* column 0: not free (AlignConfig LEFT=false) */
    S0[0] = 0.0f; H0[0] = NINF; V[0] = NINF;
    {
        float vprev = NINF, sprev = 0.0f;
        for (int64_t i = 1; i <= m; ++i) {
            float ext = vprev + p.ext_v, opn = sprev + p.open_v;
            float v = TIE_EXT(ext, opn) ? ext : opn;
            uint8_t tr = 2u | (TIE_EXT(ext, opn) ? TR_VEXT : 0u);
            S0[i] = v; H0[i] = NINF; vprev = v; sprev = v;
            trace[i] = tr;
        }
        trace[0] = 0;
    }
    float best = S0[m]; int64_t best_j = 0;
    for (int64_t j = 1; j <= n; ++j) {
        uint8_t *tr = trace + (size_t)j * rows;
        const float aj = a[j - 1];
        const float *lrow = lut ? lut + (int64_t)acode[j - 1] * nb : NULL;
        /* row 0 is free (AlignConfig TOP=true) */
        S1[0] = 0.0f; H1[0] = NINF; V[0] = NINF; tr[0] = 0;
        for (int64_t i = 1; i <= m; ++i) {
            float sc = lrow ? lrow[bcode[i - 1]] : cell_score(&p, aj, b[i - 1]);
            float D = S0[i - 1] + sc;
            float hext = H0[i] + p.ext_h, hopn = S0[i] + p.open_h;
            int he = TIE_EXT(hext, hopn);
            float Hn = he ? hext : hopn;
            float vext = V[i - 1] + p.ext_v, vopn = S1[i - 1] + p.open_v;
            int ve = TIE_EXT(vext, vopn);
            float Vn = ve ? vext : vopn;
            int gh = TIE_H_OVER_V(Hn, Vn);
            float G = gh ? Hn : Vn;
            int dd = TIE_D_OVER_G(D, G);
            S1[i] = dd ? D : G; H1[i] = Hn; V[i] = Vn;
            tr[i] = (uint8_t)((dd ? 0u : (gh ? 1u : 2u)) | (he ? TR_HEXT : 0u) | (ve ? TR_VEXT : 0u));
        }
        if (S1[m] > best) { best = S1[m]; best_j = j; }   /* strict: leftmost maximum */
        float *t = S0; S0 = S1; S1 = t; t = H0; H0 = H1; H1 = t;
    }
    /* traceback */
    int64_t i = m, j = best_j; int state = 0; /* 0=S 1=H 2=V */
    /* ops are produced back to front; count them to know view columns */
    int64_t nops = 0;
    uint8_t *ops = (uint8_t *)malloc((size_t)(n + m + 2));
    while (i > 0) {
        uint8_t tr = trace[(size_t)j * rows + i];
        if (state == 0) {
            unsigned d = tr & TR_DIR_MASK;
            if (d == 0) { ops[nops++] = 'D'; rec[i - 1] = (int32_t)(j << 1); --i; --j; }
            else state = (int)d;
        } else if (state == 1) {
            ops[nops++] = 'H'; int ext = (tr & TR_HEXT) != 0; --j; if (!ext) state = 0;
        } else {
            ops[nops++] = 'V'; int ext = (tr & TR_VEXT) != 0; rec[i - 1] = (int32_t)((j << 1) | 1); --i; if (!ext) state = 0;
        }
    }
    int64_t j0 = j;
    *score = best; *j_end_out = best_j; *j0_out = j0;
    if (a_idx || b_idx) {
        int64_t ai = j0, bi = 0, col = j0;
        if (a_idx) for (int64_t x = 0; x < j0; ++x) a_idx[x] = (uint64_t)x;
        for (int64_t o = nops - 1; o >= 0; --o, ++col) {
            if (ops[o] == 'D') { if (a_idx) a_idx[ai] = col; if (b_idx) b_idx[bi] = col; ++ai; ++bi; }
            else if (ops[o] == 'H') { if (a_idx) a_idx[ai] = col; ++ai; }
            else { if (b_idx) b_idx[bi] = col; ++bi; }
        }
        if (a_idx) for (; ai < n; ++ai, ++col) a_idx[ai] = col;
    }
    free(ops); free(trace); free(S0); free(S1); free(H0); free(H1); free(V);
    free(acode); free(bcode); free(lut);
    return 0;
}
