/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the Viterbi decode the reference obtains from pomegranate:
 *   call sites  scripts/STRique.py:434 (flankedRepeatHMM.count_repeats -> viterbi)
 *               scripts/STRique.py:493 (repeatModHMM.mod_repeats    -> viterbi)
 *   dependency  pomegranate v0.10.0 (requirements.txt:10-11), Cython, NOT in /root/reference and
 *               not installable here (no network).  This restates its published algorithm
 *               (HiddenMarkovModel._viterbi: log-space max-plus over emitting states, then the
 *               silent states in topological order, strict '>' so the first in-edge wins ties)
 *               as described in SURVEY.md Appendix A.4.
 *
 * PARITY STATUS: "parity unpinned" for log-probabilities and tie-breaks -- the reference holds
 * no golden vector for viterbi(); repeat counts are pinned by scripts/STRique_test.py
 * (n == i for every synthetic scenario) and by docs/installation/test.md:15-16 (+-2).
 * pomegranate's own in-edge order comes from networkx adjacency iteration over id()-hashed
 * objects, i.e. it is not reproducible even between two runs of the reference.
 *
 * Model arrays are the "baked" form (strique_amd/hmm.py: bake): emitting states first (sorted by
 * name), silent states after them in topological order, in-edges in CSR with ascending source.
 * Emissions: Normal  c - (x - mu)^2 * k   with c = -log(sigma * 2.50662827463), k = 1/(2 sigma^2)
 *            Uniform -log(hi - lo) inside [lo, hi], -inf outside.
 *            NaN observation: 0 under either (pomegranate's missing-value support since 0.9) [recalled].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define KIND_NORMAL 1
#define KIND_UNIFORM 2

/*
 * x[T] observations.  Outputs:
 *   logp            log-probability of the best path (-inf if none)
 *   emit_path[T]    emitting state of every observation (nullable)
 *   n_counted       sum of count_inc[state] over the whole path (nullable; count_inc nullable)
 * returns 0 ok, 1 no path, 2 out of memory / bad args.
 */
int strq_oracle_viterbi(int32_t n_states, int32_t silent_start, int32_t start, int32_t end,
                        const int32_t *in_ptr, const int32_t *in_src, const double *in_logp,
                        const int32_t *emis_kind, const double *emis_a, const double *emis_b,
                        const double *emis_c, const int32_t *count_inc,
                        const double *x, int64_t T,
                        double *logp, int32_t *emit_path, int64_t *n_counted)
{
    const int32_t m = n_states;
    if (m < 2 || T < 0) return 2;
    double *v = (double *)malloc(sizeof(double) * (size_t)(T + 1) * m);
    int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 1) * m);   /* predecessor state */
    if (!v || !bp) { free(v); free(bp); return 2; }
    const double NEGINF = -INFINITY;
    for (int32_t l = 0; l < m; ++l) { v[l] = NEGINF; bp[l] = -1; }
    v[start] = 0.0;
    /* silent states reachable before the first observation */
    for (int32_t l = silent_start; l < m; ++l) {
        if (l == start) continue;
        double best = NEGINF; int32_t arg = -1;
        for (int32_t e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
            const int32_t k = in_src[e];
            if (k < silent_start || k >= l) continue;
            const double c = v[k] + in_logp[e];
            if (c > best) { best = c; arg = k; }
        }
        v[l] = best; bp[l] = arg;
    }
    for (int64_t i = 0; i < T; ++i) {
        const double *vp = v + (size_t)i * m;
        double *vn = v + (size_t)(i + 1) * m;
        int32_t *bn = bp + (size_t)(i + 1) * m;
        const double xi = x[i];
        for (int32_t l = 0; l < silent_start; ++l) {
            double best = NEGINF; int32_t arg = -1;
            for (int32_t e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {
                const double c = vp[in_src[e]] + in_logp[e];
                if (c > best) { best = c; arg = in_src[e]; }
            }
            double em;
            if (xi != xi) {
                /* missing observation: pomegranate 0.10 distributions return log-probability 0 for NaN
                 * (NormalDistribution / UniformDistribution._log_probability, `if isnan(X[i]): log_probability[i] = 0.`) [recalled] */
                em = 0.0;
            } else if (emis_kind[l] == KIND_NORMAL) {
                const double d = xi - emis_a[l];
                em = emis_c[l] - (d * d) * emis_b[l];
            } else {
                em = (xi >= emis_a[l] && xi <= emis_b[l]) ? emis_c[l] : NEGINF;
            }
            vn[l] = best + em; bn[l] = arg;
        }
        for (int32_t l = silent_start; l < m; ++l) {
            double best = NEGINF; int32_t arg = -1;
            for (int32_t e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {     /* emitting predecessors, this step */
                const int32_t k = in_src[e];
                if (k >= silent_start) continue;
                const double c = vn[k] + in_logp[e];
                if (c > best) { best = c; arg = k; }
            }
            for (int32_t e = in_ptr[l]; e < in_ptr[l + 1]; ++e) {     /* earlier silent predecessors */
                const int32_t k = in_src[e];
                if (k < silent_start || k >= l) continue;
                const double c = vn[k] + in_logp[e];
                if (c > best) { best = c; arg = k; }
            }
            vn[l] = best; bn[l] = arg;
        }
    }
    const double lp = v[(size_t)T * m + end];
    *logp = lp;
    int rc = 0;
    if (!(lp > NEGINF)) rc = 1;
    else {
        int64_t i = T; int32_t l = end; int64_t cnt = 0;
        while (!(i == 0 && l == start)) {
            if (count_inc) cnt += count_inc[l];
            const int32_t prev = bp[(size_t)i * m + l];
            if (prev < 0) { rc = 1; break; }
            if (l < silent_start) { if (emit_path) emit_path[i - 1] = l; --i; }
            l = prev;
        }
        if (n_counted) *n_counted = cnt;
    }
    free(v); free(bp);
    return rc;
}
