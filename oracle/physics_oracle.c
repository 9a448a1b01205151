/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See physics_oracle_impl.inc for the header.
 * Builds the float32 oracle (*_f32) and its float64 shadow (*_f64) of the PhysTwin
 * spring-mass stepper from one restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define R2S_REAL float
#define R2S_SUFFIX _f32
#define R2S_SQRT sqrtf
#define R2S_EXP expf
#define R2S_ATAN2 atan2f
#include "physics_oracle_impl.inc"
#undef R2S_REAL
#undef R2S_SUFFIX
#undef R2S_SQRT
#undef R2S_EXP
#undef R2S_ATAN2

#define R2S_REAL double
#define R2S_SUFFIX _f64
#define R2S_SQRT sqrt
#define R2S_EXP exp
#define R2S_ATAN2 atan2
#include "physics_oracle_impl.inc"

/* Independent environments stepped side by side (the only parallelism the reference has:
 * one process per episode, experiments/eval_policy_parallel.py:266-280).  Arithmetic per
 * environment is exactly r2s_oracle_phys_step_f32. */
void r2s_oracle_phys_step_batch_f32(const phys_t_f32 *envs, float **x, float **v, int n_env, int first_substep,
                                    int n_run)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < n_env; ++e) r2s_oracle_phys_step_f32(&envs[e], x[e], v[e], first_substep, n_run);
}

/* bench.py's cpu_baseline driver: environments x particle chunks.  `n_env` outer threads (one per environment), each with a team of
 * `threads_per_env` that splits every per-particle loop of a substep into static chunks: eval_springs as a gather over the
 * particle's incident springs (no float atomics, same sums as the scatter), update_vel fused behind it, object_collision,
 * mesh_collision (per-face force sums: atomic adds, like the reference's), integrate_ground — the same functions the sequential
 * stepper above calls, over ranges.  Built twice from this file: into libr2s_oracle.so with the checker's strict flags (where
 * tests compare it with r2s_oracle_phys_step_f32: positions bit-equal) and into libr2s_cpu_baseline.so with -O3 -march / FMA
 * contraction (what bench.py times).  Returns the number of threads it ran (n_env x threads_per_env actually granted). */
int r2s_oracle_phys_step_batch_par_f32(const phys_t_f32 *envs, float **x, float **v, int n_env, int first_substep, int n_run,
                                       int threads_per_env)
{
    int granted = 0;
    omp_set_max_active_levels(2);
    omp_set_nested(1); /* libgomp of gcc 11 still gates inner teams on the (deprecated) nest-var */
#pragma omp parallel for num_threads(n_env) schedule(static, 1) reduction(+ : granted)
    for (int e = 0; e < n_env; ++e) {
        const phys_t_f32 *P = &envs[e];
        const int N = P->N, S = P->S;
        int *off = (int *)calloc((size_t)N + 2, sizeof(int));
        inc_rec_f32 *rec = (inc_rec_f32 *)malloc(sizeof(inc_rec_f32) * 2 * (size_t)(S > 0 ? S : 1));
        for (int s = 0; s < 2 * S; ++s) off[P->springs[s] + 2]++;
        for (int i = 0; i < N; ++i) off[i + 2] += off[i + 1];
        for (int s = 0; s < S; ++s) {            /* ascending s: a particle's records are in the order the scatter reaches it */
            const inc_rec_f32 r = {P->springs[2 * s], P->springs[2 * s + 1], P->rest[s], expf(P->log_Y[s])};
            rec[off[r.i1 + 1]++] = r;
            rec[off[r.i2 + 1]++] = r;          /* (a spring whose two ends are one particle contributes +F then -F: listed twice, like the scatter) */
        }
        float *f = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *vbc = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *vbg = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *xe = x[e], *ve = v[e];
        mesh_t_f32 m = {P->nV, P->nF, P->mesh_pts, P->faces, 0, {0, 0, 0}, {0, 0, 0}, 0, 0, 0};
        int team = 0;
#pragma omp parallel num_threads(threads_per_env)
        {
            const int nt = omp_get_num_threads(), t = omp_get_thread_num();
            const int lo = (int)((long long)N * t / nt), hi = (int)((long long)N * (t + 1) / nt);
            if (t == 0) team = nt;
            for (int s = first_substep; s < first_substep + n_run; ++s) {
                eval_springs_gather_f32(P, off, rec, xe, ve, f, lo, hi);
                update_vel_range_f32(P, ve, f, P->self_collision ? vbc : vbg, lo, hi);
                if (P->self_collision) {
#pragma omp barrier
                    object_collision_range_f32(P, xe, vbc, vbg, lo, hi);   /* reads the partners' x and pre-impulse velocity ... */
#pragma omp barrier
                }                                                           /* ... which the phases below overwrite in place */
                if (P->nF > 0) {
#pragma omp single
                    {
                        memcpy(P->mesh_pts, P->interp_pts + (size_t)s * P->n_dyn_pts * 3, sizeof(float) * 3 * (size_t)P->n_dyn_pts);
                        memset(P->collision_forces, 0, sizeof(float) * 3 * (size_t)P->nF);
                        mesh_box_f32(&m);
                    }   /* implicit barrier */
                    mesh_collision_range_f32(P, &m, s, xe, vbg, lo, hi, 1);
                }
                /* x[i] / v[i] of this thread's own particles only from here on, but the next substep's gather reads its neighbours' —
                 * and so does THIS substep's gather of a slower thread: integrate_ground overwrites x and v in place, so every thread's
                 * gather must be over first.  The self-collision block and the `omp single` of the mesh block end in barriers; a scene with
                 * neither (free springs over the ground) needs one of its own. */
                if (!P->self_collision && !(P->nF > 0)) {
#pragma omp barrier
                }
                integrate_ground_range_f32(P, xe, vbg, ve, lo, hi);
#pragma omp barrier
            }
        }
        granted += team;
        mesh_release_f32(&m);
        free(off); free(rec); free(f); free(vbc); free(vbg);
    }
    return granted;
}
