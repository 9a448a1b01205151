/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See physics_oracle_impl.inc for the header.
 * Builds the float32 oracle (*_f32) and its float64 shadow (*_f64) of the PhysTwin
 * spring-mass stepper from one restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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
