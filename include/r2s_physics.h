/*
 * r2s_physics.h — C ABI of the MI355X (gfx950) PhysTwin spring-mass soft-body stepper.
 *
 * Drop-in boundary for the reference operator `SpringMassSystemWarp`
 *   sim/physics/spring_mass_warp.py:477-995   (14 NVIDIA-Warp kernels + CUDA-graph capture)
 * as driven by its only caller
 *   sim/physics/phystwin.py:336-357 (constructor), :362-521 (per-env-step protocol).
 * The reference has no FFI for this path (Warp JIT-compiles Python to CUDA); these entry points
 * are what a ctypes/cffi binding of that operator needs, one per method the caller uses.
 *
 * One handle holds a BATCH of `n_env` independent environments that share one PhysTwin
 * (same springs / stiffness / masses / meshes topology; own particle state, own mesh motion).
 * n_env = 1 reproduces the reference object exactly.
 *
 * Conventions
 *   - "host" pointers are read during the call and may be freed afterwards;
 *     "device" pointers are HIP device memory; work on them is enqueued on `stream`;
 *   - every call returns R2S_OK (0) or a negative R2S_ERR_* code (see r2s_raster.h);
 *     no exceptions cross the ABI; a handle is single-owner and not thread-safe;
 *   - all arithmetic is float32 like Warp's default; particle state is exposed in the
 *     reference's layout (vec3 array = [n_env, N, 3] float32).
 */
#ifndef R2S_PHYSICS_H
#define R2S_PHYSICS_H

#include <stddef.h>
#include <stdint.h>
#include "r2s_raster.h" /* error codes, r2s_stream_t */

#ifdef __cplusplus
extern "C" {
#endif

/* Scalar parameters (phystwin_cfg fields read at spring_mass_warp.py:501-512, 591-618 and the
 * six collide_* arrays set through set_collide*, :955-995). */
typedef struct R2SPhysParams {
    float dt;                /* cfg.dt (5e-5)                       */
    float dashpot_damping;   /* cfg.dashpot_damping (100)           */
    float drag_damping;      /* cfg.drag_damping (3)                */
    float spring_Y_min;      /* cfg.spring_Y_min (0)                */
    float spring_Y_max;      /* cfg.spring_Y_max (1e5)              */
    float collision_dist;    /* cfg.collision_dist (0.005)          */
    float collide_elas, collide_fric;           /* ground + static meshes */
    float collide_eef_elas, collide_eef_fric;   /* dynamic (robot) meshes */
    float collide_self_elas, collide_self_fric; /* particle-particle      */
    int32_t reverse_z;       /* cfg.reverse_z                       */
    int32_t self_collision;  /* cfg.self_collision                  */
    int32_t use_pusher;      /* ctor arg use_pusher                 */
    int32_t num_substeps;    /* cfg.num_substeps (667): length of the captured step graph */
} R2SPhysParams;

/* Constructor arguments (spring_mass_warp.py:478-500).  All pointers are HOST memory. */
typedef struct R2SPhysDesc {
    R2SPhysParams params;
    int32_t n_env;             /* batch of independent environments sharing this PhysTwin       */
    int32_t num_object_points; /* N                                                              */
    int32_t num_springs;       /* S                                                              */
    const float* init_vertices;    /* [n_env, N, 3]                                              */
    const float* init_velocities;  /* [n_env, N, 3] or NULL (zeros)                              */
    const int32_t* init_springs;   /* [S, 2]                                                     */
    const float* init_rest_lengths;/* [S]                                                        */
    const float* init_spring_Y;    /* [S]  LOG stiffness (phystwin.py:344)                       */
    const float* init_masses;      /* [N]                                                        */
    const int32_t* init_collision_mask; /* [N] or NULL -> arange(N) (spring_mass_warp.py:529-533) */
    /* Combined collision mesh (spring_mass_warp.py:626-695): dynamic meshes first (mesh_map 0,1,..),
     * then static meshes (mesh_map -1,-2,..).  n_meshes == 0 means "no meshes". */
    int32_t n_dynamic_meshes, n_static_meshes;
    const int32_t* mesh_num_vertices; /* [n_dynamic + n_static]                                  */
    const int32_t* mesh_num_faces;    /* [n_dynamic + n_static]                                  */
    const float* mesh_vertices;       /* concatenated [sum nv, 3], identical for every env at t=0 */
    const int32_t* mesh_triangles;    /* concatenated [sum nf, 3], indices local to each mesh    */
    int32_t collision_capacity;       /* candidate slots per particle; 0 -> 500 (:545)           */
} R2SPhysDesc;

typedef struct R2SPhys R2SPhys; /* opaque */

/* SpringMassSystemWarp.__init__ (:478-726): uploads topology and state, builds the combined mesh,
 * mesh_map / face_map, the resting-pair set (create_resting_case, :729-740) when self_collision, and
 * captures the num_substeps step graph (the reference's wp.ScopedCapture, :723-726). */
int r2s_phys_create(const R2SPhysDesc* desc, R2SPhys** out, r2s_stream_t stream);
void r2s_phys_destroy(R2SPhys* h);

/* set_init_state (:742-767).  x, v: device [n_env, N, 3]; v may be NULL (keep). */
int r2s_phys_set_state(R2SPhys* h, const float* x, const float* v, r2s_stream_t stream);
/* wp.to_torch(wp_state.wp_x / wp_v) (phystwin.py:523-531).  Either pointer may be NULL. */
int r2s_phys_get_state(R2SPhys* h, float* x, float* v, r2s_stream_t stream);

/* create_resting_case (:729-740): rebuild the resting-pair set from the CURRENT positions. */
int r2s_phys_create_resting_case(R2SPhys* h, r2s_stream_t stream);
/* update_collision_graph (:806-821): hash-grid rebuild + candidate lists, once per env step. */
int r2s_phys_update_collision_graph(R2SPhys* h, r2s_stream_t stream);

/* set_mesh_interactive (:769-804).  Device pointers, per environment:
 *   interp_points [n_env, num_substeps, n_dynamic_points, 3], interp_center [n_env, num_substeps, 3],
 *   dynamic_velocity [n_env, n_dyn_vel, 3] (n_dyn_vel = 2 gripper fingers, 1 pusher),
 *   dynamic_omega [n_env, 1, 3]. */
int r2s_phys_set_mesh_interactive(R2SPhys* h, const float* interp_points, const float* interp_center,
                                  const float* dynamic_velocity, const float* dynamic_omega,
                                  r2s_stream_t stream);

/* On-device gripper / pusher kinematics + grasp state machine: the caller side of the stepper,
 * SpringMassDynamicsModule.step (sim/physics/phystwin.py:362-513), for all environments at once.  Replaces the
 * host-built [num_substeps, n_dynamic_points, 3] tensor of set_mesh_interactive (~100 MB per environment step for
 * the pusher) and the blocking collision_forces.numpy() read of the grasp test (:383-389).
 *
 * r2s_phys_set_eef_table: the knots of eef_pts_func = scipy interp1d(arange(n_knots) / (n_knots - 1), eef_pts_list)
 *   (robot_pc_transformations.py:190, :225; 101 knots) as HOST float64 [n_knots, n_dynamic_points, 3], init_eef_xyz
 *   (HOST [3], gs_renderer.py:513-517) and cfg.physics.grasp_force_threshold.  Resets the per-environment state
 *   (current_openness = None, grasped = False).
 * r2s_phys_set_eef_motion: DEVICE pointers eef_xyz [n_env,3], eef_vel [n_env,3], eef_rot [n_env,3,3], eef_rot_vel [n_env,3],
 *   gripper_openness [n_env] (first gripper of each environment; ignored for the pusher).  Runs the openness / grasp
 *   state machine (:391-408) on the collision forces of the previous step, then fills the stepper's interpolated
 *   vertices, centres, dynamic velocities and omega exactly as set_mesh_interactive would have received them.  For a
 *   rigid mesh with more than 256 faces only the vertices the stepper reads are evaluated.
 * r2s_phys_eef_state: device pointers to current_openness (float64 [n_env]) and grasped (int32 [n_env]).
 * r2s_phys_mesh_motion: device pointers to the stepper's current motion inputs (interp_points [n_env, num_substeps,
 *   n_dynamic_points, 3], interp_center [n_env, num_substeps, 3], dynamic_velocity [n_env, 2, 3], dynamic_omega [n_env, 3]). */
int r2s_phys_set_eef_table(R2SPhys* h, int32_t n_knots, const double* eef_pts, const float* init_eef_xyz,
                           float grasp_force_threshold, r2s_stream_t stream);
int r2s_phys_set_eef_motion(R2SPhys* h, const float* eef_xyz, const float* eef_vel, const float* eef_rot,
                            const float* eef_rot_vel, const float* gripper_openness, r2s_stream_t stream);
int r2s_phys_eef_state(R2SPhys* h, double** current_openness, int32_t** grasped);
/* Episode reset inside a batch (BaseEnv.reset, env.py:30-51 -> PhysTwinDynamics.reset, phystwin.py:39-102, builds a NEW
 * SpringMassDynamicsModule per reset): the environments whose entry of the DEVICE int32 [n_env] mask is non-zero (NULL: all) get
 * what a new module starts from — current_openness = None, grasped = False (phystwin.py:358-360), collision_forces zero — while
 * the others keep running (episodes are independent, eval_policy_parallel.py:266-280).  The particle state of those
 * environments is the caller's to set (r2s_phys_set_state); r2s_phys_update_collision_graph rebuilds the candidate lists. */
int r2s_phys_reset_envs(R2SPhys* h, const int32_t* env_mask, r2s_stream_t stream);
/* The particle state of an episode reset: like r2s_phys_set_state for the environments with a non-zero mask entry (DEVICE int32
 * [n_env], required), the others are not touched — and neither is the sticky fault word of r2s_phys_step: it is per handle and may
 * have been raised by an environment that keeps running (a full r2s_phys_set_state is what clears it). */
int r2s_phys_set_state_envs(R2SPhys* h, const float* x, const float* v, const int32_t* env_mask, r2s_stream_t stream);
/* create_resting_case (:729-740) for the masked environments only: a reset episode that starts from ANOTHER pose (the reference
 * builds a new stepper per reset, whose resting-pair set comes from the new initial positions — and depends on them through the
 * hash-grid cells) while the other environments keep theirs. */
int r2s_phys_create_resting_case_envs(R2SPhys* h, const int32_t* env_mask, r2s_stream_t stream);
/* Episode reset into another scene pose (gs_renderer.py:353-390: load_scaniverse re-poses every mesh with a grid_randomization entry by
 * the episode index before PhysTwinDynamics is rebuilt from it): the vertices of the STATIC collision meshes of the environments whose
 * env_mask[e] != 0 (null = all), DEVICE float [n_env, n_static_vertices, 3] in the order of R2SPhysDesc::mesh_vertices behind the dynamic
 * ones — ALL static meshes, n_static_vertices = their vertex count (checked: R2S_ERR_INVALID otherwise); their boxes are rebuilt.
 * R2S_ERR_INVALID for a static mesh with more than 256 faces (its triangle table is built at create). */
int r2s_phys_set_static_mesh_points(R2SPhys* h, const float* pts, int32_t n_static_vertices, const int32_t* env_mask, r2s_stream_t stream);
int r2s_phys_mesh_motion(R2SPhys* h, float** interp_points, float** interp_center, float** dynamic_velocity,
                         float** dynamic_omega);

/* step (:823-943) / wp.capture_launch(graph) (phystwin.py:515-519).  n_substeps <= 0 or
 * == params.num_substeps replays the captured graph; any other count runs substeps
 * [first_substep, first_substep + n_substeps) eagerly (they index the interpolated mesh motion).
 * Errors (R2S_ERR_INVALID + r2s_last_error): two conditions the kernels detect are reported by the first r2s_phys_step AFTER
 * the step in which they occurred has finished on the device (a word copied to pinned memory, read without waiting) — (1) a
 * self-collision impulse changed a particle's velocity by more than 40 m/s in one substep, so the "no mesh within reach"
 * decision taken before the impulse (test widened by 2 mm) may have skipped a mesh response the reference applies; (2) a
 * workgroup of the resident small-batch launch waited for its neighbour beyond the poll limit (see r2s_phys_set_resident), or a
 * block of a large batch for a particle the head of its launch finishes (see r2s_phys_set_pf).
 * The word is sticky — every later r2s_phys_step fails — until r2s_phys_set_state hands in a new state.
 * Which captured flavour a step runs (finishing code in the graph or mesh queries in place; r2s_phys_last_flavour) follows from
 * counters of the step TWO before it — copied to pinned memory behind that step and WAITED for here if they have not landed (they
 * have, in any loop that reads its observations; an open loop is held to two steps of run-ahead) — never from whatever copy happens
 * to have arrived: two runs of the same calls end in the same bits.  r2s_phys_set_state starts a new history (two steps of the default
 * flavour: queries in place — always correct, slower in contact). */
int r2s_phys_step(R2SPhys* h, int n_substeps, int first_substep, r2s_stream_t stream);

/* collision_forces (:690-695): device pointer to [n_env, n_faces, 3]; holds the LAST substep's
 * per-face force, as in the reference which zeroes it every substep (:900). */
int r2s_phys_collision_forces(R2SPhys* h, float** device_ptr, int32_t* n_faces);
/* mesh_map / face_map (:677-689), copied to host arrays of n_faces ints (either may be NULL). */
int r2s_phys_mesh_maps(R2SPhys* h, int32_t* mesh_map, int32_t* face_map);
/* Candidate lists of update_potential_collision (:544-552): device pointers to
 * collision_number [n_env, N] and collision_indices [n_env, N, capacity]. */
int r2s_phys_collision_lists(R2SPhys* h, int32_t** number, int32_t** indices, int32_t* capacity);
/* Overflow report: max candidates any particle wanted at the last update (host sync). The reference
 * writes past its 500-wide row unchecked (:226); here extra candidates are dropped and reported. */
int r2s_phys_collision_max_count(R2SPhys* h, int32_t* max_count, r2s_stream_t stream);

/* set_spring_Y (:946-953, host [S] log stiffness) and set_collide* (:955-995). */
int r2s_phys_set_spring_Y(R2SPhys* h, const float* log_Y, r2s_stream_t stream);
int r2s_phys_set_params(R2SPhys* h, const R2SPhysParams* params, r2s_stream_t stream);

/* Candidate lists written by the caller: collision_number / collision_indices are plain arrays of the reference object
 * (:544-552) that update_potential_collision fills; this is the write side.  HOST arrays in the caller's particle
 * indexing, number [n_env, N] and indices [n_env, N, capacity].  Used by the parity tests to replay lists recorded from
 * the reference's own kernels; the next r2s_phys_update_collision_graph overwrites them. */
int r2s_phys_set_collision_lists(R2SPhys* h, const int32_t* number, const int32_t* indices, r2s_stream_t stream);

/* Contact report of the last step (bench.py's phase accounting): particles that currently have self-collision candidates
 * (host view of the last update_collision_graph) and a device pointer to int32 [n_env], the particles of each environment
 * that reacted to a collision mesh (mesh_collision's err < 0 branch, :343) in the LAST substep. */
int r2s_phys_contact_stats(R2SPhys* h, int32_t* particles_with_candidates, int32_t** mesh_hits_dev);
/* The same three counts {particles with candidates, mesh hits of the last substep, grasped environments} written to a
 * DEVICE int32[3] by a tiny kernel on `stream` — a per-step log without a host synchronisation. */
int r2s_phys_log_contacts(R2SPhys* h, int32_t* out3_dev, r2s_stream_t stream);
/* Diagnostics: out[k], k < num_substeps = particles whose mesh query the fused kernel deferred to the finishing kernel in
 * substep k of the last env step; out[num_substeps] != 0 if any particle was near a collision mesh.  HOST int32
 * [num_substeps + 1]; synchronises `stream`. */
int r2s_phys_deferred_counts(R2SPhys* h, int32_t* out, r2s_stream_t stream);
/* Diagnostics: how many particles with self-collision candidates were ALSO handed to the finishing kernel's mesh list
 * (tagged entries: object_collision impulses, :230-268, then mesh_collision, :295-421, by one workgroup) at least once during
 * the last r2s_phys_step.  HOST int32[1]; synchronises `stream`. */
int r2s_phys_tagged_count(R2SPhys* h, int32_t* out, r2s_stream_t stream);
/* Which captured flavour the last r2s_phys_step ran: out[0] bit 0 self-collision variant, bit 1 the handle's per-substep kernel is
 * k_steps_resident with one substep (small-batch layout), out[1] mesh template (0 none,
 * 1 every mesh small: the fused kernel answers the rare query itself unless out[2], 2 a large mesh is present: the fused
 * kernel only lists), out[2] finishing code in the graph (0/1; always with a large mesh; 2 = the env step ran as ONE resident
 * launch, see r2s_phys_set_resident; 3 = like 1 with the finishers of substep k at the head of substep k + 1's launch, see
 * r2s_phys_set_pf), out[3] kernel chains. */
int r2s_phys_last_flavour(R2SPhys* h, int32_t* out);
/* Flavour selection as a PURE function (round 6): everything r2s_phys_step decides before it launches — which captured graph, which
 * kernels, how many chains, how many query-server workgroups — from (a) the counters the env step TWO before this one left, (b) what the
 * handle can do, fixed at create, (c) the switches.  Host only: no handle, no device, no state; r2s_phys_step fills the input from its
 * handle and calls the same function, tests/test_flavour_matrix.py enumerates it without a GPU.  The reference has one flavour
 * (spring_mass_warp.py:823-943 launches the same nine kernels every substep); the flavours here run the same arithmetic with
 * different work splits, and which pairs are bit-identical is stated in `R2SFlavourOut::sum_class` (same class = same bits). */
typedef struct R2SFlavourIn {
    /* (a) counters of env step t - 2 (have_counters = 0: a new history — they read as zero and the default flavour runs) */
    int32_t have_counters;
    int32_t near_mesh;          /* a particle was within margin + 3 cm of a collision mesh's box                         */
    int32_t query_needed;       /* a particle was inside a mesh's reach (a query had to be answered)                     */
    int32_t servers_ran_out;    /* a resident launch had more particles in contact than server units, or was running low (> 30 % claimed) */
    int32_t srv_exhausted;      /* the handle's sticky copy of that (cleared once no query is needed any more)           */
    int32_t n_candidates;       /* particles with self-collision candidates after the last update_collision_graph        */
    int32_t n_substeps;         /* substeps of this call                                                                 */
    int32_t full_step;          /* the call replays the captured env step (all substeps from 0)                          */
    /* (b) capabilities */
    int32_t n_faces;            /* faces of the combined collision mesh (0: no meshes)                                   */
    int32_t any_large;          /* some mesh has more than 256 faces                                                     */
    int32_t block;              /* particles per workgroup of the layout: 256, 128 or 64                                 */
    int32_t split_ok;           /* 64-particle layout whose slices fit k_steps_resident's registers                      */
    int32_t resident_ok;        /* the env step can run as ONE resident launch                                           */
    int32_t srv_ok;             /* a resident launch may carry query servers (small scene, CUs to spare)                 */
    int32_t pf_ok;              /* the contact flavours can run with the finishers at the head of the next launch        */
    int32_t has_vx;             /* the resident self-collision flavour's record array exists                             */
    int32_t self_collision;     /* cfg.self_collision                                                                    */
    int32_t n_blocks, n_env;    /* particle blocks per environment, environments                                         */
    int32_t n_cu, srv_wg_cap;   /* CUs a resident launch may fill, cap on its server workgroups                          */
    /* (c) switches (environment variables at create, r2s_phys_set_* later) */
    int32_t resident_pref, res_self, res_self_srv, pf_pref, force_defer /* -1 auto */, chains_override /* 0 auto */, srv_own, srv_quad /* -1 auto */;
} R2SFlavourIn;
typedef struct R2SFlavourOut {
    int32_t variant;            /* 1: the SELF templates (some particle has candidates)                                  */
    int32_t mesh;               /* mesh template: 0 none, 1 small meshes, 2 a large mesh is present                      */
    int32_t mesh_defer;         /* needy particles are listed and finished by the finishing code (else queried in place) */
    int32_t resident;           /* the env step runs as one k_steps_resident launch                                      */
    int32_t self_srv;           /* ... the self-collision flavour of it, with (answering) query servers                  */
    int32_t pf;                 /* finishing code of substep k at the head of substep k + 1's launch (k_substep_pf)      */
    int32_t contact_finish;     /* the flavour carries k_contact_finish (pf: only behind the last substep)               */
    int32_t chains;             /* concurrent kernel chains                                                              */
    int32_t n_srv, srv_quad, srv_own; /* server workgroups of a resident launch, units of four wavefronts, units own their particle */
    int32_t srv_exhausted;      /* the sticky word after this decision                                                   */
    int32_t graph_slot;         /* index of the captured graph (without the state buffer's parity bit)                   */
    int32_t sum_class;          /* flavours of one scene with the same class end in the same bits                        */
    char kernel[192];           /* what runs, as r2s_phys_last_flavour's callers print it                                */
} R2SFlavourOut;
int r2s_phys_debug_pick_flavour(const R2SFlavourIn* in, R2SFlavourOut* out);
/* The input r2s_phys_step would build for the handle's NEXT full env step (counters as they are known to the host now). */
int r2s_phys_debug_flavour_input(R2SPhys* h, R2SFlavourIn* out);

/* What the last r2s_phys_step decided (the full record; r2s_phys_last_flavour is its four-word summary). */
int r2s_phys_last_flavour_ex(R2SPhys* h, R2SFlavourOut* out);
/* The sticky fault word of r2s_phys_step, read NOW: waits for everything enqueued on `stream`, returns R2S_ERR_INVALID with the step's
 * message in r2s_last_error() if a fault is pending (R2S_OK otherwise).  r2s_phys_step reports a fault with a lag of up to two env steps
 * and a full r2s_phys_set_state clears it: a caller that ends episodes and resets every environment at once (evaluate.run_episodes)
 * asks here before it records their outcome. */
int r2s_phys_check_fault(R2SPhys* h, r2s_stream_t stream);

/* Tuning (not part of the reference surface): chains > 0 overrides the number of concurrent kernel chains of the captured
 * env step (0 = default), mesh_defer 0/1 forces the deferred large-mesh-query flavour (-1 = automatic).  The environment
 * variables R2S_CHAINS / R2S_MESH_DEFER / R2S_LAYOUT / R2S_HALO_CAP are read ONCE, by r2s_phys_create. */
int r2s_phys_set_tuning(R2SPhys* h, int chains, int mesh_defer);
/* Small batches (at most 256 work items of 64 particles: one environment of the reference's own evaluation loop,
 * eval_policy.py:180-240 -> phystwin.py:104-147 -> spring_mass_warp.py:723-726 `for i in range(num_substeps): step()`, up to a
 * few) are laid out in 64-particle blocks, and the env step's free flavour (no self-collision candidates, nothing within reach of
 * a collision mesh) runs as ONE launch that keeps the particles on the chip for all substeps (k_steps_resident: neighbouring
 * blocks hand their halo over through tagged write-through records instead of kernel boundaries).  on = 0 runs every flavour
 * with the per-substep kernels of the same layout (tests compare the two; R2S_RESIDENT=0 at create also keeps the large-batch
 * layout).  Results of the two agree to the last bits (different summation order), not bit for bit. */
int r2s_phys_set_resident(R2SPhys* h, int on);
/* Large batches (the 256-particle layout), contact flavours: what the fused substep kernel cannot finish in its own thread — deferred
 * mesh queries, particles with self-collision candidates (spring_mass_warp.py:845-943 runs object_collision and mesh_collision as
 * launches of their own every substep) — is finished by the first workgroups of the NEXT substep's launch (k_substep_pf), and only
 * the blocks that hold such a particle wait for it: one launch per substep instead of two.  on = 0 keeps the two-launch form
 * (fused kernel + k_contact_finish per substep; also R2S_PF=0 at create); states agree bit for bit (tests/test_pf_gpu.py). */
int r2s_phys_set_pf(R2SPhys* h, int on);
/* The device's pooled side stream k (1 .. 7) — the streams the library launches the kernel chains 1.. of an env step on; created on
 * first use, shared by every handle of the current device, never destroyed.  For callers that want work of their own next to the
 * launch stream BETWEEN env steps (the rollout runs update_collision_graph there while the launch stream renders): a stream of the
 * caller's own would be one more hardware queue, and with more streams than queues the chains of the next step share one and
 * serialise (measured: 10.2 -> 12.2 us per batched substep on the 32-environment T-block scene). */
int r2s_phys_side_stream(int32_t k, r2s_stream_t* out);

/* Layout report (DESIGN.md / bench.py): out[0] particle blocks, [1] largest halo (records), [2] ELL slots incl.
 * padding, [3] real neighbour slots (= 2 * active springs), [4] slots served by the global fallback instead of LDS,
 * [5] LDS bytes per workgroup, [6] concurrent kernel chains of the captured env step, [7] work items per XCD chunk. */
int r2s_phys_layout_stats(R2SPhys* h, int64_t* out);

/* Measurement hook for bench.py: HIP-event time (ms) of the last r2s_phys_step on its stream and
 * the number of substep kernels it launched (events are recorded only when enabled). */
void r2s_phys_set_timing(R2SPhys* h, int enable);
int r2s_phys_last_step_ms(R2SPhys* h, float* ms, int32_t* kernels);

#ifdef __cplusplus
}
#endif
#endif /* R2S_PHYSICS_H */
