// physics_aux.h — part of physics.hip's ONE translation unit (included there, inside its anonymous namespace, in this order: physics_mesh_query.h,
// physics_substep.h, physics_resident.h, physics_finish.h, physics_aux.h); not a stand-alone header.  Round 5 split the 4 800-line file by section;
// the token stream the compiler sees is unchanged.
// Here: auxiliary kernels: state pack / unpack, mesh boxes and rigid transforms, on-device gripper / pusher kinematics and grasp state machine, the hash grid and the candidate rebuild.

// ---- state pack / unpack: caller order [env][user index][3]  <->  internal [env][Morton index]{x,v} -------
__global__ void k_pack(int N, int E, const int* __restrict__ inv, const float* __restrict__ x, const float* __restrict__ v, const StateM xv,
                       const int* __restrict__ env_mask)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N || (env_mask && env_mask[e] == 0)) return;
    const size_t src = ((size_t)e * N + u) * 3, dst = (size_t)e * N + inv[u];
    float* f = (float*)xv.p; // x and v may be set separately: plane 1 holds one component of each
    if (x) { xv.p[st_at(xv.n, dst, 0)] = (v2f){x[src], x[src + 1]}; f[2 * st_at(xv.n, dst, 1)] = x[src + 2]; }
    if (v) { xv.p[st_at(xv.n, dst, 2)] = (v2f){v[src], v[src + 1]}; f[2 * st_at(xv.n, dst, 1) + 1] = v[src + 2]; }
}
__global__ void k_unpack(int N, int E, const int* __restrict__ inv, const StateC xv, float* __restrict__ x, float* __restrict__ v)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N) return;
    const size_t dst = ((size_t)e * N + u) * 3, src = (size_t)e * N + inv[u];
    const v2f b = xv.p[st_at(xv.n, src, 1)];
    if (x) { const v2f a = xv.p[st_at(xv.n, src, 0)]; x[dst] = a.x; x[dst + 1] = a.y; x[dst + 2] = b.x; }
    if (v) { const v2f c = xv.p[st_at(xv.n, src, 2)]; v[dst] = c.x; v[dst + 1] = c.y; v[dst + 2] = b.y; }
}
// candidate lists back to the caller's indexing (debug / parity taps)
__global__ void k_lists_to_user(int N, int E, int cap, const int* __restrict__ perm, const int* __restrict__ num, const int* __restrict__ idx,
                                int* __restrict__ num_u, int* __restrict__ idx_u)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t src = (size_t)e * N + i, dst = (size_t)e * N + perm[i];
    const int c = num[src];
    num_u[dst] = c;
    for (int k = 0; k < c; ++k) idx_u[dst * cap + k] = perm[idx[src * cap + k]];
}

// ---- mesh AABBs per (env, substep, dynamic mesh) and per (env, static mesh) ----------------------------
__global__ void k_mesh_aabb_dyn(int E, int n_sub, int n_dyn_mesh, int n_dyn_pts, const int* __restrict__ mesh_vert_off,
                                const int* __restrict__ mesh_kind, const float* __restrict__ interp, float* __restrict__ aabb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_sub * n_dyn_mesh) return;
    const int m = t % n_dyn_mesh;
    if (mesh_kind[m] & 1) return; // large rigid meshes: box from the transformed rest box (k_mesh_xf)
    const size_t es = t / n_dyn_mesh;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int vtx = mesh_vert_off[m]; vtx < mesh_vert_off[m + 1]; ++vtx) {
        const float* q = interp + (es * n_dyn_pts + vtx) * 3;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
    }
    float* o = aabb + (size_t)t * 6;
    for (int k = 0; k < 3; ++k) { o[k] = lo[k]; o[3 + k] = hi[k]; }
}
__global__ void k_mesh_aabb_static(int E, int n_static, int n_dyn_mesh, int nV, const int* __restrict__ mesh_vert_off,
                                   const float* __restrict__ pts, float* __restrict__ aabb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_static) return;
    const int m = t % n_static, e = t / n_static;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int vtx = mesh_vert_off[n_dyn_mesh + m]; vtx < mesh_vert_off[n_dyn_mesh + m + 1]; ++vtx) {
        const float* q = pts + ((size_t)e * nV + vtx) * 3;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
    }
    float* o = aabb + (size_t)t * 6;
    for (int k = 0; k < 3; ++k) { o[k] = lo[k]; o[3 + k] = hi[k]; }
}

// Rigid transform of every large dynamic mesh at every (env, substep), recovered from three reference vertices of the
// interpolated motion (rest frame = vertex positions at construction): orthonormal frames on both sides, R = Fcur Frest^T,
// t = c0 - R r0.  Also the mesh's world AABB (the 8 transformed corners of its rest box: a superset, which keeps the
// early-out conservative) and a rigidity check on a sample of vertices (max deviation -> rigid_err, float bits).
__global__ void k_mesh_xf(int E, int n_sub, int n_dyn_mesh, int n_dyn_pts, int n_xf, const int* __restrict__ xf_mesh,
                          const int* __restrict__ xf_ref, const int* __restrict__ mesh_vert_off, const float* __restrict__ rest,
                          const float* __restrict__ rest_box, const float* __restrict__ interp, float* __restrict__ xf,
                          float* __restrict__ aabb_dyn, unsigned* __restrict__ rigid_err)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_sub * n_xf) return;
    const int k = t % n_xf;
    const size_t es = t / n_xf;
    const int m = xf_mesh[k];
    const float* cur = interp + es * n_dyn_pts * 3;
    auto frame = [](f3 p0, f3 p1, f3 p2, f3& e1, f3& e2, f3& e3) {
        e1 = normalize0(p1 - p0);
        const f3 w = p2 - p0;
        e2 = normalize0(w - e1 * dot(w, e1));
        e3 = cross(e1, e2);
    };
    const int i0 = xf_ref[3 * k], i1 = xf_ref[3 * k + 1], i2 = xf_ref[3 * k + 2];
    f3 a1, a2, a3, b1, b2, b3;
    frame(ld3(rest, i0), ld3(rest, i1), ld3(rest, i2), a1, a2, a3);
    frame(ld3(cur, i0), ld3(cur, i1), ld3(cur, i2), b1, b2, b3);
    Xf X;
    // R = b1 a1^T + b2 a2^T + b3 a3^T
    X.r[0] = b1.x * a1.x + b2.x * a2.x + b3.x * a3.x; X.r[1] = b1.x * a1.y + b2.x * a2.y + b3.x * a3.y; X.r[2] = b1.x * a1.z + b2.x * a2.z + b3.x * a3.z;
    X.r[3] = b1.y * a1.x + b2.y * a2.x + b3.y * a3.x; X.r[4] = b1.y * a1.y + b2.y * a2.y + b3.y * a3.y; X.r[5] = b1.y * a1.z + b2.y * a2.z + b3.y * a3.z;
    X.r[6] = b1.z * a1.x + b2.z * a2.x + b3.z * a3.x; X.r[7] = b1.z * a1.y + b2.z * a2.y + b3.z * a3.y; X.r[8] = b1.z * a1.z + b2.z * a2.z + b3.z * a3.z;
    const f3 r0 = ld3(rest, i0), c0 = ld3(cur, i0);
    const f3 rr = xf_rotate(X, r0);
    X.t[0] = c0.x - rr.x; X.t[1] = c0.y - rr.y; X.t[2] = c0.z - rr.z;
    float* o = xf + (size_t)t * 12;
    for (int j = 0; j < 9; ++j) o[j] = X.r[j];
    for (int j = 0; j < 3; ++j) o[9 + j] = X.t[j];
    const float* rb = rest_box + (size_t)k * 6;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int c = 0; c < 8; ++c) {
        const f3 w = xf_apply(X, mk(rb[(c & 1) ? 3 : 0], rb[(c & 2) ? 4 : 1], rb[(c & 4) ? 5 : 2]));
        lo[0] = fminf(lo[0], w.x); hi[0] = fmaxf(hi[0], w.x); lo[1] = fminf(lo[1], w.y); hi[1] = fmaxf(hi[1], w.y);
        lo[2] = fminf(lo[2], w.z); hi[2] = fmaxf(hi[2], w.z);
    }
    float* bb = aabb_dyn + (es * n_dyn_mesh + m) * 6;
    for (int j = 0; j < 3; ++j) { bb[j] = lo[j] - 1e-6f; bb[3 + j] = hi[j] + 1e-6f; }
    const int v0 = mesh_vert_off[m], v1 = mesh_vert_off[m + 1];
    const int stride = max(1, (v1 - v0) / 48);
    float worst = 0.f;
    for (int v = v0; v < v1; v += stride) worst = fmaxf(worst, len(xf_apply(X, ld3(rest, v)) - ld3(cur, v)));
    atomicMax(rigid_err, __float_as_uint(worst));
}

// ---- on-device gripper / pusher kinematics + grasp state machine ----------------------------------------------
// What SpringMassDynamicsModule.step computes on the host before it calls set_mesh_interactive (phystwin.py:362-513),
// for every environment at once and without the D2H read of collision_forces.  float32 where the reference uses float32
// torch ops (same operation order, no FMA contraction), float64 for the host-side python / scipy part (openness state
// machine, interp1d of the finger vertices).
struct EefIn {
    const float* xyz;      // [E,3]   eef_xyz (first gripper)
    const float* vel;      // [E,3]   eef_vel
    const float* rot;      // [E,3,3] eef_rot
    const float* rot_vel;  // [E,3]   eef_rot_vel (axis-angle rate)
    const float* open;     // [E]     gripper_openness
};

// scipy.interpolate.interp1d(kind='linear') over x = arange(K) / (K-1.0), evaluated like scipy's _call_linear:
// hi = clip(searchsorted(x, x_new, 'left'), 1, K-1), slope = (y_hi - y_lo) / (x_hi - x_lo), y = slope * (x_new - x_lo) + y_lo.
__device__ __forceinline__ void eef_knot(double x_new, int K, int& lo, double& x_lo, double& inv_dx_num, double& x_hi)
{
    const double den = (double)(K - 1);
    int a = 0, b = K; // first index with x[i] >= x_new
    while (a < b) { const int m = (a + b) >> 1; if ((double)m / den < x_new) a = m + 1; else b = m; }
    const int hi = min(max(a, 1), K - 1);
    lo = hi - 1;
    x_lo = (double)lo / den; x_hi = (double)hi / den;
    inv_dx_num = x_hi - x_lo;
}

// One workgroup per environment: state machine (thread 0), then the per-vertex quantities that do not depend on the
// substep — relative_eef_pts at the substep-0 end (rel0), eef_pts_delta (delta) — and the finger closing velocities.
__global__ void __launch_bounds__(256) k_eef_prepare(int E, int M, int K, int use_pusher, const double* __restrict__ table, float ix, float iy,
                                                     float iz, float thr, int f_left, int f_right, int nF,
                                                     const float* __restrict__ coll_forces, EefIn in, double* __restrict__ cur_open,
                                                     int* __restrict__ grasped, int* __restrict__ has_state, float* __restrict__ rel0,
                                                     float* __restrict__ delta, float* __restrict__ dyn_vel, float* __restrict__ dyn_omega,
                                                     float two_dt_n)
{
    const int e = blockIdx.x, tid = threadIdx.x;
    __shared__ double s_open[2];
    __shared__ float s_red[2][3][256 / 64];
    if (tid == 0) {
        double now, before;
        if (use_pusher) {
            now = before = 1.0; cur_open[e] = 1.0; // phystwin.py:464, :474-477
        } else {
            double openness = (double)in.open[e]; // gripper_openness.item()
            double cur = has_state[e] ? cur_open[e] : openness; // :371-372
            int g = grasped[e];
            const float* F = coll_forces + (size_t)e * nF * 3;
            float n2[2];
            for (int side = 0; side < 2; ++side) { // :380-389: faces 18, 19, 1 of each finger, float32 sums and norm
                const float* f0 = F + (size_t)((side ? f_right : f_left) + 18) * 3;
                const float* f1 = F + (size_t)((side ? f_right : f_left) + 19) * 3;
                const float* f2 = F + (size_t)((side ? f_right : f_left) + 1) * 3;
                const float x = (f0[0] + f1[0]) + f2[0], y = (f0[1] + f1[1]) + f2[1], z = (f0[2] + f1[2]) + f2[2];
                n2[side] = sqrtf((x * x + y * y) + z * z);
            }
            before = cur;
            if (n2[0] < 100.f && n2[1] < 100.f) g = 0; // :393-394
            if (openness < cur) {                       // :395-405
                if (n2[0] > thr && n2[1] > thr) { openness = cur; g = 1; }
                else if (g) { cur = fmax(openness, cur - 0.05); openness = cur; }
                else cur = openness;
            } else cur = openness;
            cur_open[e] = cur; grasped[e] = g; has_state[e] = 1;
            now = fmin(fmax(openness, 0.0), 1.0); before = fmin(fmax(before, 0.0), 1.0); // np.clip, :411, :419
        }
        s_open[0] = now; s_open[1] = before;
    }
    __syncthreads();
    int lo_n, lo_b; double xl_n, dx_n, xh_n, xl_b, dx_b, xh_b;
    eef_knot(s_open[0], K, lo_n, xl_n, dx_n, xh_n);
    eef_knot(s_open[1], K, lo_b, xl_b, dx_b, xh_b);
    const float* R = in.rot + (size_t)e * 9;
    float accL[3] = {0.f, 0.f, 0.f}, accR[3] = {0.f, 0.f, 0.f};
    const int half = M / 2;
    for (int v = tid; v < M; v += 256) {
        float pn[3], pb[3];
        for (int c = 0; c < 3; ++c) {
            const double yl = table[((size_t)lo_n * M + v) * 3 + c], yh = table[((size_t)(lo_n + 1) * M + v) * 3 + c];
            pn[c] = (float)(((yh - yl) / dx_n) * (s_open[0] - xl_n) + yl);
            const double zl = table[((size_t)lo_b * M + v) * 3 + c], zh = table[((size_t)(lo_b + 1) * M + v) * 3 + c];
            pb[c] = (float)(((zh - zl) / dx_b) * (s_open[1] - xl_b) + zl);
        }
        float d[3] = {pn[0] - pb[0], -(pn[1] - pb[1]), -(pn[2] - pb[2])};              // :422-424 (flip y, z)
        float r[3] = {pb[0] - ix, -(pb[1] - iy), -(pb[2] - iz)};                          // :425-427
        float* o = rel0 + ((size_t)e * M + v) * 3; o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
        float* q = delta + ((size_t)e * M + v) * 3; q[0] = d[0]; q[1] = d[1]; q[2] = d[2];
        if (!use_pusher) { // closing velocity: (delta @ eef_rot[0]^T) / (2 dt n), :446-447
            const float c0 = ((d[0] * R[0] + d[1] * R[1]) + d[2] * R[2]) / two_dt_n;
            const float c1 = ((d[0] * R[3] + d[1] * R[4]) + d[2] * R[5]) / two_dt_n;
            const float c2 = ((d[0] * R[6] + d[1] * R[7]) + d[2] * R[8]) / two_dt_n;
            float* a = v < half ? accL : accR;
            a[0] += c0; a[1] += c1; a[2] += c2;
        }
    }
    // block sums of the two halves (the reference takes torch means; summation order differs in the last bits)
    for (int side = 0; side < 2; ++side)
        for (int c = 0; c < 3; ++c) {
            float x = side ? accR[c] : accL[c];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if ((tid & 63) == 0) s_red[side][c][tid >> 6] = x;
        }
    __syncthreads();
    if (tid < 3) {
        const float ev = in.vel[(size_t)e * 3 + tid] * 0.5f; // :443
        if (use_pusher) {
            dyn_vel[(size_t)e * 6 + tid] = ev; dyn_vel[(size_t)e * 6 + 3 + tid] = 0.f;
        } else {
            const float sl = (s_red[0][tid][0] + s_red[0][tid][1]) + (s_red[0][tid][2] + s_red[0][tid][3]);
            const float sr = (s_red[1][tid][0] + s_red[1][tid][1]) + (s_red[1][tid][2] + s_red[1][tid][3]);
            dyn_vel[(size_t)e * 6 + tid] = ev + sl / (float)max(half, 1);          // :448-454
            dyn_vel[(size_t)e * 6 + 3 + tid] = ev + sr / (float)max(M - half, 1);
        }
        dyn_omega[(size_t)e * 3 + tid] = -in.rot_vel[(size_t)e * 3 + tid] * 0.5f; // :457
    }
}

// kornia.geometry.conversions.axis_angle_to_rotation_matrix (third-party, not under the reference tree; restated from
// its published source): Rodrigues with w = aa / (theta + 1e-6) where theta^2 > 1e-6, first-order matrix otherwise.
__device__ __forceinline__ void eef_aa_to_matrix(float ax, float ay, float az, float* r)
{
    const float theta2 = (ax * ax + ay * ay) + az * az;
    if (theta2 > 1e-6f) {
        const float theta = sqrtf(theta2);
        const float wx = ax / (theta + 1e-6f), wy = ay / (theta + 1e-6f), wz = az / (theta + 1e-6f);
        const float c = cosf(theta), sn = sinf(theta), k = 1.0f - c;
        r[0] = c + wx * wx * k;        r[1] = wx * wy * k - wz * sn; r[2] = wy * sn + wx * wz * k;
        r[3] = wz * sn + wx * wy * k;  r[4] = c + wy * wy * k;       r[5] = -wx * sn + wy * wz * k;
        r[6] = -wy * sn + wx * wz * k; r[7] = wx * sn + wy * wz * k; r[8] = c + wz * wz * k;
    } else {
        r[0] = 1.f; r[1] = -az; r[2] = ay; r[3] = az; r[4] = 1.f; r[5] = -ax; r[6] = -ay; r[7] = ax; r[8] = 1.f;
    }
}

// interpolated_dynamic_points / interpolated_center for every (env, substep) and every vertex the stepper reads
// (all vertices of small meshes; for large rigid meshes only the three reference vertices and the rigidity sample).
__global__ void __launch_bounds__(256) k_eef_points(int E, int n_sub, int M, int n_need, const int* __restrict__ need, EefIn in,
                                                    const float* __restrict__ rel0, const float* __restrict__ delta, float dt, float dt_n,
                                                    float* __restrict__ interp, float* __restrict__ center)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y, e = blockIdx.z;
    if (k >= n_need) return;
    const int v = need[k];
    const float dts = (float)(s + 1) * dt;                                   // linspace(1, n, n) * dt, :374
    const float* X = in.xyz + (size_t)e * 3; const float* V = in.vel + (size_t)e * 3; const float* W = in.rot_vel + (size_t)e * 3;
    const float* R = in.rot + (size_t)e * 9;
    const float nx = X[0] + V[0] * dts, ny = X[1] + V[1] * dts, nz = X[2] + V[2] * dts;   // eef_xyz_next, :376
    float D[9];
    eef_aa_to_matrix(W[0] * dts, W[1] * dts, W[2] * dts, D);               // :377-378
    float Rn[9];                                                             // eef_rot_next = D^T @ eef_rot, :379
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = (D[0 * 3 + i] * R[0 * 3 + j] + D[1 * 3 + i] * R[1 * 3 + j]) + D[2 * 3 + i] * R[2 * 3 + j];
    const float* r0 = rel0 + ((size_t)e * M + v) * 3; const float* d = delta + ((size_t)e * M + v) * 3;
    const float rx = r0[0] + (d[0] / dt_n) * dts, ry = r0[1] + (d[1] / dt_n) * dts, rz = r0[2] + (d[2] / dt_n) * dts; // :429
    float* o = interp + (((size_t)e * n_sub + s) * M + v) * 3;             // xyz_next + rel @ Rn^T, :432
    o[0] = nx + ((rx * Rn[0] + ry * Rn[1]) + rz * Rn[2]);
    o[1] = ny + ((rx * Rn[3] + ry * Rn[4]) + rz * Rn[5]);
    o[2] = nz + ((rx * Rn[6] + ry * Rn[7]) + rz * Rn[8]);
    if (k == 0) { float* c = center + ((size_t)e * n_sub + s) * 3; c[0] = nx; c[1] = ny; c[2] = nz; } // :436
}

// ---- warp-style hash grid -----------------------------------------------------------------------------
__device__ __forceinline__ int grid_cell(int x, int y, int z)
{
    const int origin = 1 << 20;
    x = max(0, x + origin); y = max(0, y + origin); z = max(0, z + origin);
    return (z % GRID_DIM) * (GRID_DIM * GRID_DIM) + (y % GRID_DIM) * GRID_DIM + (x % GRID_DIM);
}

// One (cell key, USER index) pair per particle, emitted in user order so that the stable sort leaves every cell's
// points in ascending user index — the traversal order of warp's grid (its ids are the caller's indices).
__global__ void k_grid_keys(int N, int E, const int* __restrict__ inv, const StateC xv, float cell_inv,
                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N) return;
    const float4 q = st_x4(xv, (size_t)e * N + inv[u]);
    const int c = grid_cell((int)(q.x * cell_inv), (int)(q.y * cell_inv), (int)(q.z * cell_inv));
    keys[(size_t)e * N + u] = ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)c;
    vals[(size_t)e * N + u] = (uint32_t)u;
}

__device__ __forceinline__ void cell_range(const uint32_t* __restrict__ keys, int lo0, int hi0, uint32_t key, int& b, int& en)
{
    int lo = lo0, hi = hi0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    b = lo;
    hi = hi0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= key) lo = mid + 1; else hi = mid; }
    en = lo;
}

struct QBox { int xs, ys, zs, xe, ye, ze; };
__device__ __forceinline__ QBox query_box(float4 q, float r, float cell_inv)
{
    QBox b;
    b.xs = (int)((q.x - r) * cell_inv); b.ys = (int)((q.y - r) * cell_inv); b.zs = (int)((q.z - r) * cell_inv);
    b.xe = min((int)((q.x + r) * cell_inv), b.xs + GRID_DIM - 1);
    b.ye = min((int)((q.y + r) * cell_inv), b.ys + GRID_DIM - 1);
    b.ze = min((int)((q.z + r) * cell_inv), b.zs + GRID_DIM - 1);
    return b;
}

// build_resting_collision_pairs, :272-291 (bitset instead of N x N bytes; rows/bits are INTERNAL indices, the
// `index < i` test is on USER indices like the reference)
__global__ void k_build_resting(int N, int E, int words, const int* __restrict__ perm, const int* __restrict__ inv,
                                const StateC xv, float radius, float cell_inv, const uint32_t* __restrict__ keys,
                                const uint32_t* __restrict__ ids, uint32_t* __restrict__ bits, const int* __restrict__ env_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N || (env_mask && env_mask[e] == 0)) return;
    const int ui = perm[i];
    const float4 q = st_x4(xv, (size_t)e * N + i);
    const QBox b = query_box(q, radius, cell_inv);
    uint32_t* my = bits + ((size_t)e * N) * words;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int uj = (int)ids[k];
                    if (uj < ui) {
                        const int j = inv[uj];
                        atomicOr(&my[(size_t)i * words + (j >> 5)], 1u << (j & 31));
                        atomicOr(&my[(size_t)j * words + (i >> 5)], 1u << (i & 31));
                    }
                }
            }
}

// update_potential_collision, :196-227 (same candidate order: cells x-fastest, user ids ascending inside a cell)
__global__ void k_candidates(int N, int E, int words, int cap, const int* __restrict__ inv, const StateC xv,
                             const int* __restrict__ masks, float cd, float radius, float cell_inv, const uint32_t* __restrict__ keys,
                             const uint32_t* __restrict__ ids, const uint32_t* __restrict__ bits, int* __restrict__ coll_idx,
                             int* __restrict__ coll_num, int* __restrict__ max_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t eb = (size_t)e * N;
    const float4 q = st_x4(xv, eb + i);
    const f3 x1 = xyz(q);
    const int mask1 = masks[i];
    // The reference visits every cell overlapping [x - 5cd, x + 5cd] and keeps j only if |xj - xi| < cd.  Such a j
    // lies in a cell overlapping [x - cd, x + cd] (int() truncation is monotonic), and dropping the other cells
    // keeps the relative order of the survivors: visiting the smaller box yields the identical list, ~5x cheaper.
    (void)radius;
    const QBox b = query_box(q, cd, cell_inv);
    const uint32_t* row = bits + (eb + i) * words;
    int cnt = 0;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int j = inv[ids[k]];
                    if (j == i) continue;
                    const f3 dis = st_x(xv, eb + j) - x1;
                    if (!(len(dis) < cd)) continue;          // cheap test first; same set as the reference order
                    if (row[j >> 5] & (1u << (j & 31))) continue; // resting pair (stored symmetrically)
                    if (mask1 == masks[j]) continue;
                    if (cnt < cap) coll_idx[(eb + i) * (size_t)cap + cnt] = j;
                    cnt++;
                }
            }
    coll_num[eb + i] = min(cnt, cap);
    if (cnt > 0) atomicMax(max_count, cnt);
}

// Direct cell table for the per-env-step candidate rebuild: tab[(env << 21) | cell] = [first, last+1) in the sorted key
// array; all-zero between calls (the mark kernel fills the occupied cells, k_cell_clear wipes exactly those again), so a
// lookup is one load instead of two 14-step binary searches.  xs[k] = position and INTERNAL index of the k-th sorted
// particle, so a cell's points stream as consecutive 16-byte records instead of three dependent gathers each.
__global__ void k_cell_clear(int N, int E, const uint32_t* __restrict__ keys, int2* __restrict__ tab)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    tab[keys[(size_t)blockIdx.y * N + k]] = make_int2(0, 0);
}
// ---- candidate rebuild on a FINE grid (cell = collision_dist) ---------------------------------------------------------
// The reference's grid has cells of 5 cd and keeps only neighbours closer than cd, so a query walks ~250 points to keep
// a handful.  Here the points are binned at cell = cd, a query looks at its 3x3x3 fine cells (~16 points), and the
// survivors are put into the reference's order afterwards: that order is (coarse cell in z,y,x-lexicographic traversal,
// user index inside a cell), and a survivor's coarse cell differs from the query's by at most one per axis, so the sort
// key is (rank of the coarse-cell offset in 0..26, user index).  Identical lists, ~15x fewer distance tests.
__device__ __forceinline__ uint32_t fine_cell(int x, int y, int z) { return ((uint32_t)(z & 127) << 14) | ((uint32_t)(y & 127) << 7) | (uint32_t)(x & 127); }

__global__ void k_fine_keys(int N, int E, const StateC xv, float cd_inv, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const float4 q = st_x4(xv, (size_t)e * N + i);
    keys[(size_t)e * N + i] = ((uint32_t)e << GRID_CELL_BITS) | fine_cell((int)(q.x * cd_inv), (int)(q.y * cd_inv), (int)(q.z * cd_inv));
    vals[(size_t)e * N + i] = (uint32_t)i;
}
__global__ void k_fine_mark(int N, int E, const StateC xv, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids,
                            int2* __restrict__ tab, float4* __restrict__ xs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (k >= N) return;
    const size_t g = (size_t)e * N + k;
    const uint32_t key = keys[g];
    if (k == 0 || keys[g - 1] != key) tab[key].x = (int)g;
    if (k == N - 1 || keys[g + 1] != key) tab[key].y = (int)g + 1;
    const int j = (int)ids[g];
    const float4 q = st_x4(xv, (size_t)e * N + j);
    xs[g] = make_float4(q.x, q.y, q.z, __int_as_float(j));
}
__device__ __forceinline__ uint64_t cand_key(float4 qi, float4 qj, float cell_inv, int user_j)
{
    const int dx = (int)(qj.x * cell_inv) - (int)(qi.x * cell_inv), dy = (int)(qj.y * cell_inv) - (int)(qi.y * cell_inv),
              dz = (int)(qj.z * cell_inv) - (int)(qi.z * cell_inv);
    return ((uint64_t)(uint32_t)(((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1)) << 32) | (uint32_t)user_j;
}
__global__ void k_candidates_fine(int N, int E, int words, int cap, const StateC xv, const int* __restrict__ masks,
                                  const int* __restrict__ perm, float cd, float cd_inv, float cell_inv, const int2* __restrict__ tab,
                                  const float4* __restrict__ xs, const uint32_t* __restrict__ bits, int* __restrict__ coll_idx,
                                  int* __restrict__ coll_num, int* __restrict__ max_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t eb = (size_t)e * N;
    const float4 q = st_x4(xv, eb + i);
    const f3 x1 = xyz(q);
    const int mask1 = masks[i];
    const int fx = (int)(q.x * cd_inv), fy = (int)(q.y * cd_inv), fz = (int)(q.z * cd_inv);
    const uint32_t* row = bits + (eb + i) * words;
    int* out = coll_idx + (eb + i) * (size_t)cap;
    int cnt = 0;
    uint64_t worst = 0; // largest key kept so far (only needed once the row is full)
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int2 st = tab[((uint32_t)e << GRID_CELL_BITS) | fine_cell(fx + dx, fy + dy, fz + dz)];
                for (int k = st.x; k < st.y; ++k) {
                    const float4 c = xs[k];
                    const int j = __float_as_int(c.w);
                    if (j == i) continue;
                    if (!(len(xyz(c) - x1) < cd)) continue;
                    if (row[j >> 5] & (1u << (j & 31))) continue; // resting pair (stored symmetrically)
                    if (mask1 == masks[j]) continue;
                    const uint64_t key = cand_key(q, c, cell_inv, perm[j]);
                    cnt++;
                    int n = min(cnt - 1, cap); // entries currently in the row
                    if (n == cap) { // full: keep the cap smallest keys = the first cap of the reference's traversal
                        if (key > worst) continue;
                        n = cap - 1; // the current worst (last entry) drops out
                    }
                    int pos = n; // insertion sort by key
                    while (pos > 0) {
                        const int jp = out[pos - 1];
                        const float4 cp = st_x4(xv, eb + jp);
                        if (cand_key(q, cp, cell_inv, perm[jp]) < key) break;
                        out[pos] = jp;
                        --pos;
                    }
                    out[pos] = j;
                    if (n + 1 == cap) { const int jl = out[cap - 1]; worst = cand_key(q, st_x4(xv, eb + jl), cell_inv, perm[jl]); }
                }
            }
    coll_num[eb + i] = min(cnt, cap);
    if (cnt > 0) atomicMax(max_count, cnt);
}

// compact (env, particle) list of the particles that have candidates (order irrelevant: each is independent)
__global__ void k_cand_list(int N, int E, const int* __restrict__ coll_num, int2* __restrict__ list, int* __restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const int c = coll_num[(size_t)e * N + i];
    if (c > 0) { // per-environment lists (count[4 + e] entries at list + e * N), count[0] = all of them
        list[(size_t)e * N + atomicAdd(count + 4 + e, 1)] = make_int2(e | (c << 12), i); // env (< 2048) | candidate count << 12
        atomicAdd(count, 1);
    }
}

