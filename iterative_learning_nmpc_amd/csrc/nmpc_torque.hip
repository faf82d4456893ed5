// nmpc_torque.hip -- the torque layer on gfx950 (C-ABI: include/nmpc_torque.h).
//
// Batched recursive Newton-Euler inverse dynamics with external foot forces (dynamics.py:136-163), the PD
// law around it (mpc.py:592-599) and the recorded PD-target action (RolloutMPC.py:228-250).  One thread
// per robot: the recursion over the tree is serial (a body needs its parent), robots are independent, and
// a batch of rollouts brings thousands of them.  The model is read through wave-uniform (scalar) loads;
// the per-body quantities the recursion has to keep (velocity, acceleration, world rotation on the way
// out; force and moment on the way back) live in a 3.5 KB LDS slice per thread laid out [slot][thread], so
// that the parent look-ups -- a run-time index, which would push register arrays into scratch memory --
// are conflict-free LDS reads.  ~5 kFLOP per robot: the layer is latency-, not throughput-relevant.
#include <hip/hip_runtime.h>

#include "nmpc_device_guard.hpp"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nmpc.h"
#include "../../include/nmpc_torque.h"

namespace nmpc_torque {

constexpr int MAXJ = NMPC_TREE_MAX_JOINTS, MAXF = NMPC_TREE_MAX_FEET;
constexpr int TPB = 32;                    // robots per block: 27 floats x 32 joints x 32 threads = 108 KB of LDS at most
constexpr int SLOTS = 27;                  // w 3, vo 3, dw 3, dvo 3 (reused as moment 3, force 3 ...), Rw 9, n 3, l 3

struct Model {                             // device copy, fixed-size arrays
    int n, nu, nf;
    int parent[MAXJ], type[MAXJ];
    float axis[MAXJ][3], R[MAXJ][9], p[MAXJ][3], mass[MAXJ], com[MAXJ][3], inertia[MAXJ][6];
    int foot_joint[MAXF];
    float foot_offset[MAXF][3];
    float gravity[3];
};

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 v3(const float* p) { return {p[0], p[1], p[2]}; }

struct M3 {
    float m[9];
};
__device__ __forceinline__ V3 mul(const M3& R, V3 a) {
    return {R.m[0] * a.x + R.m[1] * a.y + R.m[2] * a.z, R.m[3] * a.x + R.m[4] * a.y + R.m[5] * a.z, R.m[6] * a.x + R.m[7] * a.y + R.m[8] * a.z};
}
__device__ __forceinline__ V3 mul_t(const M3& R, V3 a) {
    return {R.m[0] * a.x + R.m[3] * a.y + R.m[6] * a.z, R.m[1] * a.x + R.m[4] * a.y + R.m[7] * a.z, R.m[2] * a.x + R.m[5] * a.y + R.m[8] * a.z};
}
__device__ __forceinline__ M3 mul(const M3& A, const M3& B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}

// x_parent = R x_child + p for joint i at coordinate qi: R = R_fix Rot(axis, qi) (revolute) or R_fix, p moved
// along the axis (prismatic).  Rodrigues: Rot = I + sin K + (1 - cos) K^2.
__device__ __forceinline__ void joint_transform(const Model& m, int i, float qi, M3& R, V3& p) {
    M3 Rf;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rf.m[k] = m.R[i][k];
    const V3 ax = v3(m.axis[i]);
    p = v3(m.p[i]);
    if (m.type[i] == 0) {
        float s, c;
        sincosf(qi, &s, &c);
        const float t = 1.0f - c;
        M3 J;
        J.m[0] = 1.0f - t * (ax.y * ax.y + ax.z * ax.z); J.m[1] = -s * ax.z + t * ax.x * ax.y;           J.m[2] = s * ax.y + t * ax.x * ax.z;
        J.m[3] = s * ax.z + t * ax.x * ax.y;             J.m[4] = 1.0f - t * (ax.x * ax.x + ax.z * ax.z); J.m[5] = -s * ax.x + t * ax.y * ax.z;
        J.m[6] = -s * ax.y + t * ax.x * ax.z;            J.m[7] = s * ax.x + t * ax.y * ax.z;             J.m[8] = 1.0f - t * (ax.x * ax.x + ax.y * ax.y);
        R = mul(Rf, J);
    } else {
        R = Rf;
        p = p + mul(Rf, qi * ax);
    }
}

__global__ __launch_bounds__(TPB) void id_torques_kernel(const Model* __restrict__ mp, int B, const float* __restrict__ q,
                                                         const float* __restrict__ v, const float* __restrict__ a,
                                                         const float* __restrict__ f, float* __restrict__ tau) {
    extern __shared__ float body[];                       // [joint][SLOTS][TPB]
    const Model& m = *mp;
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= B) return;
    const int n = m.n;
    auto at = [&](int joint, int slot) -> float& { return body[(joint * SLOTS + slot) * TPB + threadIdx.x]; };
    auto get3 = [&](int joint, int slot) { return V3{at(joint, slot), at(joint, slot + 1), at(joint, slot + 2)}; };
    auto put3 = [&](int joint, int slot, V3 x) { at(joint, slot) = x.x; at(joint, slot + 1) = x.y; at(joint, slot + 2) = x.z; };
    const float* qb = q + (size_t)b * n;
    const float* vb = v + (size_t)b * n;
    const float* ab = a + (size_t)b * n;

    // outward: velocities, accelerations (gravity enters as an acceleration of the world), net force and moment
    for (int i = 0; i < n; ++i) {
        M3 R; V3 p;
        joint_transform(m, i, qb[i], R, p);
        const int par = m.parent[i];
        V3 w_p{0, 0, 0}, vo_p{0, 0, 0}, dw_p{0, 0, 0}, dvo_p{-m.gravity[0], -m.gravity[1], -m.gravity[2]};
        M3 Rw = R;
        if (par >= 0) {
            w_p = get3(par, 0); vo_p = get3(par, 3); dw_p = get3(par, 6); dvo_p = get3(par, 9);
            M3 Rp;
#pragma unroll
            for (int k = 0; k < 9; ++k) Rp.m[k] = at(par, 12 + k);
            Rw = mul(Rp, R);
        }
        V3 w = mul_t(R, w_p), vo = mul_t(R, vo_p + cross(w_p, p));
        V3 dw = mul_t(R, dw_p), dvo = mul_t(R, dvo_p + cross(dw_p, p));
        const V3 ax = v3(m.axis[i]);
        const float qd = vb[i], qdd = ab[i];
        if (m.type[i] == 0) {          // S = (axis; 0):  a += S qdd + v x (S qd)
            dw = dw + qdd * ax + cross(w, qd * ax);
            dvo = dvo + cross(vo, qd * ax);
            w = w + qd * ax;
        } else {                       // S = (0; axis)
            dvo = dvo + qdd * ax + cross(w, qd * ax);
            vo = vo + qd * ax;
        }
        put3(i, 0, w); put3(i, 3, vo); put3(i, 6, dw); put3(i, 9, dvo);
#pragma unroll
        for (int k = 0; k < 9; ++k) at(i, 12 + k) = Rw.m[k];
        // f = I a + v x* (I v) with the spatial inertia about the body origin
        const float mass = m.mass[i];
        const V3 c = v3(m.com[i]);
        const float* I = m.inertia[i];
        auto inertia = [&](V3 x) { return V3{I[0] * x.x + I[1] * x.y + I[2] * x.z, I[1] * x.x + I[3] * x.y + I[4] * x.z, I[2] * x.x + I[4] * x.y + I[5] * x.z}; };
        const V3 h_l = mass * (vo + cross(w, c)), h_n = inertia(w) + cross(c, h_l);
        const V3 g_l = mass * (dvo + cross(dw, c)), g_n = inertia(dw) + cross(c, g_l);
        put3(i, 21, g_n + cross(w, h_n) + cross(vo, h_l));      // moment about the body origin
        put3(i, 24, g_l + cross(w, h_l));                       // force
    }
    // contact forces: world-frame force at the foot point of its body (= - J^T f, dynamics.py:158-161)
    if (f) {
        for (int k = 0; k < m.nf; ++k) {
            const int j = m.foot_joint[k];
            M3 Rw;
#pragma unroll
            for (int e = 0; e < 9; ++e) Rw.m[e] = at(j, 12 + e);
            const V3 l = mul_t(Rw, v3(f + ((size_t)b * m.nf + k) * 3));
            put3(j, 24, get3(j, 24) - l);
            put3(j, 21, get3(j, 21) - cross(v3(m.foot_offset[k]), l));
        }
    }
    // inward: joint torques, forces handed to the parents
    for (int i = n - 1; i >= 0; --i) {
        const V3 fn = get3(i, 21), fl = get3(i, 24);
        const int act = i - (n - m.nu);
        if (act >= 0) tau[(size_t)b * m.nu + act] = dot(v3(m.axis[i]), m.type[i] == 0 ? fn : fl);
        const int par = m.parent[i];
        if (par >= 0) {
            M3 R; V3 p;
            joint_transform(m, i, qb[i], R, p);
            const V3 l_p = mul(R, fl);
            put3(par, 24, get3(par, 24) + l_p);
            put3(par, 21, get3(par, 21) + mul(R, fn) + cross(p, l_p));
        }
    }
}

__global__ void pd_torques_kernel(int B, int n, int nu, const float* __restrict__ tau_ff, const float* __restrict__ q,
                                  const float* __restrict__ v, const float* __restrict__ q_plan, const float* __restrict__ v_plan,
                                  float kp, float kd, float* __restrict__ tau) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * nu) return;
    const size_t b = e / nu, j = b * n + (n - nu) + (e - b * nu);
    tau[e] = (tau_ff ? tau_ff[e] : 0.0f) + kp * (q_plan[j] - q[j]) + kd * (v_plan[j] - v[j]);
}

__global__ void pd_target_action_kernel(int B, int n, int nu, const float* __restrict__ tau, const int* __restrict__ perm,
                                        const float* __restrict__ q, const float* __restrict__ v, float kp, float kd,
                                        float* __restrict__ action) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)B * nu) return;
    const size_t b = e / nu;
    const int i = (int)(e - b * nu);
    int src = perm ? perm[i] : i;
    src = src < 0 ? 0 : (src >= nu ? nu - 1 : src);
    const size_t j = b * n + (n - nu) + i;
    action[e] = (tau[b * nu + src] + kd * v[j]) / kp + q[j];
}

}  // namespace nmpc_torque

// ================================================================================================
namespace {

using namespace nmpc_torque;

thread_local std::string g_torque_create_error;

struct Torque {
    Model host{};
    Model* dev = nullptr;
    int device = 0;
    std::string err;
};

int tfail(Torque* t, int code, const std::string& msg) {
    if (t) t->err = msg; else g_torque_create_error = msg;
    return code;
}

}  // namespace

extern "C" {

int nmpc_torque_create(const nmpc_tree_model* mdl, int device_id, void** handle) {
    if (handle) *handle = nullptr;
    if (!mdl || !handle) return tfail(nullptr, NMPC_E_ARG, "null argument");
    if (mdl->n_joints < 1 || mdl->n_joints > MAXJ || mdl->n_actuated < 1 || mdl->n_actuated > mdl->n_joints ||
        mdl->n_feet < 0 || mdl->n_feet > MAXF)
        return tfail(nullptr, NMPC_E_ARG, "need 1 <= n_actuated <= n_joints <= 32, 0 <= n_feet <= 8");
    if (!mdl->parent || !mdl->type || !mdl->axis || !mdl->placement || !mdl->mass || !mdl->com || !mdl->inertia ||
        (mdl->n_feet > 0 && (!mdl->foot_joint || !mdl->foot_offset)))
        return tfail(nullptr, NMPC_E_ARG, "null model array");
    Model m{};
    m.n = mdl->n_joints; m.nu = mdl->n_actuated; m.nf = mdl->n_feet;
    for (int i = 0; i < m.n; ++i) {
        if (mdl->parent[i] < -1 || mdl->parent[i] >= i) return tfail(nullptr, NMPC_E_ARG, "parents must come before their children");
        if (mdl->type[i] != 0 && mdl->type[i] != 1) return tfail(nullptr, NMPC_E_ARG, "joint type is 0 (revolute) or 1 (prismatic)");
        const float* ax = mdl->axis + 3 * i;
        const float len = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        if (!(std::fabs(len - 1.0f) < 1e-4f)) return tfail(nullptr, NMPC_E_ARG, "joint axes must be unit vectors");
        if (!(mdl->mass[i] >= 0.0f)) return tfail(nullptr, NMPC_E_ARG, "negative mass");
        m.parent[i] = mdl->parent[i]; m.type[i] = mdl->type[i]; m.mass[i] = mdl->mass[i];
        std::memcpy(m.axis[i], ax, 12); std::memcpy(m.R[i], mdl->placement + 12 * i, 36);
        std::memcpy(m.p[i], mdl->placement + 12 * i + 9, 12); std::memcpy(m.com[i], mdl->com + 3 * i, 12);
        std::memcpy(m.inertia[i], mdl->inertia + 6 * i, 24);
    }
    for (int k = 0; k < m.nf; ++k) {
        if (mdl->foot_joint[k] < 0 || mdl->foot_joint[k] >= m.n) return tfail(nullptr, NMPC_E_ARG, "foot_joint out of range");
        m.foot_joint[k] = mdl->foot_joint[k];
        std::memcpy(m.foot_offset[k], mdl->foot_offset + 3 * k, 12);
    }
    std::memcpy(m.gravity, mdl->gravity, 12);
    Torque* t = new Torque();
    t->host = m; t->device = device_id;
    nmpc::DeviceGuard guard(device_id);
    hipError_t e = guard.err;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->dev), sizeof(Model));
    if (e == hipSuccess) e = hipMemcpy(t->dev, &m, sizeof(Model), hipMemcpyHostToDevice);
    if (e == hipSuccess)                                   // more than the default 64 KB of LDS per block
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(id_torques_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                MAXJ * SLOTS * TPB * (int)sizeof(float));
    if (e != hipSuccess) {
        const std::string msg = hipGetErrorString(e);
        if (t->dev) (void)hipFree(t->dev);
        delete t;
        return tfail(nullptr, NMPC_E_HIP, msg);
    }
    *handle = t;
    return NMPC_OK;
}

void nmpc_torque_destroy(void* handle) {
    Torque* t = static_cast<Torque*>(handle);
    if (!t) return;
    nmpc::DeviceGuard guard(t->device);
    if (t->dev) (void)hipFree(t->dev);
    delete t;
}

const char* nmpc_torque_last_error(void* handle) {
    Torque* t = static_cast<Torque*>(handle);
    return t ? t->err.c_str() : g_torque_create_error.c_str();
}

int nmpc_id_torques_batch(void* handle, int B, const float* q, const float* v, const float* a, const float* f, float* tau,
                          void* stream) {
    Torque* t = static_cast<Torque*>(handle);
    if (!t) return tfail(nullptr, NMPC_E_ARG, "null handle");
    if (B == 0) return NMPC_OK;
    nmpc::DeviceGuard guard(t->device);
    if (B < 0 || !q || !v || !a || !tau) return tfail(t, NMPC_E_ARG, "need B >= 0 and q, v, a, tau");
    const size_t lds = (size_t)t->host.n * SLOTS * TPB * sizeof(float);
    hipLaunchKernelGGL(id_torques_kernel, dim3((unsigned)((B + TPB - 1) / TPB)), dim3(TPB), lds, static_cast<hipStream_t>(stream),
                       t->dev, B, q, v, a, f, tau);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NMPC_OK : tfail(t, NMPC_E_HIP, hipGetErrorString(e));
}

int nmpc_pd_torques_batch(void* handle, int B, const float* tau_ff, const float* q, const float* v, const float* q_plan,
                          const float* v_plan, float kp, float kd, float* tau, void* stream) {
    Torque* t = static_cast<Torque*>(handle);
    if (!t) return tfail(nullptr, NMPC_E_ARG, "null handle");
    if (B == 0) return NMPC_OK;
    nmpc::DeviceGuard guard(t->device);
    if (B < 0 || !q || !v || !q_plan || !v_plan || !tau) return tfail(t, NMPC_E_ARG, "need B >= 0 and q, v, q_plan, v_plan, tau");
    const size_t n = (size_t)B * t->host.nu;
    hipLaunchKernelGGL(pd_torques_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), B,
                       t->host.n, t->host.nu, tau_ff, q, v, q_plan, v_plan, kp, kd, tau);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NMPC_OK : tfail(t, NMPC_E_HIP, hipGetErrorString(e));
}

int nmpc_pd_target_action_batch(void* handle, int B, const float* tau, const int* perm, const float* q, const float* v, float kp,
                                float kd, float* action, void* stream) {
    Torque* t = static_cast<Torque*>(handle);
    if (!t) return tfail(nullptr, NMPC_E_ARG, "null handle");
    if (B == 0) return NMPC_OK;
    nmpc::DeviceGuard guard(t->device);
    if (B < 0 || !tau || !q || !v || !action) return tfail(t, NMPC_E_ARG, "need B >= 0 and tau, q, v, action");
    if (!(kp != 0.0f)) return tfail(t, NMPC_E_ARG, "kp must not be zero");
    const size_t n = (size_t)B * t->host.nu;
    hipLaunchKernelGGL(pd_target_action_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       B, t->host.n, t->host.nu, tau, perm, q, v, kp, kd, action);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NMPC_OK : tfail(t, NMPC_E_HIP, hipGetErrorString(e));
}

}  // extern "C"
