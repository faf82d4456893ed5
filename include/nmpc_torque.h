/* nmpc_torque.h -- C-ABI of the torque layer: batched inverse dynamics + PD for the
 * plans the NMPC solve produces (libnmpc_hip.so).
 *
 * Replaces, for B robots at once:
 *   QuadrupedDynamics.id_torques            mpc_controller/utils/dynamics.py:136-163
 *       tau = pin.rnea(model, data, q, v, a)[-nu:]  -  sum_feet f_ee @ J_ee[:3, -nu:]
 *       (J_ee in LOCAL_WORLD_ALIGNED axes: f_ee is a world-frame force at the foot point)
 *   LocomotionMPC._compute_pd_torques       mpc_controller/mpc.py:592-599
 *       tau_pd = tau_ff + Kp (q_plan[-nu:] - q[-nu:]) + Kd (v_plan[-nu:] - v[-nu:])
 *   the recorded action                     DAgger/utils/RolloutMPC.py:228-250
 *       action = (tau + kd v_j) / kp + q_j       (the PD target that reproduces tau), joints re-ordered
 * The robot is a tree of 1-DoF joints with q_dot = v -- the reference's state
 * [px, py, pz, yaw, pitch, roll, joints] (dynamics.py:146-148) is three prismatic and three revolute
 * virtual joints in front of the legs.  The reference builds its model from a URDF through pinocchio
 * (both absent here); this boundary takes the same information as arrays.
 * All batch tensors are fp32 device pointers, row-major [batch][...]; calls are stream-ordered, never
 * allocate and never synchronise.  Return values: NMPC_OK / NMPC_E_* of nmpc.h. */
#ifndef NMPC_TORQUE_H
#define NMPC_TORQUE_H

#ifdef __cplusplus
extern "C" {
#endif

#define NMPC_TREE_MAX_JOINTS 32
#define NMPC_TREE_MAX_FEET 8

/* Host arrays, copied by nmpc_torque_create.  Joint i: parent[i] < i (-1 = world); type 0 revolute,
 * 1 prismatic; unit axis in the joint's own frame; fixed placement of the joint frame in the parent
 * frame, x_parent = R x_joint + p, as 12 floats (R row-major, then p); the body it carries: mass,
 * centre of mass, inertia about the centre of mass in body axes (xx, xy, xz, yy, yz, zz). */
typedef struct {
    int n_joints;             /* generalised coordinates, <= NMPC_TREE_MAX_JOINTS                  */
    int n_actuated;           /* nu: torques of the LAST n_actuated joints are returned            */
    int n_feet;               /* <= NMPC_TREE_MAX_FEET                                             */
    const int *parent;        /* [n_joints]                                                        */
    const int *type;          /* [n_joints]                                                        */
    const float *axis;        /* [n_joints][3]                                                     */
    const float *placement;   /* [n_joints][12]                                                    */
    const float *mass;        /* [n_joints]                                                        */
    const float *com;         /* [n_joints][3]                                                     */
    const float *inertia;     /* [n_joints][6]                                                     */
    const int *foot_joint;    /* [n_feet] joint whose body carries the foot                        */
    const float *foot_offset; /* [n_feet][3] foot point in that body's frame                       */
    float gravity[3];         /* world frame, e.g. {0, 0, -9.81}                                   */
} nmpc_tree_model;

int nmpc_torque_create(const nmpc_tree_model *model, int device_id, void **handle);
void nmpc_torque_destroy(void *handle);
const char *nmpc_torque_last_error(void *handle);

/* id_torques: q, v, a [B][n_joints]; f [B][n_feet][3] (world frame; NULL = no contact forces);
 * tau [B][n_actuated]. */
int nmpc_id_torques_batch(void *handle, int B, const float *q, const float *v, const float *a,
                          const float *f, float *tau, void *stream);

/* _compute_pd_torques: tau_ff [B][nu] (NULL = 0); q, v, q_plan, v_plan [B][n_joints] (their last nu
 * entries are used); tau [B][nu] (may alias tau_ff). */
int nmpc_pd_torques_batch(void *handle, int B, const float *tau_ff, const float *q, const float *v,
                          const float *q_plan, const float *v_plan, float kp, float kd, float *tau,
                          void *stream);

/* Recorded action: action[b][i] = (tau[b][perm[i]] + kd v[b][nj - nu + i]) / kp + q[b][nj - nu + i].
 * perm [nu] (device, NULL = identity) maps the actuator order of tau to the joint order (the
 * reference's ctrl is [FR, FL, RR, RL], its joints [FL, FR, RL, RR]: RolloutMPC.py:229-235). */
int nmpc_pd_target_action_batch(void *handle, int B, const float *tau, const int *perm, const float *q,
                                const float *v, float kp, float kd, float *action, void *stream);

#ifdef __cplusplus
}
#endif
#endif
