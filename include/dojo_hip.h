/*
 * dojo_hip.h -- C ABI of libdojo_hip.so: the MI355X-native batched replacement for
 * Dojo.jl's per-timestep contact-implicit forward/backward path.
 *
 * The reference (Dojo.jl, 100 % Julia) has no FFI for this path; the seams this
 * library replaces are plain Julia method calls on a mutable `Mechanism`:
 *
 *   dojo_create        <- Mechanism(origin, bodies, joints, contacts; timestep, input_scaling, gravity)
 *                         src/mechanism/constructor.jl:46-82   (topology is *read*, never rebuilt)
 *   dojo_set_options   <- SolverOptions{T}                     src/solver/options.jl:16-26
 *   dojo_step          <- step!(mechanism, z, u; opts)         src/simulation/step.jl:11-30
 *                         (= set_maximal_state! + set_input! + mehrotra! + update_state!)
 *                         mehrotra!(mechanism; opts)           src/solver/mehrotra.jl:9-73
 *   dojo_get_solution  <- get_solution(mechanism)              src/gradients/finite_difference.jl:1-18
 *                         (what the Julia shim writes back into body.state.vsol/ωsol,
 *                          joint.impulses[2], contact.impulses_dual[2]/impulses[2] so that
 *                          DojoEnvironments.get_state, ant_ars.jl:72-80, keeps working)
 *   dojo_gradients     <- get_maximal_gradients!(mechanism, z, u; opts) / get_maximal_gradients(mechanism)
 *                         src/gradients/state.jl:69-126
 *   dojo_rollout       <- simulate!(mechanism, steps, storage, control!)  src/simulation/simulate.jl:16-36
 *                         with the control callback replaced by pre-sampled inputs U[k]
 *   dojo_set_external_force <- set_external_force!(body; force, torque, vertex)  src/bodies/set.jl:110-115
 *   dojo_simulate      <- simulate!(...; record = true): the same rollout, recording save_to_storage! rows
 *                         src/simulation/storage.jl:50-67, momentum(mechanism, body) src/mechanics/momentum.jl:17-41
 *   dojo_observe       <- get_state(environment)  DojoEnvironments/src/environments.jl:100-102,
 *                         environments/ant_ars.jl:72-80, environments/quadruped_sampling.jl:67-72
 *   dojo_contact_gradients <- get_contact_gradients(mechanism) src/gradients/contact.jl:1-55
 *   dojo_minimal_to_maximal / dojo_maximal_to_minimal / dojo_step_minimal / dojo_minimal_gradients
 *                      <- minimal_to_maximal, maximal_to_minimal  src/mechanism/state.jl:9-66,
 *                         step_minimal_coordinates!               src/simulation/step.jl:42-60,
 *                         get_minimal_gradients!                  src/gradients/state.jl:183-217
 *   dojo_destroy       <- (GC of the Mechanism)
 *
 * All matrices at the ABI are row-major with the environment (batch) index slowest:
 * z[B][13*Nb], u[B][nu], dz[B][12*Nb][12*Nb], du[B][12*Nb][nu] -- except the Jacobians the `_dev` variants leave on
 * the device, which are column-major per environment (dz[B][column][row]: Julia-native, and what the kernels write coalesced).  Scalars are fp64
 * (dtype 0) or fp32 (dtype 1) as chosen at dojo_create; topology/option structs are
 * always fp64 and are cast on upload.  All arithmetic is fp64 in both modes.  An fp32 buffer cannot hold a unit
 * quaternion: with dtype 1 the state a z stands for is (x, v, q/|q|, omega), the kernels normalize q on load (the
 * reference never renormalizes; its formulas mix rotation_matrix, which scales with |q|^2, and vector_rotate, which
 * does not, so a 1e-7 norm error would otherwise be amplified by stiff contacts).  Differentiable steps of a dtype-1 handle keep an
 * internal fp64 work buffer of 8 * 18 * lanes/2 bytes per environment group and gradient column batch (Ant: 129 KB per
 * environment) so that fp32 outputs carry fp32 rounding only.  No exception crosses the boundary: every entry
 * point returns DOJO_OK (0) or a negative error code; dojo_last_error() gives the text of the last
 * failure in the process (one mutex-guarded string) and dojo_handle_error(h) the last failure on that handle.
 *
 * Pointer arguments are *host* pointers for the plain entry points and *device*
 * pointers for the `_dev` variants (used by bench.py / torch so that inputs are
 * resident in HBM when the timed region starts).  The host-pointer entry points synchronize the DEVICE
 * (hipDeviceSynchronize), not a stream: they may follow `_dev` calls that ran on caller streams the handle does not
 * keep (a non-blocking caller stream is not ordered with the null stream), and they move their data over PCIe anyway;
 * latency-sensitive callers stay on the `_dev` variants, which never synchronize.
 */
#ifndef DOJO_HIP_H
#define DOJO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOJO_OK                 0
#define DOJO_ERR_INVALID       -1   /* bad argument / malformed topology            */
#define DOJO_ERR_UNSUPPORTED   -2   /* e.g. kinematic loop, >1 parent joint per body */
#define DOJO_ERR_DEVICE        -3   /* HIP runtime error (text in dojo_last_error)   */
#define DOJO_ERR_NO_DEVICE     -4   /* no gfx950 device visible                      */

/* per-environment solver status, mirrors mehrotra!'s :success / :failed and the
 * "Excessive angular velocity" error() of src/solver/line_search.jl:18-20 */
#define DOJO_STATUS_SUCCESS      0
#define DOJO_STATUS_FAILED       1
#define DOJO_STATUS_EXCESSIVE_W  2

#define DOJO_DTYPE_F64 0
#define DOJO_DTYPE_F32 1

/* Body{T}: src/bodies/constructor.jl:14-27 (mass, inertia only; state is per-env) */
typedef struct DojoBody {
    double mass;
    double inertia[9];            /* row-major 3x3, body (COM) frame */
} DojoBody;

/*
 * JointConstraint{T,N,Nc,TJ,RJ} = Translational half + Rotational half
 * (src/joints/constraints.jl:17-86, translational/constructor.jl:18-67,
 *  rotational/constructor.jl:18-63).  Masks are exported from the live Julia object
 * (they come from an SVD, src/joints/orthogonal.jl:1-12) as
 *   cmask = constraint_mask(joint)  (nl   x 3, rows beyond nl   are ignored)
 *   amask = nullspace_mask(joint)   (3-nl x 3, rows beyond 3-nl are ignored)
 * (src/joints/joint.jl:56-64).
 */
typedef struct DojoJointHalf {
    int32_t nl;                   /* N-lambda: number of constrained directions, 0..3      */
    int32_t nlim;                 /* Nb/2: number of limited minimal coordinates (0..3-nl) */
    double  cmask[9];
    double  amask[9];
    double  spring;               /* joint.spring coefficient                              */
    double  damper;               /* joint.damper coefficient                              */
    double  spring_offset[3];     /* length 3-nl                                           */
    double  limit_lo[3];          /* joint_limits[1], length nlim                          */
    double  limit_hi[3];          /* joint_limits[2], length nlim                          */
} DojoJointHalf;

typedef struct DojoJoint {
    int32_t parent;               /* index into bodies[], -1 = origin (parent_id == 0)     */
    int32_t child;                /* index into bodies[]                                   */
    int32_t spring_on;            /* JointConstraint.spring flag                           */
    int32_t damper_on;            /* JointConstraint.damper flag                           */
    double  vertex_parent[3];     /* translational.vertices[1]                             */
    double  vertex_child[3];      /* translational.vertices[2]                             */
    double  orientation_offset[4];/* rotational.orientation_offset (s, v1, v2, v3)         */
    DojoJointHalf tra;
    DojoJointHalf rot;
} DojoJoint;

/* ContactConstraint{T,8,1,NonlinearContact,4} with SphereHalfSpaceCollision, child = origin
 * (src/contacts/nonlinear.jl:12-48, src/contacts/collisions/sphere_halfspace.jl:11-22) */
typedef struct DojoContact {
    int32_t body;                 /* parent_id as index into bodies[]                      */
    int32_t model;                /* 0: NonlinearContact (src/contacts/nonlinear.jl), 1: ImpactContact (src/contacts/impact.jl),
                                     2: LinearContact (src/contacts/linear.jl: friction pyramid, cone variables [gamma psi beta1..4]);
                                     1 and 2 forward only: the reference has no data Jacobians for them (src/gradients/data.jl:152-192);
                                     one model per mechanism; 2: <= 16 bodies, <= 4 contacts per body                        */
    double  friction_coefficient;
    double  normal[3];            /* collision.contact_normal (1x3)                        */
    double  tangent[6];           /* collision.contact_tangent (2x3, row-major)            */
    double  origin[3];            /* collision.contact_origin                              */
    double  radius;               /* collision.contact_radius                              */
    double  offset[3];            /* collision.contact_offset                              */
    /* body-body contact (src/contacts/collisions/sphere_sphere.jl:11-16; forward only; any of the three contact models): `body` is the contact's parent_id, its sphere is
     * (origin, radius); the child sphere sits on child_body, which must be a body whose joint hangs on `body` (the contact is an edge of
     * the tree next to that joint; the reference's test mechanism has no joint there at all: give the child a Floating joint to `body`).
     * normal / tangent / offset are unused. */
    int32_t collision;            /* 0: SphereHalfSpaceCollision (child = origin), 1: SphereSphereCollision */
    int32_t child_body;           /* collision 1: child_id as index into bodies[]; else ignored */
    double  child_origin[3];      /* collision 1: collision.origin_child                    */
    double  child_radius;         /* collision 1: collision.radius_child                    */
} DojoContact;

typedef struct DojoTopology {
    int32_t n_bodies, n_joints, n_contacts, reserved;
    double  timestep;
    double  input_scaling;
    double  gravity[3];
    const DojoBody*    bodies;    /* mechanism.bodies   order (defines the z layout)       */
    const DojoJoint*   joints;    /* mechanism.joints   order (defines the u layout)       */
    const DojoContact* contacts;  /* mechanism.contacts order                              */
} DojoTopology;

/* SolverOptions{T}: src/solver/options.jl:16-26 (ls_scale is unused by the reference,
 * halving is hard-coded in src/solver/line_search.jl:143; verbose has no device meaning) */
typedef struct DojoSolverOptions {
    double  rtol;                 /* 1e-6 */
    double  btol;                 /* 1e-4 */
    double  undercut;             /* Inf  */
    double  no_progress_undercut; /* 10   */
    int32_t max_iter;             /* 50   */
    int32_t max_ls;               /* 10   */
    int32_t no_progress_max;      /* 3    */
    int32_t reserved;
} DojoSolverOptions;

typedef struct DojoSim* DojoHandle;

/* dimensions derived from the topology (same formulas as the reference) */
typedef struct DojoDims {
    int32_t n_bodies, n_joints, n_contacts;
    int32_t nz;                   /* 13*Nb  maximal state                                  */
    int32_t nx;                   /* 12*Nb  attitude-reduced state (gradient rows/cols)    */
    int32_t nu;                   /* sum over joints of (3-nl_tra)+(3-nl_rot)              */
    int32_t n_joint_impulses;     /* sum over joints of N_j = sum_halves nl + 4*nlim       */
    int32_t n_solution;           /* n = n_joint_impulses + 6*Nb + 8*Nc                    */
    int32_t lanes_per_env;        /* S: wavefront lanes cooperating on one environment     */
} DojoDims;

/* gradient flavour, SURVEY.md §8a note Q2 */
#define DOJO_GRAD_REFERENCE  0    /* literal get_maximal_gradients! (data blocks on the post-update_state! state) */
#define DOJO_GRAD_CONSISTENT 1    /* data blocks on the pre-update state (what test/data.jl checks)               */

int  dojo_device_count(void);
const char* dojo_last_error(void);
/* text of the last failure of a call on THIS handle ("" if none): per-handle, safe with several handles on several threads */
const char* dojo_handle_error(DojoHandle h);

int  dojo_create(const DojoTopology* topo, int32_t batch, int32_t dtype, int32_t device, DojoHandle* out);
void dojo_destroy(DojoHandle h);
int  dojo_get_dims(DojoHandle h, DojoDims* dims);
int  dojo_set_options(DojoHandle h, const DojoSolverOptions* opts);
int  dojo_set_gradient_mode(DojoHandle h, int32_t mode);
/* Accuracy of the device's linear solves.  The reference's block LDU (GraphBasedSystems, call sites
 * src/solver/mehrotra.jl:36-37,49 and the dense `\` of src/gradients/state.jl:99) is an exact direct method; the device
 * eliminates contacts and limits first, which in fp64 costs log10(gamma/s) digits of the body blocks.  Environments whose
 * cones reach max gamma/s > stiffness therefore get every Newton and IFT solve refined against the un-eliminated KKT system
 * (DESIGN.md section 4.5).  INFINITY = never, 0 = always, negative = the default policy: DOJO_DEFAULT_REFINE_STIFFNESS when the
 * solver tolerances are rtol <= 1e-7 or btol <= 1e-6, never at the reference's default tolerances. */
#define DOJO_DEFAULT_REFINE_STIFFNESS 1.0e4
int  dojo_set_refinement(DojoHandle h, double stiffness);

/* step!: z [B,13Nb], u [B,nu] (NULL = zeros) -> z_next [B,13Nb] = the mechanism's internal
 * state after update_state! (x3,v25,q3,w25; SURVEY §8a note Q1), status [B], iters [B]
 * (status / iters may be NULL).  with_gradient != 0 additionally leaves the IFT Jacobians
 * of this step in the handle for dojo_gradients(). */
int  dojo_step(DojoHandle h, const void* z, const void* u, void* z_next,
               int32_t* status, int32_t* iters, int32_t with_gradient);

/* mehrotra!(mechanism; opts)  src/solver/mehrotra.jl:9 -- the seam a single-`Mechanism` drop-in overrides.
 * When mehrotra! runs, set_input! / input_impulse! (src/mechanism/set.jl:40-53, src/joints/translational/input.jl:5-27,
 * src/joints/rotational/input.jl:5-17) have already turned the controls into body impulses and cleared the joints' inputs:
 * jf [B, 6Nb] = per body [state.JF2 (world frame, 3); state.Jtau2 (body frame, 3)], exactly the vector the body residual
 * subtracts (src/integrators/constraint.jl:20-21).  Equivalent to dojo_step(z, u) for the u that produced jf; any
 * external force set with dojo_set_external_force stays in effect.  z, z_next, status, iters as for dojo_step. */
int  dojo_step_impulses(DojoHandle h, const void* z, const void* jf, void* z_next, int32_t* status, int32_t* iters);

/* The vector the reference's step! literally returns (src/simulation/step.jl:28: get_next_state AFTER update_state!, i.e. the
 * internal next state advanced once more with its own velocities -- SURVEY.md section 8a Q1).  dojo_step / dojo_rollout return
 * the internal state (x3, v25, q3, w25), which is what the next step and DojoEnvironments' get_state consume; this maps it to
 * the literal return value: z_out = (x3 + dt v25, v25, q3 (x) xi(w25), w25) per body (src/mechanism/get.jl:126-134). */
int  dojo_next_state(DojoHandle h, const void* z, void* z_out);
int  dojo_next_state_dev(DojoHandle h, const void* z, void* z_out, void* stream);

/* solution of the last step in get_solution order per env:
 * vel [B,6Nb] (v25,w25 per body), joint_imp [B,n_joint_impulses], contact_sg [B,8Nc] ([s(4);gamma(4)] per contact; an
 * ImpactContact uses the first of each four; LinearContact mechanisms: [B,12Nc], [s(6);gamma(6)]).
 * Any pointer may be NULL. */
int  dojo_get_solution(DojoHandle h, void* vel, void* joint_imp, void* contact_sg);
/* mechanism.mu (the central-path parameter kappa of the docs, src/solver/mehrotra.jl:45) per environment when the solve of
 * the last step returned, fp64 [B]: a mehrotra! drop-in writes it back and calls set_entries! so that mechanism.system
 * holds the final un-factored Jacobian and residual like the reference leaves it (src/solver/mehrotra.jl:69). */
int  dojo_get_mu(DojoHandle h, double* mu);
/* Diagnostics of the LAST LINEARIZATION PERFORMED in every environment's last step (quad mappings; without the refining kernels a step skips
 * the set_entries! after its converging iteration, so this is the linearization one iterate before the solution), fp64 [B, 2]: [0] max gamma/s over
 * its cones (what dojo_set_refinement's threshold is compared with), [1] the largest multiplier of the device's un-pivoted
 * Gauss-Jordan eliminations.  The first call switches the recording on (diag may be NULL), later calls read the last step.
 * [0] is 0 while no refinement threshold is in force (reference-default options and no dojo_set_refinement: the kernels then
 * skip the stiffness arithmetic); [1] is 0 unless the library was built with -DDJ_TRACK_GROWTH=1 (a diagnostics build). */
int  dojo_get_diagnostics(DojoHandle h, double* diag);

/* IFT Jacobians of the last dojo_step(..., with_gradient=1):
 * dz [B,12Nb,12Nb] = jacobian_state, du [B,12Nb,nu] = jacobian_control, row-major per environment
 * (transposed on the device from the kernels' column-major layout) */
int  dojo_gradients(DojoHandle h, void* dz, void* du);

/* simulate! with pre-sampled controls: z0 [B,13Nb], U [H,B,nu] (NULL = zeros) ->
 * Z [H,B,13Nb] (state after each step; NULL = keep only the final state, readable with
 * dojo_get_state), status [H,B] (may be NULL) */
int  dojo_rollout(DojoHandle h, const void* z0, const void* U, int32_t H, void* Z, int32_t* status);
int  dojo_get_state(DojoHandle h, void* z);

/* device-pointer variants: all pointers are device memory on the handle's GPU, work is
 * enqueued on `stream` (a hipStream_t passed as void*, NULL = the null stream) and NOT
 * synchronized; outputs are valid after the stream is synchronized. */
int  dojo_step_dev(DojoHandle h, const void* z, const void* u, void* z_next,
                   int32_t* status, int32_t* iters, void* dz, void* du, void* stream);
/* Environment groups.  Environments are independent, so dojo_step_dev steps a batch of >= 512 environments as up to 16
 * groups on internal HIP streams (of >= 256 environments each; an asynchronous handle, below, of a mechanism with one
 * wavefront per workgroup: of >= 64 workgroups, so that batches of 128 .. 2048 Ants are split further): a group that holds an environment running into max_iter (~5x the mean step time for
 * its wavefront) delays only itself.  By default the call forks from and joins into `stream`, i.e. it behaves like one
 * launch on `stream` (such a joined handle steps a batch of more workgroups than the GPU holds at once as TWO groups: each launch has the whole GPU for its
 * rounds of workgroups, the first group's IFT kernel runs under the tail of the second's step kernel; see dojo_set_dispatch_order).  dojo_set_async(h, 1) drops the join: consecutive dojo_step_dev calls then chain per group (group g
 * of call k+1 runs behind group g of call k and behind what `stream` held at call time), and dojo_join(h, stream)
 * -- or any host-pointer entry point -- orders `stream` behind everything in flight.  The caller must not touch the
 * outputs, nor overwrite the inputs, of un-joined calls.  dojo_set_groups(h, n): n groups, at most 16 (n <= 0: automatic; 1: a
 * single launch on the caller's stream).  ROCm runs at most GPU_MAX_HW_QUEUES (default 4) streams concurrently: set it
 * to >= groups + 1 before the HIP runtime starts.  (No counterpart in the reference, which is single-threaded.)
 * dojo_set_async(h, 2): asynchronous AND pipelined -- the IFT kernel of a group's step k runs on a second internal stream of the group, behind
 * its step kernel and NEXT TO the step kernel of step k + 1 (both depend on step k alone; two step -> IFT hand-off records in turn), so that the
 * SIMDs a draining step kernel leaves idle find work.  Same contract (inputs and outputs of un-joined calls stay untouched -- here the IFT of
 * step k still reads z, u of step k while step k + 1 runs: do not hand step k's z in as z_next of step k + 1); dz / du of consecutive calls
 * are written in call order.  Plain solves only (with refinement in force the groups run as under 1); 2 x groups + 1 hardware queues.
 * Where it pays: batches that leave SIMDs idle (Atlas at B = 256 in two groups: 155 k -> 240 k env-steps/s, B = 512: 260 k -> 300 k,
 * profiles/r06_g_pipeline.txt); at batches that fill the GPU the plain groups are as fast or faster (Ant B = 1024 .. 4096: -3 .. -5 %). */
int  dojo_set_async(DojoHandle h, int32_t on);
int  dojo_set_groups(DojoHandle h, int32_t n);
/* Iteration cap of the step kernel (no counterpart in the reference; the algorithm is unchanged decision for decision).  `mehrotra!` (src/solver/
 * mehrotra.jl:9-73) is a serial chain of up to max_iter Newton iterations per environment, and an iteration whose line search is exhausted
 * evaluates the residual up to max_ls times in a row (src/solver/line_search.jl:1-34).  With a cap, a step that is joined into the caller's
 * stream (dojo_step, dojo_step_dev of a handle that is not asynchronous) runs in phases: the step kernel hands every solve that is unfinished
 * after `cap` iterations to a continuation kernel, in which several wavefronts hold the same environment and evaluate the line-search trials
 * alpha, alpha/2, alpha/4, ... side by side before replaying line_search!'s accept / halve decisions over the results; the IFT kernel of the
 * finished environments runs next to it.  Results: bit for bit those of the uncapped loop under the SIMT emulator; on the GPU the continuation
 * kernel is a second instantiation of the lane program (other fused multiply-add contractions), so a continued solve differs in its last bits.
 * Measured on MI355X: a stalled iteration takes 0.087 instead of 0.110 ms; joined Ant steps gain 10 % at B = 512, 4 % at
 * B = 2048 and lose 3 % at B = 4096 (the IFT can no longer start behind its own group's step kernel) -- hence OFF by default.
 * cap > 0: in force for joined steps of the single-wavefront quad mapping (<= 16 bodies) without refinement; 0 or < 0: off (the default; the
 * environment variable DOJO_ITER_CAP=<cap> changes the default, not an explicit setting).  Asynchronous handles and rollouts never cap. */
#define DOJO_DEFAULT_ITERATION_CAP 0
int  dojo_set_iteration_cap(DojoHandle h, int32_t cap);
/* Order in which the step kernel's workgroups are handed to the GPU (no counterpart in the reference: mehrotra!, src/solver/mehrotra.jl:9-73, runs
 * one mechanism).  A launch ends with its last wavefront, and the Newton loops of a batch take 5 ... max_iter iterations: a 50-iteration solve that
 * starts in the launch's last round of workgroups IS that wavefront.  A solve that was long in one step is long in the next (the same contacts are
 * closing), so the library sorts the workgroups of a launch by the iteration counts of the previous step, longest first (a one-workgroup counting
 * sort behind the step's kernels; the step kernel reads its workgroup index through the permutation).
 * mode 0: batch order.  1 (default): sorted where it can matter -- steps joined into the caller's stream (or stepped as a single group) whose batch has
 * more workgroups than the GPU holds at once (asynchronous handles of several groups hide the tail behind the other groups' kernels).  2: always.  Results do not depend on the order: every
 * environment is computed by the same program wherever it runs.  Measurements: DESIGN.md section 6. */
int  dojo_set_dispatch_order(DojoHandle h, int32_t mode);
int  dojo_join(DojoHandle h, void* stream);

/* Multi-GPU (nothing to cite in the reference, which has no multi-device code).  Environments are
 * independent: one process per GPU, each with ONE handle for its contiguous slice of the batch (rank r of W owns the r-th
 * slice), topology and options replicated, no exchange inside the solver.  The only collective is an all-gather of per-rank
 * outputs over RCCL / xGMI, once per rollout chunk:
 *   dojo_comm_unique_id(id)            rank 0: 128 opaque bytes, which the host passes to the other ranks (Julia Distributed,
 *                                      MPI, torch.distributed ...)
 *   dojo_comm_init(h, rank, world, id) every rank: joins the handle's device into the communicator
 *   dojo_allgather_dev(h, send, recv, count, as_int32, stream)
 *                                      recv[world][count] <- send[count] of every rank, in rank order (= batch order);
 *                                      elements are scalars of the handle's dtype, or int32 (status) with as_int32 != 0;
 *                                      ordered on `stream` behind everything the handle has in flight
 * RCCL is loaded on first use (dlopen), single-GPU callers never touch it. */
int  dojo_comm_unique_id(void* id128);
int  dojo_comm_init(DojoHandle h, int32_t rank, int32_t world, const void* id128);
int  dojo_allgather_dev(DojoHandle h, const void* send, void* recv, int64_t count, int32_t as_int32, void* stream);
int  dojo_comm_info(DojoHandle h, int32_t* rank, int32_t* world);
int  dojo_rollout_dev(DojoHandle h, const void* z0, const void* U, int32_t H, void* Z,
                      int32_t* status, void* stream);

/* set_external_force!(body; force, torque, vertex)  src/bodies/set.jl:110-115, read by the body residual
 * src/integrators/constraint.jl:15-18 (state.Fext, state.τext; a3 of SURVEY.md §8a).  fext [B, 6Nb]: per body the force
 * in the world frame and the torque in the body frame, i.e. the two state fields after the reference's
 * `vertex` arithmetic.  The forces stay in effect for every following step of the handle (step!, which never clears
 * them; a rollout applies them at every step, as a controller that sets them at every step) until they are replaced
 * or removed with NULL.  The host variant copies; the device variant keeps the caller's pointer. */
int  dojo_set_external_force(DojoHandle h, const void* fext);
int  dojo_set_external_force_dev(DojoHandle h, const void* fext);

/* simulate!(mechanism, 1:H, storage, control!; record = true)  src/simulation/simulate.jl:16-37  (SURVEY.md §8f-2):
 * dojo_rollout plus the Storage of the trajectory, written by the step kernel itself.  storage [H, B, Nb, 25], one
 * save_to_storage! row (src/simulation/storage.jl:50-67) per solved step and body, taken BEFORE update_state! as
 * simulate! does:  x2(3) q2(4) v15(3) omega15(3) | px(3) pq(3) (momentum(mechanism, body), world frame,
 * src/mechanics/momentum.jl:17-41) | vl(3) = px / m, omegal(3) = J \ (pq in the body frame).
 * Z and status as for dojo_rollout (both may be NULL). */
int  dojo_simulate(DojoHandle h, const void* z0, const void* U, int32_t H, void* Z, void* storage, int32_t* status);
int  dojo_simulate_dev(DojoHandle h, const void* z0, const void* U, int32_t H, void* Z, void* storage,
                       int32_t* status, void* stream);

/* get_state(environment) of DojoEnvironments (environments.jl:100-102; quadruped_sampling.jl:67-72): the minimal state
 * of the mechanism, and with contact_forces != 0 the normal impulse of every contact of the last step clamped to
 * [-1, 1] behind it (get_state(::AntARS), ant_ars.jl:72-80).  obs [B, 2*nu (+ Nc)].
 * dojo_observe reads the state the last dojo_step / dojo_step_minimal / dojo_rollout left on the handle; the device
 * variant takes that state z [B, 13Nb] explicitly (the z_next of the step enqueued before it on `stream`; NULL = the
 * state kept on the handle, e.g. after dojo_step_minimal_dev). */
int  dojo_observe(DojoHandle h, void* obs, int32_t contact_forces);
int  dojo_observe_dev(DojoHandle h, const void* z, void* obs, int32_t contact_forces, void* stream);

/* get_contact_gradients(mechanism)  src/gradients/contact.jl:1-55  (SURVEY.md §8f-3): the Jacobian of the next state
 * [x3; v25; phi3; omega25] w.r.t. the contact data, 5 per contact [friction_coefficient, contact_radius, contact_origin(3)],
 * at the solution of the last differentiable step (dojo_step(..., with_gradient = 1) / dojo_step_dev with dz, du).
 * dc [B, 12Nb, 5Nc] row-major (host variant); the device variant takes the same z, u as that step and writes
 * dc[B][5Nc columns][12Nb rows] (column-major per environment, like dz).  Quad mappings (<= 32 bodies). */
int  dojo_contact_gradients(DojoHandle h, void* dc);
int  dojo_contact_gradients_dev(DojoHandle h, const void* z, const void* u, void* dc, void* stream);

/* Minimal <-> maximal coordinates.  x [B, 2*nu]: per joint, in mechanism.joints order,
 * [dx(nu_tra); dtheta(nu_rot); dv(nu_tra); domega(nu_rot)].
 *   minimal_to_maximal(mechanism, x)   src/mechanism/state.jl:9-22  (+ src/joints/minimal.jl:160-232)
 *   maximal_to_minimal(mechanism, z)   src/mechanism/state.jl:44-66
 *   step_minimal_coordinates!(mechanism, x, u)   src/simulation/step.jl:42-60   (x -> z -> step! -> z' -> x') */
int  dojo_minimal_to_maximal(DojoHandle h, const void* x, void* z);
int  dojo_maximal_to_minimal(DojoHandle h, const void* z, void* x);
int  dojo_step_minimal(DojoHandle h, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters);
int  dojo_minimal_to_maximal_dev(DojoHandle h, const void* x, void* z, void* stream);
int  dojo_maximal_to_minimal_dev(DojoHandle h, const void* z, void* x, void* stream);
int  dojo_step_minimal_dev(DojoHandle h, const void* x, const void* u, void* x_next,
                           int32_t* status, int32_t* iters, void* stream);

/* get_minimal_gradients!(mechanism, y, u; opts)  src/gradients/state.jl:183-217: one step in minimal coordinates and the
 * Jacobians of x_next w.r.t. x and u, = max_to_min_jacobian * (jacobian_state | jacobian_control) * min_to_max_jacobian
 * (src/gradients/state.jl:9-56, 136-181).  jx [B, 2nu, 2nu], ju [B, 2nu, nu], row-major.  With DOJO_GRAD_REFERENCE the
 * two coordinate Jacobians are evaluated where the reference evaluates them (post-update_state! states). */
int  dojo_minimal_gradients(DojoHandle h, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters,
                            void* jx, void* ju);
int  dojo_minimal_gradients_dev(DojoHandle h, const void* x, const void* u, void* x_next, int32_t* status, int32_t* iters,
                                void* jx, void* ju, void* stream);

/* timing helper for bench.py: average duration in ms of the last `n` launches of the
 * step kernel measured with hipEvents on the launch stream (roofline.achieved).
 * With environment groups (dojo_set_groups / the library's default for B >= 512) every group is a launch of its own with its own
 * event pair: dojo_last_kernel_ms / dojo_last_kernel_times then report the LAST GROUP's kernels (1/groups of the batch), and
 * dojo_kernel_time_totals sums intervals that overlap in time and counts `groups` launches per step.  For the duration of a kernel
 * over the whole batch call dojo_set_groups(h, 1) first (what bench.py does for its roofline leg). */
int  dojo_last_kernel_ms(DojoHandle h, double* ms);
/* the same, split by kernel: the step kernel (Newton loop: step!/mehrotra!) and the IFT kernel (the
 * back-solves of get_maximal_gradients!, src/gradients/state.jl:78-126; 0 when not requested) */
int  dojo_last_kernel_times(DojoHandle h, double* step_ms, double* ift_ms);
/* totals over all timed launches since the last reset (the events are kept in a ring, so timing never makes the
 * host wait inside a rollout loop): summed kernel durations in ms and the number of step launches */
int  dojo_kernel_time_totals(DojoHandle h, double* step_ms, double* ift_ms, int64_t* launches, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* DOJO_HIP_H */
