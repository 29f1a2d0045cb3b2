/*
 * idocp_hip.h -- C ABI of the MI355X-native KKT-condensation + Riccati hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI:
 * its boundary is the C++ classes idocp::Robot (include/idocp/robot/robot.hpp:26),
 * idocp::UnOCPSolver (include/idocp/unocp/unocp_solver.hpp:25-188) and
 * idocp::OCPSolver (include/idocp/ocp/ocp_solver.hpp).  The C++ facade in
 * include/idocp/ *.hpp keeps those class names / method names and forwards every
 * call to the functions declared here; each entry point cites the reference
 * method it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types
 *   - every function returns an int status: 0 ok, <0 bad argument / runtime
 *     failure (IDOCP_E_*), >0 numerical failure (1 + index of the failing stage)
 *   - caller-owned host buffers, library-owned device buffers; one HIP stream
 *     per handle; a handle is thread-compatible (one caller at a time)
 *   - all arithmetic is IEEE FP64; matrices crossing the ABI are column-major
 *     (the reference's Eigen default)
 *   - "batch" = number of independent OCP instances solved side by side by one
 *     handle (the data-parallel axis of the throughput metric).  batch == 1
 *     reproduces one reference solver object.
 */
#ifndef IDOCP_HIP_H_
#define IDOCP_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define IDOCP_MAX_JOINTS 16
#define IDOCP_MAX_NV 24
#define IDOCP_MAX_NQ 25
#define IDOCP_MAX_CONTACTS 4

#define IDOCP_OK 0
#define IDOCP_E_ARG (-1)      /* invalid argument (reference: throw + std::exit) */
#define IDOCP_E_IO (-2)       /* URDF could not be read / parsed                   */
#define IDOCP_E_DEVICE (-3)   /* HIP runtime error, no GPU, or extension missing   */
#define IDOCP_E_UNSUPPORTED (-4)

#define IDOCP_JOINT_REVOLUTE 0
#define IDOCP_JOINT_FREEFLYER 1

/*
 * Flat rigid-body model: what pinocchio::Model holds for the reference
 * (src/robot/robot.cpp:8-85).  Joint i (0-based) is pinocchio joint i+1; the
 * universe is parent -1.  Spatial vectors are (linear, angular) like
 * pinocchio::Motion / Force.  Rotations are row-major 3x3.
 */
typedef struct idocp_model {
  int njoints;
  int nq, nv, nu;              /* nu = nv - dim_passive                           */
  int has_floating_base;       /* joint 0 is a free-flyer (q: xyz + quat xyzw)    */
  int parent[IDOCP_MAX_JOINTS];
  int jtype[IDOCP_MAX_JOINTS];
  int idx_q[IDOCP_MAX_JOINTS];
  int idx_v[IDOCP_MAX_JOINTS];
  double axis[IDOCP_MAX_JOINTS][3];   /* revolute axis in the joint frame          */
  double plc_R[IDOCP_MAX_JOINTS][9];  /* joint placement in the parent joint frame */
  double plc_p[IDOCP_MAX_JOINTS][3];  /*   x_parent = R x_joint + p                */
  double mass[IDOCP_MAX_JOINTS];      /* body inertia (fixed children merged):     */
  double com[IDOCP_MAX_JOINTS][3];    /*   centre of mass in the joint frame       */
  double inertia[IDOCP_MAX_JOINTS][9];/*   rotational inertia about the com        */
  double gravity[3];                  /* (0, 0, -9.81)                             */
  double q_min[IDOCP_MAX_NV];         /* limits of the nu actuated joints          */
  double q_max[IDOCP_MAX_NV];         /* (reference: Robot::initializeJointLimits, */
  double v_max[IDOCP_MAX_NV];         /*  include/idocp/robot/robot.hxx:699-709)   */
  double u_max[IDOCP_MAX_NV];
  int ncontacts;                      /* point contacts (robot.cpp:66-70)          */
  int contact_frame_id[IDOCP_MAX_CONTACTS]; /* pinocchio frame index               */
  int contact_joint[IDOCP_MAX_CONTACTS];    /* parent joint of the contact frame   */
  double contact_R[IDOCP_MAX_CONTACTS][9];  /* frame placement in that joint frame */
  double contact_p[IDOCP_MAX_CONTACTS][3];
  double total_mass;
} idocp_model_t;

/*
 * ConfigurationSpaceCost (src/cost/configuration_space_cost.cpp:241-397), the
 * cost of configs C1/C2.  Unset weights are zero like the reference's ctor.
 */
/* One more task-space cost component of idocp_cost_t (task_extra): the fields of the task_* block of the cost, per component. */
#define IDOCP_MAX_EXTRA_TASKS 3
typedef struct idocp_task_component {
  int dim;                     /* 3 or 6 */
  int joint;                   /* parent joint of the frame */
  double frame_R[9];           /* placement of the frame in the joint frame, row-major */
  double frame_p[3];
  double weight[6], weightf[6], weighti[6];      /* stage, terminal, impulse-stage weights (layout of task_weight) */
  double ref[12];              /* constant reference: rotation (row-major, 9) then position (3) */
} idocp_task_component_t;

typedef struct idocp_cost {
  double q_ref[IDOCP_MAX_NQ];
  double v_ref[IDOCP_MAX_NV];
  double u_ref[IDOCP_MAX_NV];
  double q_weight[IDOCP_MAX_NV];
  double v_weight[IDOCP_MAX_NV];
  double a_weight[IDOCP_MAX_NV];
  double u_weight[IDOCP_MAX_NV];
  double qf_weight[IDOCP_MAX_NV];
  double vf_weight[IDOCP_MAX_NV];
  /* ContactForceCost (src/cost/contact_force_cost.cpp:153-194): per contact */
  double f_weight[IDOCP_MAX_CONTACTS][3];
  double f_ref[IDOCP_MAX_CONTACTS][3];
  /* TrottingConfigurationSpaceCost (include/idocp/cost/trotting_configuration_space_cost.hpp:
   * 126-164): when use_trotting_ref != 0, q_ref holds q_standing and the reference
   * of stage time t is generated on the host; v_ref[0] = step_length / t_period. */
  int use_trotting_ref;
  double t_start, t_period, step_length;
  double front_swing_knee, hip_swing_knee, front_stance_knee, hip_stance_knee;
  /* Impulse-stage terms (computeImpulseCostDerivatives / Hessian of the same components,
   * src/cost/trotting_configuration_space_cost.cpp:308-376, contact_force_cost.cpp:167-211):
   * weights on q, v, dv and on the impulse forces.  Only used by horizons with impulse events. */
  double qi_weight[IDOCP_MAX_NV];
  double vi_weight[IDOCP_MAX_NV];
  double dvi_weight[IDOCP_MAX_NV];
  double fi_weight[IDOCP_MAX_CONTACTS][3];
  double fi_ref[IDOCP_MAX_CONTACTS][3];
  /* TimeVaryingConfigurationSpaceCost (include/idocp/cost/time_varying_configuration_space_cost.hpp:98-118,
   * src/cost/time_varying_configuration_space_cost.cpp:58-85): when use_time_varying_ref != 0, q_ref holds q_begin,
   * v_ref the constant reference velocity; the reference of stage time t is q_begin for t <= tv_t_begin,
   * q_begin (+) (t - tv_t_begin) v_ref inside (tv_t_begin, tv_t_end) and q_begin (+) (tv_t_end - tv_t_begin) v_ref after;
   * the velocity reference is v_ref inside the window and zero outside. */
  int use_time_varying_ref;
  double tv_t_begin, tv_t_end;
  /* TaskSpace3DCost / TaskSpace6DCost and their TimeVarying variants (src/cost/task_space_{3d,6d}_cost.cpp,
   * time_varying_task_space_{3d,6d}_cost.cpp) on one frame: of a fixed-base robot (UnOCPSolver, also the TimeVarying variants) or of a
   * floating-base one (OCPSolver and ParNMPCSolver on any chain, horizons with discrete events included; constant reference here, the
   * TimeVarying variants through idocp_ocp_set_task_refs; the frame sits on the base or on a link of a leg).  SURVEY 8f row 3.
   * task_dim 0: none; 3: l = 1/2 dt |p_frame(q) - p_ref|^2_W; 6: l = 1/2 dt |log6(M_ref^-1 M_frame(q))|^2_W with the
   * Gauss-Newton Hessian of the reference.  The frame is given by its parent joint and its placement in that joint's
   * frame (idocp_model_frame_placement). */
  int task_dim;
  int task_joint;
  double task_frame_R[9];        /* row-major */
  double task_frame_p[3];
  double task_weight[6];         /* 3D: q_3d_weight; 6D: the vector the reference stores, [rotation_weight; position_weight], which
                                  * multiplies log6 = [linear; angular] entry by entry (time_varying_task_space_6d_cost.cpp:33-38, 60-62) */
  double task_weightf[6];        /* terminal weights (qf_3d_weight / qf_6d_weight), same layout */
  double task_ref[12];           /* constant reference: rotation (row-major, 9) then position (3); 3D uses the position only */
  int task_time_varying;         /* != 0: the references of the N + 1 stages come from idocp_unocp_set_task_refs */
  double task_weighti[6];        /* impulse-stage weights (qi_3d_weight / qi_6d_weight), same layout: OCPSolver on chains with impulse stages */
  /* Further task-space components (CostFunction::push_back takes any number of components, include/idocp/cost/cost_function.hpp:67;
   * round 6): up to IDOCP_MAX_EXTRA_TASKS more TaskSpace3DCost / TaskSpace6DCost terms, each on a frame of its own, summed with the one
   * above.  Constant references (the TimeVarying variants: the first component only).  task_extra_count > 0 needs task_dim != 0. */
  int task_extra_count;
  idocp_task_component_t task_extra[IDOCP_MAX_EXTRA_TASKS];
} idocp_cost_t;

/*
 * Inequality constraints handled by the primal-dual interior point method
 * (include/idocp/constraints/pdipm.hxx:13-87).  The six joint-limit components
 * are what JointConstraintsFactory::create() pushes
 * (src/utils/joint_constraints_factory.cpp:21-38), in the reference's order:
 * position lower/upper, velocity lower/upper, torque lower/upper.
 */
typedef struct idocp_constraints {
  int joint_position_limits;   /* 0/1 */
  int joint_velocity_limits;   /* 0/1 */
  int joint_torque_limits;     /* 0/1 */
  int linearized_friction_cone;/* 0/1: LinearizedFrictionCone (src/constraints/linearized_friction_cone.cpp) */
  double mu;                          /* friction coefficient                     */
  double barrier;                     /* default 1.0e-04 */
  double fraction_to_boundary_rate;   /* default 0.995   */
  int linearized_impulse_friction_cone; /* 0/1: LinearizedImpulseFrictionCone on impulse stages
                                         * (src/constraints/linearized_impulse_friction_cone.cpp), same mu */
  int friction_cone;                    /* 0/1: FrictionCone, two rows per contact: -fz <= 0, fx^2 + fy^2 - mu^2 fz^2 <= 0
                                         * (src/constraints/friction_cone.cpp; the cone of examples/anymal/ocp_benchmark.cpp:76).
                                         * Exclusive with linearized_friction_cone */
  int impulse_friction_cone;            /* 0/1: ImpulseFrictionCone on impulse stages (src/constraints/impulse_friction_cone.cpp).
                                         * Exclusive with linearized_impulse_friction_cone */
  int joint_acceleration_lower_limit;   /* 0/1: JointAccelerationLowerLimit, a.tail(dimu) >= a_min (src/constraints/joint_acceleration_lower_limit.cpp) */
  int joint_acceleration_upper_limit;   /* 0/1: JointAccelerationUpperLimit, a.tail(dimu) <= a_max (src/constraints/joint_acceleration_upper_limit.cpp) */
  double a_min[IDOCP_MAX_NV];           /* the components carry their own bounds (constructor argument amin / amax), one per actuated joint */
  double a_max[IDOCP_MAX_NV];
  int contact_distance;                 /* 0/1: ContactDistance, the frames of the contacts that are NOT active stay above z = 0
                                         * (src/constraints/contact_distance.cpp); floating-base solvers */
} idocp_constraints_t;

/* ---- Robot ------------------------------------------------------------ */

/* Replaces Robot::Robot(path_to_urdf[, contact_frames]) (src/robot/robot.cpp:8-85):
 * reads the URDF and fills *out.  contact_frames are pinocchio frame indices
 * (may be NULL when ncontacts == 0). */
int idocp_model_from_urdf(const char* path_to_urdf, const int* contact_frames,
                          int ncontacts, idocp_model_t* out);

/* Parent joint (index into the model's joint arrays) and placement (R row-major, p) in that joint's frame of the frame with
 * pinocchio-compatible index frame_id: what Robot::framePlacement / getFrameJacobian (robot.hxx:166-188) are evaluated on. */
int idocp_model_frame_placement(const char* path_to_urdf, int frame_id, int* joint, double* R, double* p);
/* Frame name -> pinocchio-compatible frame index (robot.cpp printRobotModel
 * enumerates the same table); returns -1 if absent. */
int idocp_model_frame_id(const char* path_to_urdf, const char* frame_name);

/* World positions points[ncontacts][3] of the contact frames at configuration q:
 * Robot::updateFrameKinematics(q) + Robot::setContactPoints / getContactPoints
 * (include/idocp/robot/robot.hxx:85-91, 262-283).  Host arithmetic (problem
 * set-up before the contact sequence is built, examples/anymal/ocp_benchmark.cpp:
 * 104-106), not part of the hot path. */
int idocp_model_contact_positions(const idocp_model_t* model, const double* q, double* points);
/* Robot::framePosition / frameRotation / framePlacement (include/idocp/robot/robot.hxx:206-233) after updateFrameKinematics(q): world
 * placement of a frame that sits on `joint` with the local placement (R_local row-major [9], p_local [3]) -- the pair
 * idocp_model_frame_placement returns for a frame id of the URDF; joint -1 = fixed to the world.  Host arithmetic. */
int idocp_model_frame_world_placement(const idocp_model_t* model, const double* q, int joint, const double* R_local,
                                      const double* p_local, double* R_world, double* p_world);
/* Robot::integrateConfiguration / subtractConfiguration / normalizeConfiguration (include/idocp/robot/robot.hxx:96-147), host
 * arithmetic: q_out[nq] = q (+) length v (SE(3) exponential on a floating base); diff[nv] = q_plus (-) q_minus
 * (pinocchio::difference(q_minus, q_plus)); the base quaternion of q scaled to unit length.  What an MPC loop does between
 * two solver calls; the stage kernels carry their own copies of the same functions. */
int idocp_model_integrate_configuration(const idocp_model_t* model, const double* q, const double* v, double length,
                                        double* q_out);
int idocp_model_subtract_configuration(const idocp_model_t* model, const double* q_plus, const double* q_minus,
                                       double* diff);
int idocp_model_normalize_configuration(const idocp_model_t* model, double* q);

/* Guards against a driver compiled against an older header than the loaded library:
 * pass sizeof(idocp_model_t), sizeof(idocp_cost_t), sizeof(idocp_constraints_t);
 * returns IDOCP_E_ARG on a mismatch.  The C++ facade calls it from idocp::Robot. */
int idocp_abi_check(unsigned long model_size, unsigned long cost_size, unsigned long constraints_size);

void idocp_cost_init(idocp_cost_t* cost);                 /* all zero          */
void idocp_constraints_init(idocp_constraints_t* c);      /* reference defaults */

/* ---- UnOCPSolver (fixed-base, no contacts) ----------------------------- */

typedef struct idocp_unocp idocp_unocp_t;

/* UnOCPSolver::UnOCPSolver(robot, cost, constraints, T, N, nthreads)
 * (src/unocp/unocp_solver.cpp:11-48).  nthreads has no meaning on the GPU; it
 * is replaced by `batch` independent instances and a device ordinal. */
int idocp_unocp_create(const idocp_model_t* model, const idocp_cost_t* cost,
                       const idocp_constraints_t* constraints, double T, int N,
                       int batch, int device, idocp_unocp_t** out);
void idocp_unocp_destroy(idocp_unocp_t* h);
/* Deep copy of a UnOCPSolver / UnParNMPCSolver handle (the reference classes are copyable,
 * unocp_solver.hpp:59-62): same problem, same device, every device buffer and the line-search
 * filter copied. */
int idocp_unocp_clone(idocp_unocp_t* src, idocp_unocp_t** out);
/* Replace the cost of a UnOCPSolver / UnParNMPCSolver handle: the counterpart of idocp_ocp_set_cost.  The reference's
 * solvers share the CostFunction with the driver (unocp_solver.hpp: shared_ptr members), so references and weights changed
 * between two updateSolution calls take effect at the next one; here the cost is copied at creation and this call is how
 * an MPC loop moves its goal.  A task-space component cannot be added or removed (IDOCP_E_UNSUPPORTED). */
int idocp_unocp_set_cost(idocp_unocp_t* h, const idocp_cost_t* cost);

/* UnOCPSolver::setSolution(name, value) (unocp_solver.cpp:157-181): name in
 * {"q","v","a","u"}; value[dim] is written to every stage of every instance,
 * then the constraints are re-initialised. */
int idocp_unocp_set_solution(idocp_unocp_t* h, const char* name,
                             const double* value);
/* Same, one value per instance: values[batch][dim]. */
int idocp_unocp_set_solution_batch(idocp_unocp_t* h, const char* name,
                                   const double* values);
/* OCPSolver::setSolution / ParNMPCSolver::setSolution (ocp_solver.cpp:116-165, parnmpc_solver.cpp:128-180) write the
 * value and leave the slack / dual variables as they are (the driver calls initConstraints(t)): the setter behind the
 * facade's idocp::OCPSolver / idocp::ParNMPCSolver on a fixed-base robot without contacts, which are bound to these
 * kernels -- with no contact rows the two formulations condense one Newton system in two orders
 * (tests/test_oracle_fixed_base.py). */
int idocp_unocp_set_solution_only(idocp_unocp_t* h, const char* name, const double* value);
/* Per-stage reference poses of a TimeVaryingTaskSpace3DCost / TimeVaryingTaskSpace6DCost (cost.task_dim != 0): the values of
 * TimeVaryingTaskSpace6DRefBase::compute_q_6d_ref (time_varying_task_space_6d_cost.hpp:21-42) at t + i dt, i = 0 .. N.
 * refs: host, [N + 1][12] = rotation (row-major) then position; a 3D cost reads the position only.  Without a call the
 * constant cost.task_ref applies to every stage. */
int idocp_unocp_set_task_refs(idocp_unocp_t* h, const double* refs);
/* UnOCPSolver::initConstraints() (unocp_solver.cpp:59-70). */
int idocp_unocp_init_constraints(idocp_unocp_t* h);

/* UnOCPSolver::updateSolution(t, q, v, line_search) (unocp_solver.cpp:73-134).
 * q[batch][nq], v[batch][nv] are host pointers.  line_search != 0: filter line
 * search (unocp_solver.cpp:116-120), one filter per instance, see below. */
int idocp_unocp_update_solution(idocp_unocp_t* h, double t, const double* q,
                                const double* v, int line_search);
/* Same with q, v already resident in device memory (HBM); asynchronous on the
 * handle's stream -- call idocp_unocp_synchronize() before reading results. */
int idocp_unocp_update_solution_device(idocp_unocp_t* h, double t,
                                       const double* d_q, const double* d_v);
int idocp_unocp_synchronize(idocp_unocp_t* h);
/* Native stream handle (hipStream_t) so callers can record HIP events on it. */
void* idocp_unocp_stream(idocp_unocp_t* h);
/* Device allocation helpers for callers without a HIP binding (bench/tests). */
int idocp_device_alloc(void** d_ptr, unsigned long long nbytes);
int idocp_device_free(void* d_ptr);
int idocp_device_upload(void* d_dst, const void* h_src, unsigned long long nbytes);
int idocp_device_count(int* count);

/* UnOCPSolver::computeKKTResidual(t, q, v) + KKTError() (unocp_solver.cpp:
 * 184-225): kkt_error[batch]. */
int idocp_unocp_compute_kkt_residual(idocp_unocp_t* h, double t, const double* q,
                                     const double* v);
int idocp_unocp_kkt_error(idocp_unocp_t* h, double* kkt_error);

/* UnOCPSolver::getSolution(name) (unocp_solver.cpp:240-309): name in
 * {"q","v","a","u","lmd","gmm","beta"}; out[(N+1)][dim] for one instance
 * (a, u, beta: N stages). */
int idocp_unocp_get_solution(idocp_unocp_t* h, const char* name, int instance,
                             double* out);
/* UnOCPSolver::getSolution(stage) (unocp_solver.hpp: `const SplitSolution& getSolution(int) const`):
 * the split solution of ONE stage in one device-to-host copy; out[7 nv] = lmd gmm q v a u beta
 * (a, u, beta are meaningless on the terminal stage N). */
int idocp_unocp_get_split_solution(idocp_unocp_t* h, int instance, int stage, double* out);
/* Newton direction of the last updateSolution (parity tests): name in
 * {"dq","dv","da","du","dlmd","dgmm","dbeta"}; same shapes. */
int idocp_unocp_get_direction(idocp_unocp_t* h, const char* name, int instance,
                              double* out);
/* Step sizes of the last updateSolution: primal[batch], dual[batch]. */
int idocp_unocp_get_step_sizes(idocp_unocp_t* h, double* primal, double* dual);
/* Riccati factorization of one instance: P[N+1][2nv*2nv] (col-major, blocks
 * [Pqq Pqv; Pvq Pvv]), s[N+1][2nv], K[N][nv*2nv] (col-major), k[N][nv]. Any
 * pointer may be NULL.  (split_riccati_factorization.hpp:15-134,
 * lqr_state_feedback_policy.hpp:11-28) */
int idocp_unocp_get_riccati(idocp_unocp_t* h, int instance, double* P, double* s,
                            double* K, double* k);
/* OCPSolver::getStateFeedbackGain (ocp_solver.cpp:101-111) on the fixed-base robot without contacts: the TORQUE policy
 * du = Kq dq + Kv dv, i.e. the acceleration policy above mapped through the linearised inverse dynamics of the same
 * linearisation (unconstrained_dynamics.hxx:84-92): Kq = dID/dq + M Ka_q, Kv = dID/dv + M Ka_v; nv x nv col-major. */
int idocp_unocp_get_torque_feedback_gain(idocp_unocp_t* h, int instance, int stage,
                                         double* Kq, double* Kv);
/* Slack / dual variables of the IPM, [N][dimc] for one instance. */
int idocp_unocp_get_constraint_data(idocp_unocp_t* h, int instance,
                                    double* slack, double* dual);
int idocp_unocp_dimc(const idocp_unocp_t* h);
/* UnOCPSolver::isCurrentSolutionFeasible (unocp_solver.cpp:228-237): feasible[batch]
 * = 1 if the primal iterate satisfies the joint limits on every stage (with the
 * time-step gating of constraints_data.hpp:18-42), else 0; where[batch] (may be
 * NULL) = first offending stage or -1. */
int idocp_unocp_is_current_solution_feasible(idocp_unocp_t* h, int* feasible, int* where);

/* Filter line search of the two fixed-base solvers (UnLineSearch, include/idocp/line_search/unline_search.hpp:62-92;
 * LineSearchFilter, src/line_search/line_search_filter.cpp): idocp_unocp_update_solution / idocp_unparnmpc_update_solution
 * with line_search != 0 evaluate the trial iterates on the device and keep one filter per instance.
 * UnOCPSolver / UnParNMPCSolver::clearLineSearchFilter: */
int idocp_unocp_clear_line_search_filter(idocp_unocp_t* h);
/* UnLineSearch::computeCostAndViolation of s + alpha[b] d for every instance (alpha = 0: the iterate itself), with the
 * measured state of the last host-pointer update / residual call: cost[batch], violation[batch]. */
int idocp_unocp_line_search_eval(idocp_unocp_t* h, const double* alpha, double* cost,
                                 double* violation);

/* ---- UnParNMPCSolver (include/idocp/unocp/unparnmpc_solver.hpp:31-189, src/unocp/unparnmpc_solver.cpp) ----
 * The stage-parallel solver of the fixed-base problem: N backward-Euler stages (stage i at t + (i+1) dt, created with
 * constraint time step i + 1, the last one terminal), per-stage KKT inverse, coarse update and the four correction
 * sweeps of UnBackwardCorrection (src/unocp/unbackward_correction.cpp:55-132).  The handle type is shared with
 * UnOCPSolver: idocp_unocp_set_solution[_batch], _init_constraints, _get_solution, _get_direction, _get_step_sizes,
 * _kkt_error, _get_constraint_data, _is_current_solution_feasible, _synchronize, _stream and _destroy work on it (the
 * stage-wise getters return the N stages in their first N rows); the entry points below replace the rest. */
int idocp_unparnmpc_create(const idocp_model_t* model, const idocp_cost_t* cost,
                           const idocp_constraints_t* constraints, double T, int N, int batch,
                           int device, idocp_unocp_t** out);
/* UnParNMPCSolver::initBackwardCorrection (unparnmpc_solver.cpp:69-71) */
int idocp_unparnmpc_init_backward_correction(idocp_unocp_t* h, double t);
/* UnParNMPCSolver::updateSolution (unparnmpc_solver.cpp:74-103); q, v: [batch][nv] on the host */
int idocp_unparnmpc_update_solution(idocp_unocp_t* h, double t, const double* q, const double* v,
                                    int line_search);
/* the same with device-resident (q, v), asynchronous on the handle's stream */
int idocp_unparnmpc_update_solution_device(idocp_unocp_t* h, double t, const double* d_q,
                                           const double* d_v);
/* UnParNMPCSolver::computeKKTResidual (unparnmpc_solver.cpp:169-187); read it with idocp_unocp_kkt_error */
int idocp_unparnmpc_compute_kkt_residual(idocp_unocp_t* h, double t, const double* q,
                                         const double* v);
/* One phase of updateSolution (parity tests, roofline measurement): 0 linearize, 1 KKT inverse + coarse update,
 * 2 backward serial, 3 backward parallel, 4 forward serial, 5 forward parallel + direction + step sizes, 6 integrate. */
int idocp_unparnmpc_launch_phase(idocp_unocp_t* h, int phase, const double* d_q, const double* d_v);
/* The coarse / corrected iterate s_new of the backward correction (fields lmd, gmm, a, q, v), like
 * idocp_unocp_get_solution. */
int idocp_unparnmpc_get_new_solution(idocp_unocp_t* h, const char* name, int instance, double* out);

/* Horizon sharding of UnParNMPCSolver (the fixed-base twin of idocp_parnmpc_create_shard; protocol and halo kinds as in
 * idocp_amd/parnmpc_dist.py): the handle holds the stages [stage_begin, stage_end) of a horizon of N stages over T.
 * Halo kinds: 0 state_last (q, v of the last stage -> the right neighbour's previous state), 1 costate_first (lmd, gmm of the
 * first stage -> left), 2 aux_first (aux_mat of the first stage -> left), 3 bwd_first (corrected lmd, gmm of the first stage
 * -> left, pipeline of the backward serial sweep), 4 fwd_last (corrected q, v of the last stage -> right, pipeline of the
 * forward serial sweep).  Buffers are device pointers, [batch][idocp_unparnmpc_halo_size(kind)]. */
int idocp_unparnmpc_create_shard(const idocp_model_t* model, const idocp_cost_t* cost,
                                 const idocp_constraints_t* constraints, double T, int N, int stage_begin,
                                 int stage_end, int batch, int device, idocp_unocp_t** out);
int idocp_unparnmpc_halo_size(int kind);
int idocp_unparnmpc_export_halo(idocp_unocp_t* h, int kind, double* d_buf);
int idocp_unparnmpc_import_halo(idocp_unocp_t* h, int kind, const double* d_buf);
/* device buffers of the previous state (rank 0: the measured state) and of the (primal, dual) step sizes [batch][2] */
int idocp_unparnmpc_prev_state(idocp_unocp_t* h, double** d_q, double** d_v);
int idocp_unparnmpc_step_sizes_device(idocp_unocp_t* h, double** d_steps);
/* squared KKT error of the local stages, d_err2[batch] on the device (to be summed over the ranks) */
int idocp_unparnmpc_kkt_error_squared_device(idocp_unocp_t* h, double t, double* d_err2);

/* Kernel-level entry points used by the parity tests and the roofline
 * measurement (one launch each, on the handle's stream). */
int idocp_unocp_launch_linearize(idocp_unocp_t* h, double t, const double* d_q,
                                 const double* d_v);
int idocp_unocp_launch_riccati(idocp_unocp_t* h, const double* d_q,
                               const double* d_v);
int idocp_unocp_launch_expand(idocp_unocp_t* h);
int idocp_unocp_launch_integrate(idocp_unocp_t* h);
/* One single kernel launch, for per-kernel timing: kernel_id 0 = K1 linearize,
 * 1 = S1 backward Riccati, 2 = S2 forward Riccati, 3 = K2 expand, 4 = step-size
 * reduction, 5 = K3 integrate. */
int idocp_unocp_launch_kernel(idocp_unocp_t* h, int kernel_id, const double* d_q,
                              const double* d_v);
/* Inverse dynamics + its derivatives for `n` independent samples on the device
 * (Robot::RNEA / RNEADerivatives, include/idocp/robot/robot.hxx:444-500).
 * q[n][nq], v[n][nv], a[n][nv] host; tau[n][nv]; dq/dv/da[n][nv*nv] col-major. */
int idocp_rnea_derivatives(const idocp_model_t* model, int n, const double* q,
                           const double* v, const double* a, double* tau,
                           double* dtau_dq, double* dtau_dv, double* dtau_da,
                           int device);

/* ---- OCPSolver (floating base + point contacts) --------------------------- */

typedef struct idocp_ocp idocp_ocp_t;

/* OCPSolver::OCPSolver(robot, cost, constraints, T, N, max_num_impulse, nthreads)
 * (src/ocp/ocp_solver.cpp:9-47).  This build carries the kernels for a quadruped
 * (free-flyer + 4 legs x 3 revolute joints, one point contact per foot) and a
 * contact sequence WITHOUT discrete events (setContactStatusUniformly only);
 * created this way it holds event-free horizons only (max_num_impulse = 0). */
int idocp_ocp_create(const idocp_model_t* model, const idocp_cost_t* cost,
                     const idocp_constraints_t* constraints, double T, int N,
                     int batch, int device, idocp_ocp_t** out);
/* OCPSolver(robot, cost, constraints, T, N, max_num_impulse, nthreads) (ocp_solver.cpp:10-47):
 * reserves stages for up to max_num_impulse discrete events (impulse + aux stage or a
 * lift stage each) so that contact sequences can be pushed. */
int idocp_ocp_create_hybrid(const idocp_model_t* model, const idocp_cost_t* cost,
                            const idocp_constraints_t* constraints, double T, int N,
                            int max_num_impulse, int batch, int device, idocp_ocp_t** out);
void idocp_ocp_destroy(idocp_ocp_t* h);
/* OCPSolver::setContactStatusUniformly (ocp_solver.cpp:169-171): active[ncontacts]
 * flags and the world contact points contact_points[ncontacts][3]
 * (ContactStatus::setContactPoints). */
int idocp_ocp_set_contact_status_uniformly(idocp_ocp_t* h, const int* active,
                                           const double* contact_points);
/* OCPSolver::pushBackContactStatus / setContactPoints / popBackContactStatus /
 * popFrontContactStatus (ocp_solver.cpp:174-197) -> ContactSequence (include/idocp/hybrid/
 * contact_sequence.hxx:52-268) and DiscreteEvent (discrete_event.hxx:57-84): a status
 * that activates a contact is an impulse event (impulse + aux stage and a switching
 * constraint two stages ahead), one that only deactivates contacts is a lift event. */
int idocp_ocp_push_back_contact_status(idocp_ocp_t* h, const int* active,
                                       const double* contact_points, double switching_time);
int idocp_ocp_set_contact_points(idocp_ocp_t* h, int contact_phase, const double* contact_points);
int idocp_ocp_pop_back_contact_status(idocp_ocp_t* h);
int idocp_ocp_pop_front_contact_status(idocp_ocp_t* h);
/* The stages in time order after OCPDiscretizer::discretizeOCP(contact_sequence, t)
 * (include/idocp/hybrid/ocp_discretizer.hxx:65-374).  Returns the chain length
 * M = N + 1 + 2 N_impulse + N_lift (or a negative error code); arrays of `capacity`
 * entries, any may be NULL: kind (0 stage, 1 impulse, 2 aux, 3 lift, 4 terminal), index
 * (grid stage / impulse index / lift index), storage slot, time step, dimf, rows of the
 * switching constraint carried by the stage. */
/* Warm start along the chain of the current discretisation (event stages included): values[M][dim] in the order of
 * idocp_ocp_get_chain, written to every instance.  (The grid-stage setters idocp_ocp_set_solution_stages /
 * idocp_parnmpc_set_aux_mat cannot reach the impulse / aux / lift slots; this is what an MPC loop that shifts a solution with
 * discrete events along the horizon needs -- the reference does it through the public members of its Solution container.) */
int idocp_ocp_set_solution_chain(idocp_ocp_t* h, const char* name, int M, const double* values);
int idocp_parnmpc_set_aux_mat_chain(idocp_ocp_t* h, int M, const double* values);
/* ... and its counterpart: aux_mat of every stage of the chain of one instance, out[M][nx * nx] (column-major per stage).  With
 * idocp_ocp_get_solution_chain this carries a converged ParNMPC solver over into another handle (another batch size, a shard). */
int idocp_parnmpc_get_aux_mat_chain(idocp_ocp_t* h, int instance, double* out);
/* The coarse / corrected iterate s_new of the backward correction along the chain (SplitBackwardCorrection::s_new_,
 * split_backward_correction.hxx:50-82; after idocp_parnmpc_launch_phase 0 - 2 the coarse update, after the correction phases the corrected
 * iterate).  name: lmd gmm u q v xi; on an impulse stage "u" holds the impulse forces f and "xi" the multipliers mu of the velocity
 * constraint, both packed (active contacts first); "xi" of an aux stage is its switching multiplier.  out[M][dim], dim = nv, nv, nu, nq,
 * nv, 3 * max_point_contacts. */
int idocp_parnmpc_get_new_solution_chain(idocp_ocp_t* h, const char* name, int instance, double* out);
int idocp_ocp_get_chain(idocp_ocp_t* h, double t, int capacity, int* kind, int* index, int* slot,
                        double* dt, int* dimf, int* sw_dimi);
/* TimeVaryingTaskSpace3DCost / TimeVaryingTaskSpace6DCost on a floating-base robot (src/cost/time_varying_task_space_3d_cost.cpp,
 * time_varying_task_space_6d_cost.cpp: the reference object is asked for its pose at the time of every stage).  The device cannot call the
 * caller's reference object, so the poses are evaluated up front: idocp_ocp_get_chain_times lists the times of the M stages of the chain
 * discretised at t (the order of idocp_ocp_get_chain: grid stages at t + i dt, impulse / aux / lift stages at their event times, the
 * terminal stage at t + T; returns M), idocp_ocp_set_task_refs hands the M poses over, refs[M][12] = rotation (row-major, 9; identity for
 * the 3D cost) then position (3).  They apply to every call with this t; a call with another t, or a chain of another length, fails with
 * IDOCP_E_ARG until new poses are set.  cost.task_time_varying must be set at creation (OCPSolver and ParNMPCSolver on any chain,
 * discrete events included -- as for the constant-reference costs). */
int idocp_ocp_get_chain_times(idocp_ocp_t* h, double t, int capacity, double* times);
int idocp_ocp_set_task_refs(idocp_ocp_t* h, double t, int M, const double* refs);
/* OCPSolver::setSolution (ocp_solver.cpp:95-165): name in {"q","v","a","f","u"};
 * "f" takes one 3-vector written to every contact.  Does not re-initialise the
 * constraints (like the reference). */
int idocp_ocp_set_solution(idocp_ocp_t* h, const char* name, const double* value);
int idocp_ocp_set_solution_batch(idocp_ocp_t* h, const char* name, const double* values);
/* Warm start (an MPC loop keeps the iterate of the previous sampling instant; the reference does so implicitly because the solver
 * object lives on): one field of every grid stage, values[nstages][dim], the same for all instances.  name: q v a u beta f mu
 * (dim 3 * max contacts) lmd gmm nu_passive; nstages <= N + 1 for lmd gmm q v of an OCPSolver, <= N otherwise. */
int idocp_ocp_set_solution_stages(idocp_ocp_t* h, const char* name, int nstages, const double* values);
/* OCPSolver::initConstraints(t) (ocp_solver.cpp:60-64). */
int idocp_ocp_init_constraints(idocp_ocp_t* h, double t);
/* OCPSolver::updateSolution (ocp_solver.cpp:67-92). q[batch][nq], v[batch][nv].  line_search != 0: the filter line search of
 * src/line_search/line_search.cpp on the primal step (one filter per instance; cost and l1 constraint violation of the trial
 * iterates are evaluated on the device, the filter logic runs on the host). */
int idocp_ocp_update_solution(idocp_ocp_t* h, double t, const double* q,
                              const double* v, int line_search);
int idocp_ocp_update_solution_device(idocp_ocp_t* h, double t, const double* d_q,
                                     const double* d_v);
/* OCPSolver::clearLineSearchFilter (ocp_solver.cpp:196-199). */
int idocp_ocp_clear_line_search_filter(idocp_ocp_t* h);
/* Probing the line search (tests): the Newton direction and step sizes of the current iterate WITHOUT integrating, then
 * LineSearch::computeCostAndViolation of s (+) alpha[b] d per instance (alpha = 0: the iterate itself). */
int idocp_ocp_compute_direction(idocp_ocp_t* h, double t, const double* q, const double* v);
int idocp_ocp_line_search_eval(idocp_ocp_t* h, const double* alpha, double* cost, double* violation);
/* The same iteration submitted as ONE hipGraph launch (captured on first use, re-captured when the discretisation or the input
 * buffers change): the latency mode of the solver -- at batch 1 the launches, not the kernels, set the pace. */
int idocp_ocp_update_solution_graph(idocp_ocp_t* h, double t, const double* d_q,
                                    const double* d_v);
int idocp_ocp_synchronize(idocp_ocp_t* h);
void* idocp_ocp_stream(idocp_ocp_t* h);
/* OCPSolver::computeKKTResidual + KKTError (ocp_solver.cpp:202-213). */
int idocp_ocp_compute_kkt_residual(idocp_ocp_t* h, double t, const double* q,
                                   const double* v);
int idocp_ocp_kkt_error(idocp_ocp_t* h, double* kkt_error);
/* OCPSolver::getSolution(name): q v a u f lmd gmm beta mu nu_passive; out[(N+1)][dim]
 * (f, mu: [ncontacts*3] per stage; stage-only fields fill N rows). */
int idocp_ocp_get_solution(idocp_ocp_t* h, const char* name, int instance, double* out);
/* OCPSolver::getSolution(stage) (ocp_solver.hpp:97) / ParNMPCSolver::getSolution(stage): the split
 * solution of ONE grid stage in one device-to-host copy -- the call an MPC loop makes every cycle.
 * out[3 nv + nq + nv + nu + nv + 2 * 3 ncontacts + 6] = lmd gmm q v a u beta f mu nu_passive
 * (split_solution.hxx:10-31); only lmd gmm q v are meaningful on the terminal stage. */
int idocp_ocp_get_split_solution(idocp_ocp_t* h, int instance, int stage, double* out);
/* The contact status grid stage `stage` of the current discretisation is linearised with
 * (SplitSolution::setContactStatus / isContactActive / dimf, split_solution.hxx:41-57): active[ncontacts]
 * flags; returns dimf (3 per active contact; 0 on the terminal stage) or a negative error code.  It sizes
 * SplitSolution::f_stack() / mu_stack() (split_solution.hpp:93-122) in the facade's getSolution(stage). */
int idocp_ocp_get_stage_contact_status(idocp_ocp_t* h, int stage, int* active);
/* Newton direction: dq dv da du df dlmd dgmm dbeta dmu dnu_passive dxi.  df / dmu are [ncontacts][3] per stage with zeros for the
 * contacts that are not active there, dxi has zeros behind the stage's switching-constraint rows (SplitDirection::df / dmu / dxi only
 * have the active rows, split_direction.hxx:150-229). */
int idocp_ocp_get_direction(idocp_ocp_t* h, const char* name, int instance, double* out);
/* The same fields for every stage of the chain (incl. impulse / aux / lift stages), in
 * chain order: out[M][dim].  Extra names: "xi" / "dxi" (multiplier of the switching
 * constraint, max_dimf entries).  On impulse stages "a" / "da" hold dv / ddv. */
int idocp_ocp_get_solution_chain(idocp_ocp_t* h, const char* name, int instance, double* out);
int idocp_ocp_get_direction_chain(idocp_ocp_t* h, const char* name, int instance, double* out);
int idocp_ocp_get_step_sizes(idocp_ocp_t* h, double* primal, double* dual);
/* P[N+1][2nv*2nv], s[N+1][2nv], K[N][nu*2nv] (nu x 2nv col-major), k[N][nu]. */
int idocp_ocp_get_riccati(idocp_ocp_t* h, int instance, double* P, double* s, double* K,
                          double* k);
/* Chain order: P[M][..], s[M][..], K[M-1][..], k[M-1][..]. */
int idocp_ocp_get_riccati_chain(idocp_ocp_t* h, int instance, double* P, double* s, double* K,
                                double* k);
/* OCPSolver::getStateFeedbackGain (ocp_solver.cpp:101-111): Kq, Kv (nu x nv col-major). */
int idocp_ocp_get_state_feedback_gain(idocp_ocp_t* h, int instance, int stage, double* Kq,
                                      double* Kv);
int idocp_ocp_dimc(const idocp_ocp_t* h);
/* OCPSolver::isCurrentSolutionFeasible (ocp_solver.cpp:216-248) and
 * ParNMPCSolver::isCurrentSolutionFeasible (parnmpc_solver.cpp:231-273): feasible[batch]
 * = 1 if the primal iterate satisfies the joint limits and the linearised (impulse)
 * friction cones on every stage of the current chain; where[batch] (may be NULL) =
 * chain position of the first offending stage (stages, then impulses, aux, lifts) or -1. */
int idocp_ocp_is_current_solution_feasible(idocp_ocp_t* h, int* feasible, int* where);
int idocp_ocp_get_constraint_data(idocp_ocp_t* h, int instance, double* slack, double* dual);
/* Condensed LQR data of one stage after the linearisation kernels (parity tests):
 * Qxx[2nv*2nv], Qxu[2nv*nu], Quu[nu*nu], A[2nv*2nv], B[2nv*nu], lx[2nv], lu[nu], Fx[2nv]. */
/* The inverse of idocp_ocp_get_lqr_stage, for every instance of the handle: writes the condensed LQR stage the backward Riccati sweep reads
 * (SplitKKTMatrix / SplitKKTResidual after condensation, split_kkt_matrix.hxx, split_kkt_residual.hxx).  Blocks column-major: Qxx [2nv x 2nv], Qxu
 * [2nv x nu], Quu [nu x nu], Fqq6 / Fqv6 [6 x 6] (the base blocks of Fqq, Fqv), Fvq, Fvv [nv x nv], Fvu [nv x nu], lx [2nv], lu [nu], Fx [2nv];
 * terminal != 0: Qxx and lx only.  For tests that run the sweep alone (idocp_ocp_launch_kernel id 2) on a problem with a known answer. */
int idocp_ocp_set_lqr_stage(idocp_ocp_t* h, int stage, int terminal, const double* Qxx, const double* Qxu, const double* Quu,
                            const double* Fqq6, const double* Fqv6, const double* Fvq, const double* Fvv, const double* Fvu,
                            const double* lx, const double* lu, const double* Fx);
/* ContactDynamicsData of a grid stage after the condensation (include/idocp/ocp/contact_dynamics_data.hxx:8-29): MJtJinv [n * n], MJtJinv_dIDCdqv
 * [n * 2 nv], MJtJinv_IDC [n], dense column-major with n = nv + dimf rows (the rows of the active contacts packed).  Returns dimf (>= 0) or an
 * error code (< 0).  For tests: M, J and the derivatives of [ID; C] follow from the three by one inverse. */
int idocp_ocp_get_contact_dynamics(idocp_ocp_t* h, int instance, int stage, double* MJtJinv, double* MJtJinv_dIDCdqv,
                                   double* MJtJinv_IDC);
/* The same by position in the chain of the current discretisation: grid, aux, lift and IMPULSE stages (there: ImpulseDynamicsForwardEulerData,
 * include/idocp/impulse/impulse_dynamics_forward_euler_data.hxx -- MJtJinv over the impulse's contacts, the blocks of [ImD; V] in place of [ID; C]). */
int idocp_ocp_get_contact_dynamics_chain(idocp_ocp_t* h, int instance, int position, double* MJtJinv, double* MJtJinv_dIDCdqv,
                                         double* MJtJinv_IDC);
int idocp_ocp_get_lqr_stage(idocp_ocp_t* h, int instance, int stage, double* Qxx, double* Qxu,
                            double* Quu, double* A, double* B, double* lx, double* lu,
                            double* Fx);
/* Diagnostic: wall-clock stamps (100 MHz ticks) taken at the phase boundaries of one
 * workgroup of the condensation kernel during the last launch; n <= 64. */
int idocp_ocp_get_profile(idocp_ocp_t* h, long long* out, int n);
/* ---- ParNMPCSolver (src/ocp/parnmpc_solver.cpp), horizons without discrete events -------------
 * Backward-Euler stages + backward correction (stage-parallel Newton) instead of the Riccati
 * sweep.  The handle is an idocp_ocp_t: contact status, setSolution, initConstraints, KKTError,
 * getters and step sizes go through the idocp_ocp_* entry points above with N stages 0..N-1
 * (stage i lives at t + (i+1) T/N; stage N-1 carries the terminal cost). */
int idocp_parnmpc_create(const idocp_model_t* model, const idocp_cost_t* cost,
                         const idocp_constraints_t* constraints, double T, int N, int batch,
                         int device, idocp_ocp_t** out);
/* ParNMPCSolver(robot, cost, constraints, T, N, max_num_impulse, nthreads) (include/idocp/ocp/parnmpc_solver.hpp) for horizons
 * with discrete events: idocp_ocp_push_back_contact_status / set_contact_points / pop_* act on the handle as for OCPSolver;
 * the chain of stages (idocp_ocp_get_chain) follows ParNMPCDiscretizer (aux / impulse / lift stages in FRONT of the grid stage
 * that follows the event). */
int idocp_parnmpc_create_hybrid(const idocp_model_t* model, const idocp_cost_t* cost,
                                const idocp_constraints_t* constraints, double T, int N, int max_num_impulse,
                                int batch, int device, idocp_ocp_t** out);
/* ParNMPCSolver::initBackwardCorrection (parnmpc_solver.cpp:66-70): aux_mat of every stage
 * = terminal cost Hessian. */
int idocp_parnmpc_init_backward_correction(idocp_ocp_t* h, double t);
/* Warm start of the correction state: BackwardCorrectionSolver::aux_mat_ of stages 0 .. nstages - 1, values[nstages][nx * nx]
 * column-major (what initBackwardCorrection fills with the terminal cost Hessian, backward_correction_solver.cpp:62-80). */
int idocp_parnmpc_set_aux_mat(idocp_ocp_t* h, int nstages, const double* values);
/* ParNMPCSolver::updateSolution (parnmpc_solver.cpp:73-103): coarseUpdate,
 * backwardCorrectionSerial / Parallel, forwardCorrectionSerial / Parallel, step sizes,
 * integrateSolution.  q[batch][nq], v[batch][nv]; line_search != 0: the filter line search on the primal step
 * (LineSearch::computeStepSize for ParNMPC, src/line_search/line_search.cpp:199-301) on a handle that holds the whole horizon,
 * discrete events included; on a shard of a horizon IDOCP_E_UNSUPPORTED -- the sharded driver's
 * idocp_parnmpc_dist_update_solution_ls evaluates every probe collectively (its hooks are installed for the duration of that call only).  idocp_ocp_line_search_eval / idocp_ocp_clear_line_search_filter
 * work on ParNMPC handles as well. */
int idocp_parnmpc_update_solution(idocp_ocp_t* h, double t, const double* q, const double* v,
                                  int line_search);
/* The same up to and including the step sizes, without integrating (the state idocp_ocp_line_search_eval probes). */
int idocp_parnmpc_compute_direction(idocp_ocp_t* h, double t, const double* q, const double* v);
int idocp_parnmpc_update_solution_device(idocp_ocp_t* h, double t, const double* d_q,
                                         const double* d_v);
/* One phase of updateSolution (bench / tests): 0 tangent RNEA, 1 backward-Euler condensation,
 * 2 KKT inverse + coarse update, 3 backward serial, 4 backward parallel, 5 forward serial,
 * 6 forward parallel + direction, 7 expansion + step sizes, 8 step-size reduction,
 * 9 dual expansion + integration. */
int idocp_parnmpc_launch_phase(idocp_ocp_t* h, int phase, const double* d_q, const double* d_v);
/* ParNMPCSolver::computeKKTResidual (parnmpc_solver.cpp:200-206); read with idocp_ocp_kkt_error. */
int idocp_parnmpc_compute_kkt_residual(idocp_ocp_t* h, double t, const double* q, const double* v);
/* Horizon sharding (BASELINE.json configs[3]; protocol in idocp_amd/parnmpc_dist.py): one handle per
 * process holds the stages [stage_offset, stage_offset + N) of a longer horizon (T = N * dt of the
 * shard).  Halos are packed into / unpacked from caller-owned DEVICE buffers d_buf[batch][size] that
 * the caller moves between ranks (RCCL send / recv); kinds: 0 state_last (q, v -> right),
 * 1 costate_first (lmd, gmm, q -> left), 2 aux_first (-> left), 3 bwd_first (corrected lmd, gmm ->
 * left), 4 fwd_last (corrected q, v -> right), 5 aux_all (initBackwardCorrection broadcast). */
int idocp_parnmpc_create_shard(const idocp_model_t* model, const idocp_cost_t* cost,
                               const idocp_constraints_t* constraints, double T, int N,
                               int stage_offset, int has_terminal, int has_prev, int batch,
                               int device, idocp_ocp_t** out);
/* The same for a horizon WITH discrete events: every rank creates its handle with the whole horizon (T, N, max_num_impulse) and
 * pushes the whole contact sequence; the handle keeps the grid stages [stage_begin, stage_end) of the ParNMPCDiscretizer chain
 * and the event stages in front of each of them.  Halos as above. */
int idocp_parnmpc_create_hybrid_shard(const idocp_model_t* model, const idocp_cost_t* cost,
                                      const idocp_constraints_t* constraints, double T, int N, int max_num_impulse,
                                      int stage_begin, int stage_end, int batch, int device, idocp_ocp_t** out);
int idocp_parnmpc_halo_size(int kind);
int idocp_parnmpc_export_halo(idocp_ocp_t* h, int kind, double* d_buf);
int idocp_parnmpc_import_halo(idocp_ocp_t* h, int kind, const double* d_buf);
/* The same pack / unpack kernels enqueued on the handle's stream WITHOUT the host waiting for them: for a
 * stream-ordered transport (the C++ driver below: pack -> ncclSend, ncclRecv -> unpack on one stream). */
int idocp_parnmpc_export_halo_async(idocp_ocp_t* h, int kind, double* d_buf);
int idocp_parnmpc_import_halo_async(idocp_ocp_t* h, int kind, const double* d_buf);
/* ---- multi-GPU driver of the sharded horizon, in C++ over RCCL (idocp_amd/csrc/parnmpc_dist.hip) ------------------------
 * Counterpart of BackwardCorrectionSolver (src/ocp/backward_correction_solver.cpp:255-366) for one process per GPU: every rank
 * creates its shard (idocp_parnmpc_create_shard / _create_hybrid_shard), a communicator, attaches one to the other and then calls
 * the idocp_parnmpc_dist_* entry points collectively.  The halo exchange (point-to-point ncclSend / ncclRecv with the two
 * neighbours, all-reduce of step sizes and KKT error) is enqueued on the shard's stream between its kernels; nothing in an
 * iteration synchronises with the host.  The unique id is made on rank 0 and distributed by the caller (MPI, a file, a store). */
#define IDOCP_COMM_ID_BYTES 128
typedef struct idocp_comm idocp_comm_t;
int idocp_comm_get_unique_id(void* id /* IDOCP_COMM_ID_BYTES */);
int idocp_comm_init_rank(const void* id, int rank, int world, int device, idocp_comm_t** out);
/* `world` endpoints on ONE GPU in one process (out[world]; one host thread per endpoint): the transport of the single-GPU test
 * of this driver, not a product path. */
int idocp_comm_init_local(int world, int device, idocp_comm_t** out);
/* Host-staged transport through caller-supplied functions (an MPI or gloo communicator the caller already owns; the cross-process test of
 * the driver on a one-GPU box).  The driver makes the SAME calls in the same order and grouping as on RCCL
 * (src/ocp/backward_correction_solver.cpp:255-366 is the computation being distributed):
 *   send / recv   n doubles to / from rank `peer`.  in_group != 0: between group_start and group_end the call may only POST the transfer
 *                 (buffers stay valid until group_end returns; ncclGroupStart / ncclGroupEnd semantics: a send and its matching receive of the
 *                 same group never block each other); in_group == 0: complete before returning
 *   group_start / group_end   optional (NULL: every send / recv must then be able to complete on its own)
 *   allreduce     in place on n doubles, op 0 = sum, 1 = min;   broadcast   n doubles from rank `root`
 *   destroy       optional, called by idocp_comm_destroy
 * Every function returns 0 on success. */
typedef struct idocp_comm_callbacks {
  void* ctx;
  int (*send)(void* ctx, const double* buf, unsigned long n, int peer, int in_group);
  int (*recv)(void* ctx, double* buf, unsigned long n, int peer, int in_group);
  int (*group_start)(void* ctx);
  int (*group_end)(void* ctx);
  int (*allreduce)(void* ctx, double* buf, unsigned long n, int op);
  int (*broadcast)(void* ctx, double* buf, unsigned long n, int root);
  void (*destroy)(void* ctx);
} idocp_comm_callbacks_t;
int idocp_comm_init_callbacks(int rank, int world, int device, const idocp_comm_callbacks_t* cb, idocp_comm_t** out);
void idocp_comm_destroy(idocp_comm_t* c);
int idocp_comm_rank(const idocp_comm_t* c);
int idocp_comm_world(const idocp_comm_t* c);
int idocp_parnmpc_dist_attach(idocp_ocp_t* shard, idocp_comm_t* comm);
int idocp_parnmpc_dist_detach(idocp_ocp_t* shard);
/* rank 0: the measured state q[batch][nq], v[batch][nv] (host buffers) */
int idocp_parnmpc_dist_set_initial_state(idocp_ocp_t* shard, const double* q, const double* v, int nq, int nv);
int idocp_parnmpc_dist_init_backward_correction(idocp_ocp_t* shard, double t);
/* One iteration of the whole horizon; returns when everything is enqueued (idocp_ocp_synchronize before reading results). */
int idocp_parnmpc_dist_update_solution(idocp_ocp_t* shard, double t);
/* ... with the filter line search on the primal step (ParNMPCSolver::updateSolution(t, q, v, true)): collective like the call above;
 * host-synchronous (one device-to-host copy of the all-reduced sums per probe, like the single-handle path). */
int idocp_parnmpc_dist_update_solution_ls(idocp_ocp_t* shard, double t);
/* KKT error of the whole horizon on every rank, kkt_error[batch] (host). */
int idocp_parnmpc_dist_kkt_error(idocp_ocp_t* shard, double t, double* kkt_error);
int idocp_ocp_batch(idocp_ocp_t* h);
/* Storage precision of the Riccati factorisation P, s of OCPSolver: bits = 64 (default) or 32 (every entry of the cost-to-go rounded to
 * single precision when a stage of the backward sweep stores it; the arithmetic stays FP64).  BASELINE.json configs[4]'s tolerance study
 * on the device (tests/test_hybrid_gpu.py::test_configs4_grid_and_fp32_storage); FP64 is the product's precision. */
int idocp_ocp_set_riccati_storage(idocp_ocp_t* h, int bits);
/* dimensions of the configuration / velocity the handle's model has (q[batch][nq], v[batch][nv] of the update entries) */
int idocp_ocp_state_dims(idocp_ocp_t* h, int* nq, int* nv);
/* Test switches of the transport (tests/test_rccl_gpu.py; the one-GPU box cannot run more than one rank):
 *  idocp_comm_set_force_collectives   world == 1: issue all-reduce / broadcast through RCCL anyway instead of skipping them
 *  idocp_parnmpc_dist_transport_selftest   world 1: grouped ncclSend / ncclRecv of every halo kind to this very rank, all-reduce (sum, min)
 *                                          and broadcast on the shard's stream.  world > 1 (COLLECTIVE: every rank calls it; bench.py does
 *                                          before the timed region): every halo kind to the right neighbour and from the left one in one
 *                                          group, then the other way round -- the grouping of the driver's boundary exchange
 *                                          (backward_correction_solver.cpp:255-366 is what those halos feed) --, a pattern that names kind,
 *                                          element and sending rank; all-reduce (sum, min) and broadcast against their closed forms.
 *                                          *max_abs_diff = deviation of what came back (0 expected)
 *  idocp_comm_info                         what the communicator reports about itself: ncclCommCount, ncclCommUserRank, ncclGetVersion
 *                                          (any pointer may be NULL; transport: 1 RCCL, 0 the in-process test transport) */
int idocp_comm_set_force_collectives(idocp_comm_t* c, int on);
int idocp_parnmpc_dist_transport_selftest(idocp_ocp_t* shard, double* max_abs_diff);
int idocp_comm_info(const idocp_comm_t* c, int* nranks, int* user_rank, int* rccl_version, int* transport);
/* Deep copy of a solver handle (the reference's solver classes are copyable): same configuration, device records, contact
 * sequence and discretisation. */
int idocp_ocp_clone(idocp_ocp_t* src, idocp_ocp_t** out);
/* Replace the cost of a solver (OCPSolver / ParNMPCSolver handles).  The reference's solvers share the
 * CostFunction with the driver (ocp_solver.hpp:37-39: shared_ptr), so references and weights changed
 * between two updateSolution calls take effect at the next one; here the cost is copied at creation and
 * this call is how an MPC loop moves its goal.  A task-space cost cannot be added or removed
 * (IDOCP_E_UNSUPPORTED).  Takes effect with the next call that discretises the horizon. */
int idocp_ocp_set_cost(idocp_ocp_t* h, const idocp_cost_t* cost);
/* Device pointers of the state in front of the first stage (q[batch][nq], v[batch][nv]) and of
 * the step sizes ([batch][2]: primal, dual) -- the latter is all-reduced (min) between phases 8 and 9. */
int idocp_parnmpc_prev_state(idocp_ocp_t* h, double** d_q, double** d_v);
int idocp_parnmpc_step_sizes_device(idocp_ocp_t* h, double** d_steps);
/* Filter line search of ParNMPCSolver on a sharded horizon (round 4; src/line_search/line_search.cpp:199-301 evaluates every stage against
 * the trial iterate of its predecessor -- across a cut that is the left neighbour's trial iterate -- and sums over the horizon): the pieces
 * the sharded driver puts together.  idocp_parnmpc_set_line_search_hooks installs the two collective steps of one probe (pre: after the
 * trial iterate is formed; post: after the shard's cost / violation sums are formed); idocp_parnmpc_trial_halo_async packs / unpacks the
 * state_last halo of the trial iterate; idocp_parnmpc_merit_device gives the sums [batch][2] to all-reduce; idocp_parnmpc_line_search runs
 * LineSearch::computeStepSize on the direction of phases 0 .. 8 (every shard calls it: the filters evolve identically on identical sums).
 * Application code calls idocp_parnmpc_dist_update_solution_ls. */
int idocp_parnmpc_set_line_search_hooks(idocp_ocp_t* h, int (*pre)(idocp_ocp_t*), int (*post)(idocp_ocp_t*));
int idocp_parnmpc_trial_halo_async(idocp_ocp_t* h, int do_import, double* d_buf);
int idocp_parnmpc_merit_device(idocp_ocp_t* h, double** d_merit);
int idocp_parnmpc_line_search(idocp_ocp_t* h);
/* Discretise at time t (stage references, constraint levels) before launching phases by hand. */
int idocp_parnmpc_discretize(idocp_ocp_t* h, double t);
/* Squared KKT error of this shard's stages, d_err2[batch] in device memory (summed over the ranks by
 * the caller); the state in front of the first stage is the one held on the device. */
int idocp_parnmpc_kkt_error_squared_device(idocp_ocp_t* h, double t, double* d_err2);
/* Device-to-device copy (moving halos between the library's buffers and the caller's). */
int idocp_device_copy(void* d_dst, const void* d_src, unsigned long nbytes);

/* One kernel launch: 0 = tangent RNEA, 1 = condense, 2 = backward Riccati,
 * 3 = forward Riccati, 4 = expand primal, 5 = step-size reduction,
 * 6 = expand dual + integrate; 7 and 8 = the two halves of 1 (7: the nominal
 * rigid-body sweeps and the rows of the external terms, 8: the condensation
 * launches), so that a profiler of the caller can bracket them apart.
 * Since round 5 the forward sweep of a BATCH of instances expands as it walks (riccati_recursion_solver.cpp:129-251 in one kernel): id 3
 * is then the whole of 3 + 4 + 5 and ids 4, 5 launch nothing; idocp_ocp_fused_forward(h) says which form the handle runs (small batches
 * keep the three kernels: latency mode; IDOCP_FUSED_FORWARD=0 / 1 forces one, IDOCP_FUSED_FORWARD_MIN_BATCH moves the threshold). */
int idocp_ocp_launch_kernel(idocp_ocp_t* h, int kernel_id, const double* d_q,
                            const double* d_v);
int idocp_ocp_fused_forward(idocp_ocp_t* h);
/* mode -1: by batch size (default); 0: S4 + K6 + reduction; 1: the fused forward sweep -- per handle (tuning / the parity tests of both forms) */
int idocp_ocp_set_fused_forward(idocp_ocp_t* h, int mode);

/* The backward Riccati sweep S3 (riccati_recursion_solver.cpp:48-107) has two forms: one wavefront per instance with P in registers (batches:
 * the throughput form) and eight wavefronts per instance with P staged in LDS (a handful of instances: the latency form, 8 % faster at batch
 * 1).  idocp_ocp_riccati_sweep(h): 1 when the handle runs the latency form.  set: mode -1 by batch size (default: batch <= 16 ->
 * latency form; IDOCP_RICCATI_WIDE_MAX_BATCH moves the threshold), 0 throughput form, 1 latency form -- per handle, copied by clone. */
int idocp_ocp_riccati_sweep(idocp_ocp_t* h);
int idocp_ocp_set_riccati_sweep(idocp_ocp_t* h, int mode);

const char* idocp_last_error(void);
const char* idocp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IDOCP_HIP_H_ */
