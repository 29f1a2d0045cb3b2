// K5a -- inverse dynamics + contact (Baumgarte) constraint and ALL their
// derivatives for one stage of a legged robot, one tangent seed per lane.
//
// Replaces, for every stage of the horizon (OCPLinearizer::runParallel,
// include/idocp/ocp/ocp_linearizer.hxx:113-228), the rigid-body part of
// ContactDynamics::linearizeContactDynamics (include/idocp/ocp/contact_dynamics.hxx:48-102):
//   robot.updateKinematics(q,v,a); robot.setContactForces; robot.RNEA; robot.RNEADerivatives;
//   robot.computeBaumgarteResidual / computeBaumgarteDerivatives
// (include/idocp/robot/robot.hxx:193-203,237-279,408-500; point_contact.hxx:15-144).
//
// One wavefront per stage.  Lane (kind, k), kind in {q, v, a}, k < NV, carries the
// tangent d/d(kind_k) through a body-frame Newton-Euler sweep over the tree
// (floating base + NL legs of LJ joints) in registers; at each foot the same
// dual numbers give the tangents of the contact frame's position, spatial
// velocity and acceleration, from which the Baumgarte derivative is assembled
// WITH THE REFERENCE'S OWN FORMULA (point_contact.hxx:117-143 adds
// skew(v_lin) * d(omega); an exact derivative would subtract it) so that the
// Newton direction matches the reference, not just the mathematics.
// Output: lin record = [dID;dC]/d(q,v), dID/da (= M), dC/da (= J), [ID; C].
//
// Impulse stages (ImpulseDynamicsForwardEuler::linearizeImpulseDynamics,
// include/idocp/impulse/impulse_dynamics_forward_euler.hxx:18-58; robot.hxx:283-320, 505-541) run the same
// sweep twice: pass 0 with (v, a, g) = (0, dv, 0) gives ImD = rnea_impulse(q, dv), dImD/dq and dImD/ddv;
// pass 1 with the velocity v + dv gives the contact-velocity constraint C = v_foot and dC/dq, dC/dv
// (= dC/ddv = J, emitted by the v- AND the a-seed lanes).
#include <hip/hip_runtime.h>

#include "dev_dense.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

// quaternion (x y z w) -> rotation, plain doubles
__device__ __forceinline__ void quatToRot(const double* __restrict__ qt, double* R) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// IMPULSE = false: regular (stage / aux / lift) stages, one pass; launched over every non-terminal stage of the chain,
// impulse stages return at once.  IMPULSE = true: the impulse stages only (two passes), launched over the list
// B.impulse_pos of their chain positions.
//
// Work decomposition.  A tangent seed on a joint of leg L only perturbs leg L and (through the transmitted force) the base
// rows; a seed on a base coordinate perturbs everything.  So the unit of work is an ITEM = (seed, leg): 18 base seeds x NL
// legs + 9 joint seeds of each leg = 27 NL items (108 for a quadruped), each ONE leg sweep with one tangent -- half the
// 54 seeds x 4 legs of the dense scheme.  A wavefront takes the items of a stage in ceil(27 NL / 64) rounds; the base rows
// of a base seed are summed from its NL items (and the base's own inertial term) in a fixed order afterwards.
template <typename D, bool IMPULSE>
__global__ __launch_bounds__(64) void ocp_rnea_kernel(OcpBuffers B, int nlist) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NL = D::NL, LJ = D::LJ, NF = D::NF, NVF = D::NVF, NX = D::NX;
  constexpr int IPL = 18 + 3 * LJ, NITEMS = NL * IPL, ROUNDS = (NITEMS + 63) / 64;      // items per leg, items, rounds
  typedef Dual T;
  __shared__ double s_out[3 * NV][NVF];   // column of each seed: rows [dID (NV) ; dC (NF)]
  __shared__ double s_idc[NVF];           // nominal [ID ; C]
  __shared__ double s_cs[D::NU][2];       // cos / sin of the leg joint angles
  __shared__ double s_v[NV], s_a[NV];     // velocity / acceleration inputs of the current pass
  __shared__ double s_bt[NITEMS][6];      // tangent of the force each item's leg transmits to the base
  __shared__ double s_bown[18][6];        // tangent of the base's own inertial force, per base seed
  __shared__ double s_bn[NL + 1][6];      // nominal base force: own, then per leg
  const DevModel* __restrict__ m = B.model;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;                       // one non-terminal stage of the chain per wavefront
  const int per = IMPULSE ? nlist : (M - 1);
  const long b = unit / per;
  const int pos = IMPULSE ? B.impulse_pos[(int)(unit - b * per)] : (int)(unit - b * per);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  constexpr bool impulse = IMPULSE;
  if (!IMPULSE && nd->kind == 1) return;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ q = s + L::S_Q;
  if (lane < D::NU) {
    double sj, cj;
    sincos(q[7 + lane], &sj, &cj);
    s_cs[lane][0] = cj; s_cs[lane][1] = sj;
  }
  // zero the C rows of inactive contacts / rows a seed does not reach
  for (int r = lane; r < 3 * NV * NVF; r += 64) (&s_out[0][0])[r] = 0.0;
  if (lane < NVF) s_idc[lane] = 0.0;
  const double gz = impulse ? 0.0 : m->gravity[2];
  const double wv = 2.0 / P->baumgarte_time_step, wp = 1.0 / (P->baumgarte_time_step * P->baumgarte_time_step);
  double Rn[9];
  quatToRot(q + 3, Rn);

  const int npass = impulse ? 2 : 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
    const bool do_dyn = (pass == 0), do_con = (!impulse) || (pass == 1);
    waveLdsSync();
    if (lane < NV) {
      const double vin = s[L::S_V + lane], ain = s[L::S_A + lane];
      // kinematic pass of an impulse stage: post-impact velocity = v + dv (forward Euler, impulse_split_ocp.hxx:47) or v itself
      // (backward Euler: v IS the post-impact velocity, impulse_split_parnmpc.hxx:43)
      s_v[lane] = impulse ? (pass == 0 ? 0.0 : (P->backward_euler ? vin : vin + ain)) : vin;
      s_a[lane] = impulse ? (pass == 0 ? ain : 0.0) : ain;
    }
    waveLdsSync();
#pragma unroll 1
    for (int round = 0; round < ROUNDS; ++round) {
      const int item = round * 64 + lane;
      if (item >= NITEMS) continue;
      const int leg = item / IPL, j0 = item - leg * IPL;
      const bool base_seed = j0 < 18;
      const int kind = base_seed ? j0 / 6 : (j0 - 18) / LJ;                           // 0: q, 1: v, 2: a
      const int k = base_seed ? j0 - 6 * kind : 6 + leg * LJ + (j0 - 18 - LJ * kind);   // velocity index of the seed
      double* __restrict__ col = &s_out[kind * NV + k][0];
      // velocity seeds: the v seeds; in the kinematic pass of an impulse stage also the a seeds (dC/ddv = dC/dv)
      const bool vseed = (kind == 1) || (impulse && pass == 1 && kind == 2);
      const bool aseed = (kind == 2) && !(impulse && pass == 1);
      // ---- base (free-flyer): placement, velocity, acceleration (with the gravity field) ----
      // tangent of q (+) e_k on the manifold: dp = R e_lin, dR = R skew(e_ang)   (local-frame perturbation)
      const double el[3] = {(kind == 0 && k == 0) ? 1.0 : 0.0, (kind == 0 && k == 1) ? 1.0 : 0.0, (kind == 0 && k == 2) ? 1.0 : 0.0};
      const double ea[3] = {(kind == 0 && k == 3) ? 1.0 : 0.0, (kind == 0 && k == 4) ? 1.0 : 0.0, (kind == 0 && k == 5) ? 1.0 : 0.0};
      Mat3<T> Rw;                                       // world pose of the current frame (starts at the base)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        Rw.m[3 * r + 0] = T(Rn[3 * r + 0], Rn[3 * r + 1] * ea[2] - Rn[3 * r + 2] * ea[1]);
        Rw.m[3 * r + 1] = T(Rn[3 * r + 1], Rn[3 * r + 2] * ea[0] - Rn[3 * r + 0] * ea[2]);
        Rw.m[3 * r + 2] = T(Rn[3 * r + 2], Rn[3 * r + 0] * ea[1] - Rn[3 * r + 1] * ea[0]);
      }
      Vec3<T> pw = mk<T>(T(q[0], Rn[0] * el[0] + Rn[1] * el[1] + Rn[2] * el[2]), T(q[1], Rn[3] * el[0] + Rn[4] * el[1] + Rn[5] * el[2]),
                         T(q[2], Rn[6] * el[0] + Rn[7] * el[1] + Rn[8] * el[2]));
      auto seedV = [&](int idx) { return T(s_v[idx], (vseed && k == idx) ? 1.0 : 0.0); };
      auto seedA = [&](int idx) { return T(s_a[idx], (aseed && k == idx) ? 1.0 : 0.0); };
      Vec3<T> v = mk<T>(seedV(0), seedV(1), seedV(2)), w = mk<T>(seedV(3), seedV(4), seedV(5));
      // a_gf = a_joint + R^T (0, 0, -g_z)  (base acceleration in the gravity field; v x vJ = 0 for the root)
      Vec3<T> bl = mk<T>(seedA(0) - gz * Rw.m[6], seedA(1) - gz * Rw.m[7], seedA(2) - gz * Rw.m[8]);
      Vec3<T> bw = mk<T>(seedA(3), seedA(4), seedA(5));
      if (do_dyn && leg == 0 && base_seed) {
        // the base's own inertial force: once per base seed (and once for the nominal value)
        Vec3<T> hl, hn, f, n;
        inertiaMul<T>(m, 0, v, w, hl, hn);
        inertiaMul<T>(m, 0, bl, bw, f, n);
        const Vec3<T> Fbl = f + cross(w, hl);
        const Vec3<T> Fbn = n + cross(w, hn) + cross(v, hl);
        double* o = &s_bown[j0][0];
        o[0] = Fbl.x.d; o[1] = Fbl.y.d; o[2] = Fbl.z.d; o[3] = Fbn.x.d; o[4] = Fbn.y.d; o[5] = Fbn.z.d;
        if (j0 == 0) { double* on = &s_bn[0][0]; on[0] = Fbl.x.v; on[1] = Fbl.y.v; on[2] = Fbl.z.v; on[3] = Fbn.x.v; on[4] = Fbn.y.v; on[5] = Fbn.z.v; }
      }
      // ---- the leg of this item ----
#pragma unroll 1
      for (int j = 0; j < LJ; ++j) {
        const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j, ci = leg * LJ + j;
        const bool mine = (k == dof);
        const T cqi(s_cs[ci][0], (mine && kind == 0) ? -s_cs[ci][1] : 0.0);
        const T sqi(s_cs[ci][1], (mine && kind == 0) ? s_cs[ci][0] : 0.0);
        const T qdi(s_v[dof], (mine && vseed) ? 1.0 : 0.0);
        const T qddi(s_a[dof], (mine && aseed) ? 1.0 : 0.0);
        Mat3<T> R;
        revoluteRotation<T>(m->R[ji], m->axis[ji], cqi, sqi, R);
        const double* p = m->p[ji];
        const double* u = m->axis[ji];
        pw = pw + mul(Rw, mk<T>(T(p[0]), T(p[1]), T(p[2])));
        {
          Mat3<T> Rn2;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rn2.m[3 * r + c] = Rw.m[3 * r] * R.m[c] + Rw.m[3 * r + 1] * R.m[3 + c] + Rw.m[3 * r + 2] * R.m[6 + c];
          Rw = Rn2;
        }
        const Vec3<T> wc = mulT(R, w);
        const Vec3<T> vc = mulT(R, v + crossVC<T>(w, p));
        const Vec3<T> bwc = mulT(R, bw);
        const Vec3<T> blc = mulT(R, bl + crossVC<T>(bw, p));
        const Vec3<T> vJ = mk<T>(u[0] * qdi, u[1] * qdi, u[2] * qdi);
        w = wc + vJ;
        v = vc;
        bw = bwc + mk<T>(u[0] * qddi, u[1] * qddi, u[2] * qddi) + cross(w, vJ);
        bl = blc + cross(v, vJ);
      }
      // ---- contact frame at the foot (tip joint of this leg) ----
      Vec3<T> fel = mk<T>(T(0.0), T(0.0), T(0.0)), fen = fel;     // contact force as a spatial force on the tip joint
      if (nd->active[leg]) {
        const double* Rc = P->contact_R[leg];
        const double* pc = P->contact_p[leg];
        const int row = NV + nd->row_of[leg];
        if (do_con) {
          // frame spatial velocity / acceleration (acceleration WITHOUT gravity: a = a_gf + R_w^T g)
          const Vec3<T> al_ng = mk<T>(bl.x + gz * Rw.m[6], bl.y + gz * Rw.m[7], bl.z + gz * Rw.m[8]);
          const Vec3<T> vj = v + crossVC<T>(w, pc);
          const Vec3<T> aj = al_ng + crossVC<T>(bw, pc);
          auto rotT = [&](Vec3<T> x) {
            return mk<T>(Rc[0] * x.x + Rc[3] * x.y + Rc[6] * x.z, Rc[1] * x.x + Rc[4] * x.y + Rc[7] * x.z, Rc[2] * x.x + Rc[5] * x.y + Rc[8] * x.z);
          };
          const Vec3<T> fv = rotT(vj), fw = rotT(w), fa = rotT(aj);
          const Vec3<T> pf = pw + mul(Rw, mk<T>(T(pc[0]), T(pc[1]), T(pc[2])));
          double cx, cy, cz, dx, dy, dz;
          if (impulse) {
            // contact-velocity constraint (point_contact.hxx:145-175): LOCAL linear velocity of the frame
            cx = fv.x.v; cy = fv.y.v; cz = fv.z.v;
            dx = fv.x.d; dy = fv.y.d; dz = fv.z.d;
          } else {
            // nominal residual (point_contact.hxx:67-87)
            cx = fa.x.v + (fw.y.v * fv.z.v - fw.z.v * fv.y.v) + wv * fv.x.v + wp * (pf.x.v - nd->contact_point[leg][0]);
            cy = fa.y.v + (fw.z.v * fv.x.v - fw.x.v * fv.z.v) + wv * fv.y.v + wp * (pf.y.v - nd->contact_point[leg][1]);
            cz = fa.z.v + (fw.x.v * fv.y.v - fw.y.v * fv.x.v) + wv * fv.z.v + wp * (pf.z.v - nd->contact_point[leg][2]);
            // derivative column (point_contact.hxx:117-143): da_lin + skew(w) dv_lin + skew(v_lin) dw + (2/D) dv_lin + (1/D^2) dp_world
            dx = fa.x.d + (fw.y.v * fv.z.d - fw.z.v * fv.y.d) + (fv.y.v * fw.z.d - fv.z.v * fw.y.d) + wv * fv.x.d + wp * pf.x.d;
            dy = fa.y.d + (fw.z.v * fv.x.d - fw.x.v * fv.z.d) + (fv.z.v * fw.x.d - fv.x.v * fw.z.d) + wv * fv.y.d + wp * pf.y.d;
            dz = fa.z.d + (fw.x.v * fv.y.d - fw.y.v * fv.x.d) + (fv.x.v * fw.y.d - fv.y.v * fw.x.d) + wv * fv.z.d + wp * pf.z.d;
          }
          col[row] = dx; col[row + 1] = dy; col[row + 2] = dz;
          if (j0 == 0) { s_idc[row] = cx; s_idc[row + 1] = cy; s_idc[row + 2] = cz; }
        }
        // PointContact::computeJointForceFromContactForce (point_contact.hxx:15-20): jXf.act(Force(f, 0))
        const double* f = s + L::S_F + 3 * leg;
        const double fx = Rc[0] * f[0] + Rc[1] * f[1] + Rc[2] * f[2], fy = Rc[3] * f[0] + Rc[4] * f[1] + Rc[5] * f[2],
                     fz = Rc[6] * f[0] + Rc[7] * f[1] + Rc[8] * f[2];
        fel = mk<T>(T(fx), T(fy), T(fz));
        fen = mk<T>(T(pc[1] * fz - pc[2] * fy), T(pc[2] * fx - pc[0] * fz), T(pc[0] * fy - pc[1] * fx));
      }
      if (!do_dyn) continue;                          // kinematic pass of an impulse stage: no forces
      // ---- inward sweep: accumulate forces, emit tau, undo the kinematic steps ----
      Vec3<T> Fl = mk<T>(T(0.0), T(0.0), T(0.0)) - fel, Fn = mk<T>(T(0.0), T(0.0), T(0.0)) - fen;
#pragma unroll 1
      for (int j = LJ - 1; j >= 0; --j) {
        const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j, ci = leg * LJ + j;
        const double* u = m->axis[ji];
        Vec3<T> hl, hn, f, n;
        inertiaMul<T>(m, ji, v, w, hl, hn);
        inertiaMul<T>(m, ji, bl, bw, f, n);
        Fl = Fl + f + cross(w, hl);
        Fn = Fn + n + cross(w, hn) + cross(v, hl);
        const T ti = u[0] * Fn.x + u[1] * Fn.y + u[2] * Fn.z;
        col[dof] = ti.d;
        if (j0 == 0) s_idc[dof] = ti.v;
        const bool mine = (k == dof);
        const T cqi(s_cs[ci][0], (mine && kind == 0) ? -s_cs[ci][1] : 0.0);
        const T sqi(s_cs[ci][1], (mine && kind == 0) ? s_cs[ci][0] : 0.0);
        Mat3<T> R;
        revoluteRotation<T>(m->R[ji], m->axis[ji], cqi, sqi, R);
        const double* p = m->p[ji];
        const Vec3<T> Rf = mul(R, Fl);
        Fn = mul(R, Fn) + crossC<T>(p, Rf);
        Fl = Rf;
        if (j > 0) {
          const T qdi(s_v[dof], (mine && vseed) ? 1.0 : 0.0);
          const T qddi(s_a[dof], (mine && aseed) ? 1.0 : 0.0);
          const Vec3<T> vJ = mk<T>(u[0] * qdi, u[1] * qdi, u[2] * qdi);
          const Vec3<T> bwc = bw - mk<T>(u[0] * qddi, u[1] * qddi, u[2] * qddi) - cross(w, vJ);
          const Vec3<T> blc = bl - cross(v, vJ);
          const Vec3<T> wc = w - vJ;
          w = mul(R, wc);
          v = mul(R, v) - crossVC<T>(w, p);
          bw = mul(R, bwc);
          bl = mul(R, blc) - crossVC<T>(bw, p);
        }
      }
      {
        double* o = &s_bt[item][0];
        o[0] = Fl.x.d; o[1] = Fl.y.d; o[2] = Fl.z.d; o[3] = Fn.x.d; o[4] = Fn.y.d; o[5] = Fn.z.d;
        if (j0 == 0) { double* on = &s_bn[1 + leg][0]; on[0] = Fl.x.v; on[1] = Fl.y.v; on[2] = Fl.z.v; on[3] = Fn.x.v; on[4] = Fn.y.v; on[5] = Fn.z.v; }
      }
    }
    if (do_dyn) {
      waveLdsSync();
      // base rows: tau[0:6] = total spatial force on the base (S = identity): own term + legs, in leg order
      for (int e = lane; e < 3 * NV * 6; e += 64) {
        const int c = e / 6, r = e - 6 * c, kind = c / NV, k = c - kind * NV;
        double acc;
        if (k < 6) {
          acc = s_bown[kind * 6 + k][r];
          for (int leg = 0; leg < NL; ++leg) acc += s_bt[leg * IPL + kind * 6 + k][r];
        } else {
          const int leg = (k - 6) / LJ;
          acc = s_bt[leg * IPL + 18 + LJ * kind + (k - 6 - leg * LJ)][r];
        }
        s_out[c][r] = acc;
      }
      if (lane < 6) {
        double acc = s_bn[0][lane];
        for (int leg = 0; leg < NL; ++leg) acc += s_bn[1 + leg][lane];
        s_idc[lane] = acc;
      }
    }
  }
  waveLdsSync();
  // ---- coalesced write of the lin record ----
  double* __restrict__ lin = B.lin + rec * L::LIN;
  for (int e = lane; e < NVF * NX; e += 64) lin[L::L_DIDC + e] = (&s_out[0][0])[e];          // q and v seeds: 2 NV columns of NVF
  for (int e = lane; e < NV * NV; e += 64) {                                                 // a seeds, rows 0..NV-1 = M
    const int c = e / NV, r = e - c * NV;
    lin[L::L_M + e] = s_out[2 * NV + c][r];
  }
  for (int e = lane; e < NF * NV; e += 64) {                                                 // a seeds, rows NV.. = J
    const int c = e / NF, r = e - c * NF;
    lin[L::L_J + e] = s_out[2 * NV + c][NV + r];
  }
  // ID - u on the actuated rows (contact_dynamics.hxx:88); the impulse stage has no torques
  if (lane < NVF) lin[L::L_IDC + lane] = s_idc[lane] - ((nd->has_u && lane >= 6 && lane < NV) ? s[L::S_U + lane - 6] : 0.0);
}

// The regular stages no longer have a kernel of their own: the condensation kernel runs the tangent-only sweep of
// dev_rnea_tangent.hpp in its first phase (ocp_condense_kernel.hip, phase A) and the lin record is only written for the
// impulse stages (consumed by the general instantiation of K5b and by K9i).
template <typename D>
void OcpLaunch<D>::rnea(const OcpBuffers& B, long batch, int M, int n_impulse, hipStream_t st) {
  (void)M;
  if (B.n_impulse_fe > 0) return;      // forward-Euler chains (OCPSolver): the impulse stages have nominal records and tangent items like every other stage (round 3)
  if (n_impulse > 0) hipLaunchKernelGGL((ocp_rnea_kernel<D, true>), dim3((unsigned)(batch * n_impulse)), dim3(64), 0, st, B, n_impulse);
}

template void OcpLaunch<LeggedDims<4, 3>>::rnea(const OcpBuffers&, long, int, int, hipStream_t);

}  // namespace idocp_dev
