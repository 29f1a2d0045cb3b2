// K5n -- the NOMINAL Newton-Euler sweeps of every stage, one lane per (stage, leg).
//
// Replaces, for the values (not the derivatives), the rigid-body calls at the head of ContactDynamics::linearizeContactDynamics
// (include/idocp/ocp/contact_dynamics.hxx:48-60): robot.updateKinematics(q, v, a), setContactForces, RNEA and the nominal part of
// computeBaumgarteResidual (include/idocp/robot/robot.hxx:85-140, 193-203, 237-260, 444-465; point_contact.hxx:15-20, 67-87).
//
// Rounds 1 / 2 ran these sweeps INSIDE the condensation kernel: five active lanes (four legs + the base) of a 256-thread workgroup
// walking a ~1500-instruction dependent chain while the stage's 35 kB of LDS and four wavefronts waited -- 5.1 of the 28.5 us a
// stage spent in that kernel, and a fifth of its vector-instruction issue for 5 / 64 of a wavefront's lanes.  Here a wavefront
// carries 64 STAGES through the same chain: blockIdx.y selects the task (motion sweep of leg y, the base body, pose sweep of a
// leg), so a wavefront never diverges and the model constants of its leg are scalar loads.  The sweeps leave, per stage, the
// record the tangent items of the condensation kernel read as LDS broadcasts (OcpLayout::O_JOINT .. O_IDC, 5.2 kB, copied into
// its LDS scratch in 16-byte pieces; dev_rnea_tangent.hpp RneaScratch).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "dev_rnea_tangent.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

namespace {
using namespace rt;
__device__ __forceinline__ void st3g(double* __restrict__ p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
}  // namespace

// XYY: the joint axes of the legs are known at compile time (OcpBuffers::leg_axes_xyy, dev_rnea_tangent.hpp JointFrame): the sweeps use
// the same reduced rotations / cross products as the tangent items of the condensation kernel.
//
// Stores.  A lane owns a stage, and the records of neighbouring lanes are 5 kB apart: written straight from the registers every
// store instruction would touch 64 different lines with 8 bytes each (the first version: 0.40 ms, bound by write requests).  The
// joint records -- four fifths of the bytes -- therefore go through a transposition in LDS ([lane][field] -> runs of consecutive
// fields of one stage over consecutive lanes), twice per joint: the outward part (30 doubles) and the inward part (12 doubles).
// IMP: the IMPULSE stages of a forward-Euler chain (launched over B.impulse_pos; ImpulseDynamicsForwardEuler::linearizeImpulseDynamics,
// impulse_dynamics_forward_euler.hxx:18-58; robot.hxx:283-320, 505-541).  Their dynamics ImD = rnea_impulse(q, dv) are evaluated at ZERO
// velocity without gravity, their constraint is the local linear velocity of the contact frame at v + dv: the record carries the
// velocity-level fields (wc, vc, vJ, w; the foot's velocity) of the (v + dv) motion next to the acceleration-level and force fields of the
// (0, dv) motion -- one outward sweep with both recursions, uncoupled (rounds 1 / 2: a kernel of its own, two passes of dual numbers per
// lane, a 13 kB record per impulse stage).
template <typename D, bool XYY, bool IMP>
__global__ __launch_bounds__(64) void ocp_nominal_kernel(OcpBuffers B, int dbg, int nlist, const double* __restrict__ q0) {
  using L = OcpLayout<D>;
  constexpr int NL = D::NL, LJ = D::LJ, NV = D::NV;
  constexpr int NOUT = 30, NIN = L::NJ_DYN - NOUT, TS = 31;      // outward / inward part of a joint record; row stride of the transposition
  __shared__ double tr[64 * TS];
  __shared__ long long recs[64];
  const OcpProblem* __restrict__ P = B.prob;
  const DevModel* __restrict__ m = B.model;
  const int M = B.M;
  const int lane = threadIdx.x;
  // XCD-aware block order: workgroup n runs on XCD n % 8 and every XCD has an L2 of its own, so the 2 NL + 1 task blocks that read the
  // SAME 64 solution records are given ids that agree mod 8 and lie within 8 (2 NL + 1) of each other: n = (g_hi (2 NL + 1) + task) 8 + g_lo
  // for the stage group g = 8 g_hi + g_lo.
  // q0 != nullptr: three more tasks per stage group, the Lie-group terms of the floating base (rounds 1 / 2: ocp_lie_kernel, a launch
  // of its own in front of this one; its latency-bound lanes now run next to the store-bound sweeps)
  const int NTASK = 2 * NL + 1 + ((!IMP && q0 != nullptr) ? 3 : 0);
  const unsigned n_blk = blockIdx.x;
  const unsigned g_lo = n_blk & 7u, tg = n_blk >> 3;
  const unsigned g_hi = tg / NTASK;
  const int task = (int)(tg - g_hi * NTASK);   // uniform: 0 .. NL-1 motion of a leg, NL base, NL+1 .. 2 NL pose of a leg
  const long unit = (long)(g_hi * 8u + g_lo) * 64 + lane;
  const int per = IMP ? nlist : M;
  const bool in_range = unit < (long)P->batch * per;
  const long b = in_range ? unit / per : 0;
  const int pos = in_range ? (IMP ? B.impulse_pos[(int)(unit - b * per)] : (int)(unit - b * per)) : (IMP ? B.impulse_pos[0] : 0);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  // the terminal stage has no dynamics; impulse stages have a launch of their own (IMP)
  const bool valid = in_range && pos != M - 1 && (IMP || nd->kind != 1);
  const long rec = (dbg & 2) ? 0 : b * B.NS + nd->slot;       // (lanes that are not valid compute on this record too and store nothing)
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ nom = B.nom + rec * L::NOM;
  const double gz = IMP ? 0.0 : m->gravity[2];
  const double wv = 2.0 / P->baumgarte_time_step, wp = 1.0 / (P->baumgarte_time_step * P->baumgarte_time_step);
  const double* __restrict__ sq = s + L::S_Q;
  const double* __restrict__ sv = s + L::S_V;
  const double* __restrict__ sa = s + L::S_A;
  const double qx = sq[3], qy = sq[4], qz = sq[5], qw = sq[6];

  if (!IMP && task > 2 * NL) {
    // ---- Lie-group terms of the floating base (state_equation.hxx:12-63, cost Jacobian of q (-) q_ref); every stage of the chain,
    // the terminal and the impulse stages included ----
    //   lt 0: qdiff = q (-) q_ref and Jq = dSubtractdConfigurationPlus(q, q_ref)
    //   lt 1: Fq.head(6) = (q (-) q_next).head(6), Fqq = dSubtractdConfigurationPlus(q, q_next), Fqq_inv = dSubtractdConfigurationMinus(q, q_next)^-1
    //   lt 2: Fqq_prev = dSubtractdConfigurationMinus(q_prev, q), Fqq_prev_inv
    if (!in_range) return;
    const int lt = task - 2 * NL - 1;
    double* __restrict__ zz = B.lie + rec * L::LIE;
    double R[9], p[3], Ja[36], Jb[36], d6[6];
    if (lt == 0) {
      lieRelative(B.q_ref + (long)pos * D::NQ, sq, R, p);
      lieLog6(R, p, d6);
      lieJlog6(R, p, Ja);
      for (int k = 0; k < 36; ++k) zz[L::Z_JQ + k] = Ja[k];
      for (int k = 0; k < 6; ++k) zz[L::Z_QDIFF + k] = d6[k];
    } else if (lt == 1) {
      if (pos == M - 1) return;
      lieRelative(B.sol + (b * B.NS + nd->next) * L::SOL + L::S_Q, sq, R, p);
      lieLog6(R, p, d6);
      lieJlog6(R, p, Ja);
      for (int k = 0; k < 36; ++k) zz[L::Z_FQQ + k] = Ja[k];
      for (int k = 0; k < 6; ++k) zz[L::Z_FQ6 + k] = d6[k];
      lieDDiffArg0(R, p, Ja, Jb);
      lieBlockInverse(Jb, Ja);
      for (int k = 0; k < 36; ++k) zz[L::Z_FQQI + k] = Ja[k];
    } else {
      const double* __restrict__ q_prev = (nd->prev < 0) ? (q0 + b * D::NQ) : (B.sol + (b * B.NS + nd->prev) * L::SOL + L::S_Q);      // ocp_linearizer.hxx:231-248
      lieRelative(sq, q_prev, R, p);
      lieJlog6(R, p, Ja);
      lieDDiffArg0(R, p, Ja, Jb);
      for (int k = 0; k < 36; ++k) zz[L::Z_FQQP + k] = Jb[k];
      lieBlockInverse(Jb, Ja);
      for (int k = 0; k < 36; ++k) zz[L::Z_FQQPI + k] = Ja[k];
    }
    return;
  }
  if (task <= NL) {
    // ---- motion: velocities, accelerations in the gravity field (a_gf = a - R_w^T g), forces, tau ----
    const V3 zb = v3(2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy));      // third row of R_w
    // IMP: (v, w) is the motion at v + dv -- it only feeds the velocity-level fields and the contact velocity, never a force
    V3 v = IMP ? ld3(sv) + ld3(sa) : ld3(sv), w = IMP ? ld3(sv + 3) + ld3(sa + 3) : ld3(sv + 3);
    V3 bl = ld3(sa) - gz * zb, bw = ld3(sa + 3);
    if (task == NL) {
      if (!valid) return;
      V3 hl = v3(0, 0, 0), hn = hl, f, n;
      const V3 mc = ld3(m->mc[0]);
      if (!IMP) inertia(m->mass[0], mc, m->Io[0], v, w, hl, hn);
      inertia(m->mass[0], mc, m->Io[0], bl, bw, f, n);
      double* br = nom + L::O_BASE;
      st3g(br + 0, zb); st3g(br + 3, v); st3g(br + 6, w); st3g(br + 9, hl); st3g(br + 12, hn); br[15] = 0.0;
      double* bn = nom + L::O_BN;
      if (IMP) { st3g(bn, f); st3g(bn + 3, n); }
      else { st3g(bn, f + cross(w, hl)); st3g(bn + 3, n + cross(w, hn) + cross(v, hl)); }
      // rows of [ID; C] no leg writes: the base rows (assembled from BN by the consumer) and the rows of inactive contacts
      double* idc = nom + L::O_IDC;
#pragma unroll
      for (int r = 0; r < 6; ++r) idc[r] = 0.0;
      for (int r = NV + nd->dimf; r < L::NVF; ++r) idc[r] = 0.0;
      return;
    }
    const int leg = task;
    recs[lane] = valid ? (long long)rec : -1;
    // [lane][field] -> global: instruction i stores element f = 64 i + lane of the 64 x N block, i.e. field f % N of stage f / N
    auto flush = [&](auto ncols, int offset) {
      constexpr int N = decltype(ncols)::value;
      waveLdsSync();
#pragma unroll 6
      for (int i = 0; i < N; ++i) {
        const int f = 64 * i + lane, st = f / N, k = f - st * N;
        const long long r = recs[st];
        if (r >= 0 && !(dbg & 1)) B.nom[r * L::NOM + offset + k] = tr[st * TS + k];
      }
      waveLdsSync();
    };
    double* my = tr + lane * TS;
    double Rj[LJ][12];                   // rotation (row-major) + axis: what JointFrame<-1> reads
    V3 wj[LJ], vj[LJ], bwj[LJ], blj[LJ];
    V3 zc = zb;
#pragma unroll
    for (int j = 0; j < LJ; ++j) {
      const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j;
      double sj, cj;
      sincos(sq[dof + 1], &sj, &cj);
      Mat3<double> Rm;
      revoluteRotation<double>(m->R[ji], m->axis[ji], cj, sj, Rm);
#pragma unroll
      for (int e = 0; e < 9; ++e) Rj[j][e] = Rm.m[e];
      const V3 u = ld3(m->axis[ji]), p = ld3(m->p[ji]);
      st3(&Rj[j][9], u);
      const double qdd = sa[dof], qd = IMP ? sv[dof] + qdd : sv[dof];
      V3 wc, vc, bwc, blc, vJ;
      auto step = [&](auto tag) {
        const JointFrame<decltype(tag)::value> F(Rj[j], 0, 9);
        wc = F.mulT(w); vc = F.mulT(v + cross(w, p)); bwc = F.mulT(bw); blc = F.mulT(bl + cross(bw, p));
        zc = F.mulT(zc);                 // R_w,child^T e_z: the third row of the world rotation, carried along the leg
        vJ = F.timesU(qd);
        w = wc + vJ; v = vc;
        if constexpr (IMP) { bw = bwc + F.timesU(qdd); bl = blc; }      // (the dynamics of an impulse see no velocity)
        else {
          bw = bwc + F.timesU(qdd) + F.crossKU(w, qd);
          bl = blc + F.crossKU(v, qd);
        }
      };
      if constexpr (XYY) { if (j == 0) step(AxisTag<0>{}); else step(AxisTag<1>{}); } else step(AxisTag<-1>{});
      wj[j] = w; vj[j] = v; bwj[j] = bw; blj[j] = bl;
#pragma unroll
      for (int e = 0; e < 9; ++e) my[e] = Rj[j][e];
      st3(my + 9, wc); st3(my + 12, vc); st3(my + 15, bwc); st3(my + 18, blc); st3(my + 21, zc); st3(my + 24, vJ); st3(my + 27, w);
      flush(AxisTag<NOUT>{}, L::O_JOINT + (leg * LJ + j) * L::NJ_DYN);
    }
    // ---- contact frame at the foot (tip joint of this leg): the motion-dependent part of the residual (point_contact.hxx:67-87) ----
    double* fr = nom + L::O_FEET + leg * L::NF_DYN;
    V3 fel = v3(0, 0, 0), fen = fel, fv = fel, fw = fel;
    double* idc = nom + L::O_IDC;
    if (valid && nd->active[leg]) {
      const double* __restrict__ Rc = P->contact_R[leg];
      const V3 pc = ld3(P->contact_p[leg]);
      const int row = NV + nd->row_of[leg];
      // (XYY also promises an identity rotation of the contact frame in its joint, like ANYmal's feet: Rc^T a = a)
      fv = XYY ? v + cross(w, pc) : mulT(Rc, v + cross(w, pc)); fw = XYY ? w : mulT(Rc, w);
      if constexpr (IMP) {
        // contact-velocity constraint (point_contact.hxx:145-175): LOCAL linear velocity of the frame at v + dv
        idc[row] = fv.x; idc[row + 1] = fv.y; idc[row + 2] = fv.z;
      } else {
        const V3 fam = XYY ? bl + cross(bw, pc) : mulT(Rc, bl + cross(bw, pc));      // still in the gravity field
        const V3 wxv = cross(fw, fv);
        idc[row] = fam.x + wxv.x + wv * fv.x;
        idc[row + 1] = fam.y + wxv.y + wv * fv.y;
        idc[row + 2] = fam.z + wxv.z + wv * fv.z;
      }
      // PointContact::computeJointForceFromContactForce (point_contact.hxx:15-20): jXf.act(Force(f, 0))
      fel = XYY ? ld3(s + L::S_F + 3 * leg) : mul(Rc, ld3(s + L::S_F + 3 * leg));
      fen = cross(pc, fel);
    }
    if (valid) { st3g(fr + 9, fv); st3g(fr + 12, fw); }
    // ---- inward sweep: accumulate forces, emit tau ----
    V3 Fl = v3(0, 0, 0) - fel, Fn = v3(0, 0, 0) - fen;
#pragma unroll
    for (int j = LJ - 1; j >= 0; --j) {
      const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j;
      const V3 p = ld3(m->p[ji]), mc = ld3(m->mc[ji]);
      V3 hl = v3(0, 0, 0), hn = hl, f, n;
      inertia(m->mass[ji], mc, m->Io[ji], blj[j], bwj[j], f, n);
      if constexpr (IMP) { Fl = Fl + f; Fn = Fn + n; }
      else {
        inertia(m->mass[ji], mc, m->Io[ji], vj[j], wj[j], hl, hn);
        Fl = Fl + f + cross(wj[j], hl);
        Fn = Fn + n + cross(wj[j], hn) + cross(vj[j], hl);
      }
      st3(my, hl); st3(my + 3, hn); st3(my + 6, Fl); st3(my + 9, Fn);
      flush(AxisTag<NIN>{}, L::O_JOINT + (leg * LJ + j) * L::NJ_DYN + NOUT);
      auto back = [&](auto tag) {
        const JointFrame<decltype(tag)::value> F(Rj[j], 0, 9);
        if (valid) idc[dof] = F.dotU(Fn);
        const V3 Rf = F.mul(Fl);
        Fn = F.mul(Fn) + cross(p, Rf);
        Fl = Rf;
      };
      if constexpr (XYY) { if (j == 0) back(AxisTag<0>{}); else back(AxisTag<1>{}); } else back(AxisTag<-1>{});
    }
    if (valid) { double* bn = nom + L::O_BN + 6 * (1 + leg); st3g(bn, Fl); st3g(bn + 3, Fn); }
    return;
  }
  // ---- pose: world rotation and position along the leg; at the foot, R_wf Rc and the pose-dependent part of the residual
  //   Rc^T (g_z z_f)  [takes the gravity field out of the frame acceleration]  +  (1 / D^2) (p_foot - p_contact) ----
  if (!valid) return;
  const int leg = task - NL - 1;
  double* fr = nom + L::O_FEET + leg * L::NF_DYN;
  V3 cp = v3(0, 0, 0);
  double rwc[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) rwc[e] = 0.0;
  if (!IMP && nd->active[leg]) {      // (an impulse stage has no position term: zeros)
    double Rw[9];
    Rw[0] = 1 - 2 * (qy * qy + qz * qz); Rw[1] = 2 * (qx * qy - qz * qw);     Rw[2] = 2 * (qx * qz + qy * qw);
    Rw[3] = 2 * (qx * qy + qz * qw);     Rw[4] = 1 - 2 * (qx * qx + qz * qz); Rw[5] = 2 * (qy * qz - qx * qw);
    Rw[6] = 2 * (qx * qz - qy * qw);     Rw[7] = 2 * (qy * qz + qx * qw);     Rw[8] = 1 - 2 * (qx * qx + qy * qy);
    V3 pw = ld3(sq);
#pragma unroll
    for (int j = 0; j < LJ; ++j) {
      const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j;
      double sj, cj;
      sincos(sq[dof + 1], &sj, &cj);
      Mat3<double> Rm;
      revoluteRotation<double>(m->R[ji], m->axis[ji], cj, sj, Rm);
      pw = pw + mul(Rw, ld3(m->p[ji]));
      // Rw <- Rw R: row r of the product is (R^T applied to row r of Rw)
      double Rl[12];
#pragma unroll
      for (int e = 0; e < 9; ++e) Rl[e] = Rm.m[e];
      st3(&Rl[9], ld3(m->axis[ji]));
      auto turn = [&](auto tag) {
        const JointFrame<decltype(tag)::value> F(Rl, 0, 9);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const V3 row = F.mulT(v3(Rw[3 * r], Rw[3 * r + 1], Rw[3 * r + 2]));
          Rw[3 * r] = row.x; Rw[3 * r + 1] = row.y; Rw[3 * r + 2] = row.z;
        }
      };
      if constexpr (XYY) { if (j == 0) turn(AxisTag<0>{}); else turn(AxisTag<1>{}); } else turn(AxisTag<-1>{});
    }
    const double* __restrict__ Rc = P->contact_R[leg];
    const V3 pf = pw + mul(Rw, ld3(P->contact_p[leg]));
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) rwc[3 * r + c] = XYY ? Rw[3 * r + c] : Rw[3 * r] * Rc[c] + Rw[3 * r + 1] * Rc[3 + c] + Rw[3 * r + 2] * Rc[6 + c];
    const V3 g = XYY ? gz * v3(Rw[6], Rw[7], Rw[8]) : mulT(Rc, gz * v3(Rw[6], Rw[7], Rw[8]));
    cp = v3(g.x + wp * (pf.x - nd->contact_point[leg][0]), g.y + wp * (pf.y - nd->contact_point[leg][1]),
            g.z + wp * (pf.z - nd->contact_point[leg][2]));
  }
#pragma unroll
  for (int e = 0; e < 9; ++e) fr[e] = rwc[e];
  st3g(fr + 15, cp);
}

template <typename D>
void OcpLaunch<D>::nominal(const OcpBuffers& B, long batch, int M, hipStream_t st, const double* q0_lie, hipStream_t st_imp) {
  const unsigned ntask = 2 * D::NL + 1 + (q0_lie ? 3 : 0);
  const unsigned groups = (unsigned)((batch * M + 63) / 64), blocks = ((groups + 7) / 8) * 8 * ntask;
  static const int dbg = getenv("IDOCP_NOM_DBG") ? atoi(getenv("IDOCP_NOM_DBG")) : 0;
  if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_nominal_kernel<D, true, false>), dim3(blocks), dim3(64), 0, st, B, dbg, 0, q0_lie);
  else hipLaunchKernelGGL((ocp_nominal_kernel<D, false, false>), dim3(blocks), dim3(64), 0, st, B, dbg, 0, q0_lie);
  // the impulse stages of a forward-Euler chain (ParNMPC's go through K5a / K9i)
  if (B.n_impulse_fe > 0) {
    const unsigned gi = (unsigned)((batch * B.n_impulse_fe + 63) / 64), bi = ((gi + 7) / 8) * 8 * (2 * D::NL + 1);
    const double* none = nullptr;
    hipStream_t si = st_imp ? st_imp : st;      // (impulse stages have records of their own: independent of the launch above)
    if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_nominal_kernel<D, true, true>), dim3(bi), dim3(64), 0, si, B, dbg, B.n_impulse_fe, none);
    else hipLaunchKernelGGL((ocp_nominal_kernel<D, false, true>), dim3(bi), dim3(64), 0, si, B, dbg, B.n_impulse_fe, none);
  }
}

template void OcpLaunch<LeggedDims<4, 3>>::nominal(const OcpBuffers&, long, int, hipStream_t, const double*, hipStream_t);

}  // namespace idocp_dev
