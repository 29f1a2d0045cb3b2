// Host-side forward kinematics of the contact frames.
//
// Replaces the set-up calls the reference's drivers make before building the
// contact sequence (examples/anymal/ocp_benchmark.cpp:104-106):
//   Robot::updateFrameKinematics(q)   (include/idocp/robot/robot.hxx:85-91)
//   Robot::setContactPoints(status)   (robot.hxx:262-271, PointContact::framePosition)
// This is problem set-up (one call per contact phase), not the hot path; the
// stage kernels evaluate their own kinematics on the device.
#include <cmath>

#include "host_util.hpp"
#include "idocp_hip.h"

namespace {

struct Xf { double R[9]; double p[3]; };   // x_world = R x_local + p, R row-major

void compose(const Xf& a, const double* Rb, const double* pb, Xf* out) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += a.R[3 * r + k] * Rb[3 * k + c];
      out->R[3 * r + c] = acc;
    }
    out->p[r] = a.p[r] + a.R[3 * r] * pb[0] + a.R[3 * r + 1] * pb[1] + a.R[3 * r + 2] * pb[2];
  }
}

void rodrigues(const double* axis, double angle, double* R) {
  const double c = std::cos(angle), s = std::sin(angle), t = 1.0 - c;
  const double x = axis[0], y = axis[1], z = axis[2];
  R[0] = t * x * x + c;     R[1] = t * x * y - s * z; R[2] = t * x * z + s * y;
  R[3] = t * x * y + s * z; R[4] = t * y * y + c;     R[5] = t * y * z - s * x;
  R[6] = t * x * z - s * y; R[7] = t * y * z + s * x; R[8] = t * z * z + c;
}

void quatToRot(const double* qq, double* R) {     // xyzw
  const double x = qq[0], y = qq[1], z = qq[2], w = qq[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

}  // namespace

namespace {
// world placement of every joint frame at q
void jointPlacements(const idocp_model_t* m, const double* q, Xf* world) {
  for (int i = 0; i < m->njoints; ++i) {
    Xf base;
    if (m->parent[i] < 0) {
      for (int k = 0; k < 9; ++k) base.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
      base.p[0] = base.p[1] = base.p[2] = 0.0;
    } else {
      base = world[m->parent[i]];
    }
    Xf placed;
    compose(base, m->plc_R[i], m->plc_p[i], &placed);
    double Rj[9], pj[3] = {0.0, 0.0, 0.0};
    if (m->jtype[i] == IDOCP_JOINT_FREEFLYER) {
      const double* qb = q + m->idx_q[i];
      pj[0] = qb[0]; pj[1] = qb[1]; pj[2] = qb[2];
      quatToRot(qb + 3, Rj);
    } else {
      rodrigues(m->axis[i], q[m->idx_q[i]], Rj);
    }
    compose(placed, Rj, pj, &world[i]);
  }
}
}  // namespace

extern "C" int idocp_model_contact_positions(const idocp_model_t* m, const double* q, double* points) {
  if (m == nullptr || q == nullptr || points == nullptr) {
    idocp_host::set_last_error("invalid argument: model, q and points must not be null!");
    return IDOCP_E_ARG;
  }
  Xf world[IDOCP_MAX_JOINTS];
  jointPlacements(m, q, world);
  for (int c = 0; c < m->ncontacts; ++c) {
    Xf f;
    compose(world[m->contact_joint[c]], m->contact_R[c], m->contact_p[c], &f);
    for (int k = 0; k < 3; ++k) points[3 * c + k] = f.p[k];
  }
  return IDOCP_OK;
}

// Robot::framePosition / frameRotation / framePlacement (robot.hxx:206-233) of a frame that sits on `joint` with the local placement (R_local, p_local)
// -- what idocp_model_frame_placement returns for a frame id of the URDF.  R_world row-major [9], p_world [3].
extern "C" int idocp_model_frame_world_placement(const idocp_model_t* m, const double* q, int joint, const double* R_local, const double* p_local,
                                                 double* R_world, double* p_world) {
  if (m == nullptr || q == nullptr || R_local == nullptr || p_local == nullptr || R_world == nullptr || p_world == nullptr) {
    idocp_host::set_last_error("invalid argument: null pointer!");
    return IDOCP_E_ARG;
  }
  if (joint < -1 || joint >= m->njoints) {
    idocp_host::set_last_error("invalid argument: the frame's joint does not exist in this model!");
    return IDOCP_E_ARG;
  }
  Xf world[IDOCP_MAX_JOINTS], f, root;
  jointPlacements(m, q, world);
  for (int k = 0; k < 9; ++k) root.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
  root.p[0] = root.p[1] = root.p[2] = 0.0;
  compose(joint < 0 ? root : world[joint], R_local, p_local, &f);      // (joint -1: a frame fixed to the world)
  for (int k = 0; k < 9; ++k) R_world[k] = f.R[k];
  for (int k = 0; k < 3; ++k) p_world[k] = f.p[k];
  return IDOCP_OK;
}
