// URDF -> idocp_model_t.
//
// Host-side replacement for the part of idocp::Robot's constructor that the
// reference delegates to pinocchio::urdf::buildModel (src/robot/robot.cpp:26,62):
// it reads the URDF, walks the kinematic tree in pinocchio/urdfdom order and
// produces the flat model the HIP kernels consume.  Conventions reproduced
// (SURVEY.md section 9.4):
//   * child joints of a link are visited in alphabetical order of JOINT name
//     (urdfdom keeps joints in a std::map and fills child lists from it);
//   * fixed joints do not create a joint: the child link's inertia is merged
//     into the body of the nearest moving ancestor joint and the fixed
//     transform is folded into the placements of what hangs below it;
//   * a URDF joint of type "floating" becomes a free-flyer (q = xyz + quat
//     xyzw, v = local linear + local angular velocity);
//   * frames are numbered universe(0), root_joint(1), <root link>(2), then
//     (joint frame, body frame) per visited link, depth first.
// No third-party XML library: a ~100-line recursive-descent reader that
// understands elements, attributes, comments and processing instructions is
// all a URDF needs.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "idocp_hip.h"
#include "host_util.hpp"

namespace idocp_host {

// ----------------------------------------------------------------- XML ----
struct XmlNode {
  std::string name;
  std::map<std::string, std::string> attr;
  std::vector<std::unique_ptr<XmlNode>> children;
  const XmlNode* child(const char* n) const {
    for (auto& c : children) if (c->name == n) return c.get();
    return nullptr;
  }
  std::string get(const char* a, const char* dflt = "") const {
    auto it = attr.find(a);
    return it == attr.end() ? std::string(dflt) : it->second;
  }
};

class XmlReader {
 public:
  explicit XmlReader(const std::string& text) : s_(text), i_(0) {}
  std::unique_ptr<XmlNode> parse() {
    skipMisc();
    return parseElement();
  }
  bool ok() const { return ok_; }

 private:
  const std::string& s_;
  size_t i_;
  bool ok_ = true;
  void skipWs() { while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_; }
  bool starts(const char* p) const { return s_.compare(i_, std::strlen(p), p) == 0; }
  void skipMisc() {
    for (;;) {
      skipWs();
      if (starts("<?")) { size_t e = s_.find("?>", i_); i_ = (e == std::string::npos) ? s_.size() : e + 2; }
      else if (starts("<!--")) { size_t e = s_.find("-->", i_); i_ = (e == std::string::npos) ? s_.size() : e + 3; }
      else if (starts("<!")) { size_t e = s_.find('>', i_); i_ = (e == std::string::npos) ? s_.size() : e + 1; }
      else break;
    }
  }
  std::string parseName() {
    size_t b = i_;
    while (i_ < s_.size() && (std::isalnum((unsigned char)s_[i_]) || s_[i_] == '_' || s_[i_] == ':' || s_[i_] == '-' || s_[i_] == '.')) ++i_;
    return s_.substr(b, i_ - b);
  }
  // (Every exit on damaged input sets ok_ and returns null: a truncated closing tag used to send the position back to 0 -- `npos + 1` -- and the
  //  reader into unbounded recursion; tests/test_urdf_reader_robustness.py feeds it truncated and corrupted files.)
  std::unique_ptr<XmlNode> parseElement(int depth = 0) {
    if (depth > 64 || i_ >= s_.size() || s_[i_] != '<') { ok_ = false; return nullptr; }      // (a URDF nests five deep)
    ++i_;
    auto node = std::make_unique<XmlNode>();
    node->name = parseName();
    for (;;) {
      skipWs();
      if (i_ >= s_.size()) { ok_ = false; return nullptr; }
      if (s_[i_] == '/') {                                   // "/>"
        if (i_ + 1 >= s_.size() || s_[i_ + 1] != '>') { ok_ = false; return nullptr; }
        i_ += 2;
        return node;
      }
      if (s_[i_] == '>') { ++i_; break; }
      std::string key = parseName();
      if (key.empty()) { ok_ = false; return nullptr; }      // (neither an attribute nor the end of the tag: the loop would not advance)
      skipWs();
      if (i_ >= s_.size() || s_[i_] != '=') { ok_ = false; return nullptr; }
      ++i_; skipWs();
      if (i_ >= s_.size() || (s_[i_] != '"' && s_[i_] != '\'')) { ok_ = false; return nullptr; }
      char quote = s_[i_++];
      size_t e = s_.find(quote, i_);
      if (e == std::string::npos) { ok_ = false; return nullptr; }
      node->attr[key] = s_.substr(i_, e - i_);
      i_ = e + 1;
    }
    for (;;) {                                               // content
      size_t lt = s_.find('<', i_);
      if (lt == std::string::npos) { ok_ = false; return nullptr; }
      i_ = lt;
      if (starts("<!--") || starts("<?") || starts("<!")) { skipMisc(); continue; }
      if (starts("</")) {
        size_t e = s_.find('>', i_);
        if (e == std::string::npos) { ok_ = false; return nullptr; }
        i_ = e + 1;
        return node;
      }
      auto c = parseElement(depth + 1);
      if (!c) return nullptr;
      node->children.push_back(std::move(c));
    }
  }
};

// ------------------------------------------------------------ geometry ----
struct SE3 { double R[9]; double p[3]; };

static SE3 se3Identity() { SE3 m{{1,0,0,0,1,0,0,0,1},{0,0,0}}; return m; }

static void matmul3(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0; for (int k = 0; k < 3; ++k) s += A[3*i+k] * B[3*k+j]; C[3*i+j] = s;
  }
}
static SE3 compose(const SE3& a, const SE3& b) {            // a * b
  SE3 c; matmul3(a.R, b.R, c.R);
  for (int i = 0; i < 3; ++i) c.p[i] = a.p[i] + a.R[3*i]*b.p[0] + a.R[3*i+1]*b.p[1] + a.R[3*i+2]*b.p[2];
  return c;
}
// URDF rpy: R = Rz(yaw) Ry(pitch) Rx(roll)
static void rpyToR(const double rpy[3], double* R) {
  const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
  const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
  const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  R[0] = cy*cp; R[1] = cy*sp*sr - sy*cr; R[2] = cy*sp*cr + sy*sr;
  R[3] = sy*cp; R[4] = sy*sp*sr + cy*cr; R[5] = sy*sp*cr - cy*sr;
  R[6] = -sp;   R[7] = cp*sr;            R[8] = cp*cr;
}
static void parse3(const std::string& s, double out[3], double d0, double d1, double d2) {
  out[0] = d0; out[1] = d1; out[2] = d2;
  if (s.empty()) return;
  std::istringstream is(s); is >> out[0] >> out[1] >> out[2];
}
static SE3 parseOrigin(const XmlNode* n) {
  SE3 m = se3Identity();
  if (!n) return m;
  double xyz[3], rpy[3];
  parse3(n->get("xyz"), xyz, 0, 0, 0);
  parse3(n->get("rpy"), rpy, 0, 0, 0);
  rpyToR(rpy, m.R);
  for (int i = 0; i < 3; ++i) m.p[i] = xyz[i];
  return m;
}

struct Inertia { double m = 0; double c[3] = {0,0,0}; double I[9] = {0,0,0,0,0,0,0,0,0}; };

// Y expressed in frame B, placement M of B in A  ->  Y expressed in A.
static Inertia transformInertia(const SE3& M, const Inertia& Y) {
  Inertia o; o.m = Y.m;
  for (int i = 0; i < 3; ++i) o.c[i] = M.p[i] + M.R[3*i]*Y.c[0] + M.R[3*i+1]*Y.c[1] + M.R[3*i+2]*Y.c[2];
  double RI[9], Rt[9];
  matmul3(M.R, Y.I, RI);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rt[3*i+j] = M.R[3*j+i];
  matmul3(RI, Rt, o.I);
  return o;
}
// Sum of two inertias expressed in the same frame (parallel-axis theorem).
static Inertia addInertia(const Inertia& a, const Inertia& b) {
  if (a.m == 0 && b.m == 0) return a;
  Inertia o; o.m = a.m + b.m;
  for (int i = 0; i < 3; ++i) o.c[i] = (a.m * a.c[i] + b.m * b.c[i]) / o.m;
  auto shift = [&](const Inertia& y, double* I) {
    double d[3] = {y.c[0]-o.c[0], y.c[1]-o.c[1], y.c[2]-o.c[2]};
    double dd = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
      I[3*i+j] += y.I[3*i+j] + y.m * ((i == j ? dd : 0.0) - d[i]*d[j]);
  };
  shift(a, o.I); shift(b, o.I);
  return o;
}

// ---------------------------------------------------------------- URDF ----
struct UrdfJoint {
  std::string name, type, parent, child;
  SE3 origin; double axis[3]; double lower, upper, effort, velocity; bool has_limit;
};
struct UrdfLink { std::string name; Inertia Y; bool has_inertial = false; };

struct FrameEntry { std::string name; int joint; SE3 placement; };

struct UrdfTree {
  std::map<std::string, UrdfLink> links;
  std::map<std::string, UrdfJoint> joints;            // sorted by joint name
  std::map<std::string, std::vector<std::string>> child_joints;   // link -> joints
  std::string root;
};

static bool loadUrdf(const std::string& path, UrdfTree& t, std::string& err) {
  std::ifstream f(path);
  if (!f) { err = "cannot open URDF file: " + path; return false; }
  std::stringstream ss; ss << f.rdbuf();
  const std::string text = ss.str();
  XmlReader reader(text);
  auto root = reader.parse();
  if (!root || !reader.ok() || root->name != "robot") { err = "malformed URDF: " + path; return false; }
  for (auto& c : root->children) {
    if (c->name == "link") {
      UrdfLink l; l.name = c->get("name");
      if (const XmlNode* in = c->child("inertial")) {
        l.has_inertial = true;
        SE3 o = parseOrigin(in->child("origin"));
        Inertia y;
        if (const XmlNode* m = in->child("mass")) y.m = std::atof(m->get("value", "0").c_str());
        if (const XmlNode* I = in->child("inertia")) {
          const double ixx = std::atof(I->get("ixx","0").c_str()), ixy = std::atof(I->get("ixy","0").c_str());
          const double ixz = std::atof(I->get("ixz","0").c_str()), iyy = std::atof(I->get("iyy","0").c_str());
          const double iyz = std::atof(I->get("iyz","0").c_str()), izz = std::atof(I->get("izz","0").c_str());
          double Ic[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz};
          std::memcpy(y.I, Ic, sizeof(Ic));
        }
        // inertial frame -> link frame: com = origin.p, I = R I R^T
        l.Y = transformInertia(o, y);
      }
      t.links[l.name] = l;
    } else if (c->name == "joint" && c->child("parent") && c->child("child")) {
      UrdfJoint j; j.name = c->get("name"); j.type = c->get("type");
      j.parent = c->child("parent")->get("link");
      j.child = c->child("child")->get("link");
      j.origin = parseOrigin(c->child("origin"));
      parse3(c->child("axis") ? c->child("axis")->get("xyz") : "", j.axis, 1, 0, 0);
      j.has_limit = false; j.lower = j.upper = j.effort = j.velocity = 0;
      if (const XmlNode* lim = c->child("limit")) {
        j.has_limit = true;
        j.lower = std::atof(lim->get("lower","0").c_str());
        j.upper = std::atof(lim->get("upper","0").c_str());
        j.effort = std::atof(lim->get("effort","0").c_str());
        j.velocity = std::atof(lim->get("velocity","0").c_str());
      }
      t.joints[j.name] = j;
    }
  }
  std::map<std::string, bool> is_child;
  for (auto& kv : t.joints) {                        // alphabetical by joint name
    t.child_joints[kv.second.parent].push_back(kv.first);
    is_child[kv.second.child] = true;
  }
  for (auto& kv : t.links) if (!is_child.count(kv.first)) { t.root = kv.first; break; }
  if (t.root.empty()) { err = "URDF has no root link"; return false; }
  return true;
}

struct Builder {
  const UrdfTree& t;
  idocp_model_t& m;
  std::vector<FrameEntry> frames;
  std::vector<Inertia> body;           // per joint
  std::string err;

  // Visit `link`, attached below moving joint `pj` (-1 universe) with the link
  // frame placed at `M` in that joint's frame.
  void visit(const std::string& link, int pj, const SE3& M) {
    auto it = t.child_joints.find(link);
    if (it == t.child_joints.end()) return;
    for (const std::string& jn : it->second) {
      const UrdfJoint& j = t.joints.at(jn);
      const UrdfLink& cl = t.links.at(j.child);
      SE3 Mj = compose(M, j.origin);             // joint frame in pj's frame
      if (j.type == "fixed") {
        frames.push_back({j.name, pj, Mj});
        frames.push_back({cl.name, pj, Mj});
        if (pj >= 0 && cl.has_inertial) body[pj] = addInertia(body[pj], transformInertia(Mj, cl.Y));
        visit(j.child, pj, Mj);
      } else if (j.type == "revolute" || j.type == "continuous" || j.type == "floating") {
        if (m.njoints >= IDOCP_MAX_JOINTS) { err = "too many joints"; return; }
        const int id = m.njoints++;
        const bool ff = (j.type == "floating");
        m.parent[id] = pj;
        m.jtype[id] = ff ? IDOCP_JOINT_FREEFLYER : IDOCP_JOINT_REVOLUTE;
        m.idx_q[id] = m.nq; m.idx_v[id] = m.nv;
        m.nq += ff ? 7 : 1; m.nv += ff ? 6 : 1;
        if (m.nv > IDOCP_MAX_NV || m.nq > IDOCP_MAX_NQ) { err = "too many degrees of freedom"; return; }
        double n = std::sqrt(j.axis[0]*j.axis[0] + j.axis[1]*j.axis[1] + j.axis[2]*j.axis[2]);
        if (!ff && !(n > 0)) { err = "joint '" + j.name + "' has no axis direction (axis xyz = 0 0 0)"; return; }
        for (int k = 0; k < 3; ++k) m.axis[id][k] = ff ? 0.0 : j.axis[k] / (n > 0 ? n : 1.0);
        std::memcpy(m.plc_R[id], Mj.R, sizeof(Mj.R));
        std::memcpy(m.plc_p[id], Mj.p, sizeof(Mj.p));
        body.push_back(cl.has_inertial ? cl.Y : Inertia());
        if (ff) { if (id != 0 || pj != -1) { err = "floating joint must be the root joint"; return; } m.has_floating_base = 1; }
        else {
          const int a = m.idx_v[id];
          lim_lo[a] = j.lower; lim_hi[a] = j.upper; lim_e[a] = j.effort; lim_v[a] = j.velocity;
        }
        frames.push_back({j.name, id, se3Identity()});
        frames.push_back({cl.name, id, se3Identity()});
        visit(j.child, id, se3Identity());
      } else {
        err = "unsupported URDF joint type '" + j.type + "' (joint " + j.name + ")";
        return;
      }
      if (!err.empty()) return;
    }
  }
  double lim_lo[IDOCP_MAX_NV] = {0}, lim_hi[IDOCP_MAX_NV] = {0}, lim_e[IDOCP_MAX_NV] = {0}, lim_v[IDOCP_MAX_NV] = {0};
};

static bool buildFrames(const std::string& path, idocp_model_t& m, std::vector<FrameEntry>& frames, std::string& err) {
  UrdfTree t;
  if (!loadUrdf(path, t, err)) return false;
  std::memset(&m, 0, sizeof(m));
  Builder b{t, m, {}, {}, {}};
  b.frames.push_back({"universe", -1, se3Identity()});
  b.frames.push_back({"root_joint", -1, se3Identity()});
  b.frames.push_back({t.root, -1, se3Identity()});
  b.visit(t.root, -1, se3Identity());
  if (!b.err.empty()) { err = b.err; return false; }
  if (m.njoints == 0) { err = "URDF has no moving joints"; return false; }
  const int np = m.has_floating_base ? 6 : 0;
  m.nu = m.nv - np;
  m.total_mass = 0;
  for (int i = 0; i < m.njoints; ++i) {
    m.mass[i] = b.body[i].m; m.total_mass += b.body[i].m;
    std::memcpy(m.com[i], b.body[i].c, sizeof(double)*3);
    std::memcpy(m.inertia[i], b.body[i].I, sizeof(double)*9);
  }
  for (int k = 0; k < m.nu; ++k) {
    m.q_min[k] = b.lim_lo[np + k]; m.q_max[k] = b.lim_hi[np + k];
    m.u_max[k] = b.lim_e[np + k];  m.v_max[k] = b.lim_v[np + k];
  }
  m.gravity[0] = 0; m.gravity[1] = 0; m.gravity[2] = -9.81;
  // every number the kernels will compute with is one (a "nan" or an overflowing literal in the file parses to a non-finite double and would
  // surface as NaN directions a thousand lines from here); joint limits may be infinite -- an unlimited joint -- but not NaN
  auto finite = [](const double* p, int n) { for (int k = 0; k < n; ++k) if (!std::isfinite(p[k])) return false; return true; };
  bool good = std::isfinite(m.total_mass);
  for (int i = 0; i < m.njoints && good; ++i)
    good = std::isfinite(m.mass[i]) && finite(m.com[i], 3) && finite(m.inertia[i], 9) && finite(m.axis[i], 3) && finite(m.plc_R[i], 9) && finite(m.plc_p[i], 3);
  for (int k = 0; k < m.nu && good; ++k) good = !(std::isnan(m.q_min[k]) || std::isnan(m.q_max[k]) || std::isnan(m.u_max[k]) || std::isnan(m.v_max[k]));
  for (const FrameEntry& f : b.frames) if (good) good = finite(f.placement.R, 9) && finite(f.placement.p, 3);
  if (!good) { err = "malformed URDF: a value that is not a finite number: " + path; return false; }
  frames = b.frames;
  return true;
}

}  // namespace idocp_host

using namespace idocp_host;

extern "C" int idocp_model_from_urdf(const char* path_to_urdf, const int* contact_frames,
                                     int ncontacts, idocp_model_t* out) {
  if (!path_to_urdf || !out || ncontacts < 0 || ncontacts > IDOCP_MAX_CONTACTS ||
      (ncontacts > 0 && !contact_frames)) {
    set_last_error("idocp_model_from_urdf: invalid argument");
    return IDOCP_E_ARG;
  }
  std::vector<FrameEntry> frames; std::string err;
  if (!buildFrames(path_to_urdf, *out, frames, err)) { set_last_error(err); return IDOCP_E_IO; }
  out->ncontacts = ncontacts;
  for (int c = 0; c < ncontacts; ++c) {
    const int fid = contact_frames[c];
    if (fid < 0 || fid >= (int)frames.size() || frames[fid].joint < 0) {
      set_last_error("idocp_model_from_urdf: invalid contact frame index " + std::to_string(fid));
      return IDOCP_E_ARG;
    }
    out->contact_frame_id[c] = fid;
    out->contact_joint[c] = frames[fid].joint;
    std::memcpy(out->contact_R[c], frames[fid].placement.R, sizeof(double)*9);
    std::memcpy(out->contact_p[c], frames[fid].placement.p, sizeof(double)*3);
  }
  return IDOCP_OK;
}

extern "C" int idocp_model_frame_placement(const char* path_to_urdf, int frame_id, int* joint, double* R, double* p) {
  if (!path_to_urdf || !joint || !R || !p) return IDOCP_E_ARG;
  idocp_model_t m; std::vector<FrameEntry> frames; std::string err;
  if (!buildFrames(path_to_urdf, m, frames, err)) { set_last_error(err); return IDOCP_E_IO; }
  if (frame_id < 0 || frame_id >= (int)frames.size() || frames[frame_id].joint < 0) {
    set_last_error("idocp_model_frame_placement: frame " + std::to_string(frame_id) + " does not exist or is not attached to a joint");
    return IDOCP_E_ARG;
  }
  *joint = frames[frame_id].joint;
  std::memcpy(R, frames[frame_id].placement.R, sizeof(double) * 9);
  std::memcpy(p, frames[frame_id].placement.p, sizeof(double) * 3);
  return IDOCP_OK;
}

extern "C" int idocp_model_frame_id(const char* path_to_urdf, const char* frame_name) {
  if (!path_to_urdf || !frame_name) return -1;
  idocp_model_t m; std::vector<FrameEntry> frames; std::string err;
  if (!buildFrames(path_to_urdf, m, frames, err)) { set_last_error(err); return -1; }
  for (size_t i = 0; i < frames.size(); ++i) if (frames[i].name == frame_name) return (int)i;
  return -1;
}

extern "C" void idocp_cost_init(idocp_cost_t* cost) { if (cost) std::memset(cost, 0, sizeof(*cost)); }

extern "C" void idocp_constraints_init(idocp_constraints_t* c) {
  if (!c) return;
  c->joint_position_limits = 1; c->joint_velocity_limits = 1; c->joint_torque_limits = 1;
  c->linearized_friction_cone = 0; c->mu = 0.7;
  c->barrier = 1.0e-04; c->fraction_to_boundary_rate = 0.995;
  c->linearized_impulse_friction_cone = 0;
  c->friction_cone = 0; c->impulse_friction_cone = 0;
  c->joint_acceleration_lower_limit = 0; c->joint_acceleration_upper_limit = 0;
  for (int i = 0; i < IDOCP_MAX_NV; ++i) { c->a_min[i] = 0.0; c->a_max[i] = 0.0; }
  c->contact_distance = 0;
}
