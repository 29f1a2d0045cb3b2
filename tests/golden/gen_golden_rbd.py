#!/usr/bin/env python3
"""Generate golden rigid-body vectors for the oracle (run in the build container).

The reference (mayataka/idocp) delegates rigid-body arithmetic to pinocchio
(SURVEY.md section 8c), which is neither vendored nor installed here, and its own
tests hold no numeric vectors.  This script is an INDEPENDENT restatement used to
pin the C++ oracle and the URDF reader:

  * URDF parsing with xml.etree (pinocchio/urdfdom conventions: children in
    alphabetical order of joint name, fixed joints merged, `floating` -> free
    flyer with q = xyz + quaternion xyzw, local-frame velocities);
  * inverse dynamics by the textbook body-frame recursive Newton-Euler
    algorithm (Featherstone, RBDA ch. 5) -- a different formulation from the
    world-frame analytic-derivative algorithm the oracle restates;
  * derivatives by complex-step differentiation (h = 1e-30), which is exact to
    rounding and shares no code path with the analytic derivatives.

Outputs tests/golden/rbd_<robot>.json (model constants + samples).  Only data is
written; no reference source text is copied.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------- URDF ----
def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def parse_origin(el):
    R, p = np.eye(3), np.zeros(3)
    if el is not None:
        p = np.array([float(x) for x in el.get("xyz", "0 0 0").split()])
        R = rpy_to_R([float(x) for x in el.get("rpy", "0 0 0").split()])
    return R, p


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def transform_inertia(R, p, Y):
    m, c, I = Y
    return (m, p + R @ c, R @ I @ R.T)


def add_inertia(a, b):
    ma, ca, Ia = a
    mb, cb, Ib = b
    if ma == 0 and mb == 0:
        return a
    m = ma + mb
    c = (ma * ca + mb * cb) / m
    I = np.zeros((3, 3))
    for (mi, ci, Ii) in (a, b):
        d = ci - c
        I += Ii + mi * (d @ d * np.eye(3) - np.outer(d, d))
    return (m, c, I)


def load_model(path, contact_frames=()):
    root = ET.parse(path).getroot()
    links, joints = {}, {}
    for l in root.findall("link"):
        Y = None
        inr = l.find("inertial")
        if inr is not None:
            R, p = parse_origin(inr.find("origin"))
            m = float(inr.find("mass").get("value"))
            i = inr.find("inertia")
            ixx, ixy, ixz, iyy, iyz, izz = [float(i.get(k, "0")) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]
            Ic = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
            Y = transform_inertia(R, p, (m, np.zeros(3), Ic))
        links[l.get("name")] = Y
    children = {}
    is_child = set()
    for j in root.findall("joint"):
        if j.find("parent") is None:
            continue
        joints[j.get("name")] = j
    for name in sorted(joints):
        j = joints[name]
        children.setdefault(j.find("parent").get("link"), []).append(name)
        is_child.add(j.find("child").get("link"))
    root_link = [n for n in sorted(links) if n not in is_child][0]

    M = dict(parent=[], jtype=[], idx_q=[], idx_v=[], axis=[], plc_R=[], plc_p=[], body=[],
             q_min=[], q_max=[], v_max=[], u_max=[], nq=0, nv=0, floating=0)
    frames = [("universe", -1, np.eye(3), np.zeros(3)), ("root_joint", -1, np.eye(3), np.zeros(3)),
              (root_link, -1, np.eye(3), np.zeros(3))]

    def visit(link, pj, R, p):
        for jn in children.get(link, []):
            j = joints[jn]
            child = j.find("child").get("link")
            Ro, po = parse_origin(j.find("origin"))
            Rj, pj_ = R @ Ro, p + R @ po
            typ = j.get("type")
            if typ == "fixed":
                frames.append((jn, pj, Rj, pj_))
                frames.append((child, pj, Rj, pj_))
                if pj >= 0 and links[child] is not None:
                    M["body"][pj] = add_inertia(M["body"][pj], transform_inertia(Rj, pj_, links[child]))
                visit(child, pj, Rj, pj_)
            else:
                jid = len(M["parent"])
                ff = typ == "floating"
                M["parent"].append(pj)
                M["jtype"].append(1 if ff else 0)
                M["idx_q"].append(M["nq"])
                M["idx_v"].append(M["nv"])
                M["nq"] += 7 if ff else 1
                M["nv"] += 6 if ff else 1
                ax = np.zeros(3)
                if not ff:
                    ax = np.array([float(x) for x in j.find("axis").get("xyz").split()])
                    ax = ax / np.linalg.norm(ax)
                    lim = j.find("limit")
                    M["q_min"].append(float(lim.get("lower", "0")))
                    M["q_max"].append(float(lim.get("upper", "0")))
                    M["v_max"].append(float(lim.get("velocity", "0")))
                    M["u_max"].append(float(lim.get("effort", "0")))
                else:
                    M["floating"] = 1
                M["axis"].append(ax)
                M["plc_R"].append(Rj)
                M["plc_p"].append(pj_)
                M["body"].append(links[child] if links[child] is not None else (0.0, np.zeros(3), np.zeros((3, 3))))
                frames.append((jn, jid, np.eye(3), np.zeros(3)))
                frames.append((child, jid, np.eye(3), np.zeros(3)))
                visit(child, jid, np.eye(3), np.zeros(3))

    visit(root_link, -1, np.eye(3), np.zeros(3))
    M["frames"] = frames
    M["contacts"] = [(fid, frames[fid][1], frames[fid][2], frames[fid][3]) for fid in contact_frames]
    M["njoints"] = len(M["parent"])
    M["gravity"] = np.array([0, 0, -9.81])
    return M


# ------------------------------------------------------- body-frame RNEA ----
def rot_axis(axis, q):
    c, s = np.cos(q), np.sin(q)
    K = skew(axis)
    return c * np.eye(3) + s * K + (1 - c) * np.outer(axis, axis)


def quat_to_R(x, y, z, w):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def joint_transform(M, i, q, dq_seed=None):
    """Placement of joint frame i in its parent joint frame (R, p).

    dq_seed (complex, size nv or None): tangent perturbation applied on the
    configuration manifold, q (+) dq -- first-order exact, used only with the
    complex step where |dq| ~ 1e-30."""
    iq, iv = M["idx_q"][i], M["idx_v"][i]
    if M["jtype"][i] == 0:
        ang = q[iq] + (dq_seed[iv] if dq_seed is not None else 0)
        Rj, pj = rot_axis(M["axis"][i], ang), np.zeros(3)
    else:
        Rj = quat_to_R(*q[iq + 3:iq + 7]).astype(complex if dq_seed is not None else float)
        pj = np.array(q[iq:iq + 3], dtype=Rj.dtype)
        if dq_seed is not None:
            dv, dw = dq_seed[iv:iv + 3], dq_seed[iv + 3:iv + 6]
            pj = pj + Rj @ dv
            Rj = Rj @ (np.eye(3) + skew(dw))
    return M["plc_R"][i] @ Rj, M["plc_p"][i] + M["plc_R"][i] @ pj


def rnea(M, q, v, a, fext=None, dq_seed=None, gravity=True):
    """tau = ID(q, v, a, fext); fext[i] = (f, n) acting on body i in its joint frame."""
    n = M["njoints"]
    cplx = dq_seed is not None or np.iscomplexobj(v) or np.iscomplexobj(a) or (fext is not None and np.iscomplexobj(fext))
    dt = complex if cplx else float
    Rs, ps, vl, vw, al, aw, fl, fn = ([None] * n for _ in range(8))
    g = M["gravity"] if gravity else np.zeros(3)
    for i in range(n):
        R, p = joint_transform(M, i, q, dq_seed)
        Rs[i], ps[i] = R, p
        pa = M["parent"][i]
        iv = M["idx_v"][i]
        if pa < 0:
            pvl, pvw, pal, paw = np.zeros(3, dt), np.zeros(3, dt), -g.astype(dt), np.zeros(3, dt)
        else:
            pvl, pvw, pal, paw = vl[pa], vw[pa], al[pa], aw[pa]
        # motion transform parent -> child
        w_ = R.T @ pvw
        v_ = R.T @ (pvl + np.cross(pvw, p))
        aw_ = R.T @ paw
        al_ = R.T @ (pal + np.cross(paw, p))
        if M["jtype"][i] == 0:
            ax = M["axis"][i]
            vJl, vJw = np.zeros(3, dt), ax * v[iv]
            aJl, aJw = np.zeros(3, dt), ax * a[iv]
        else:
            vJl, vJw = np.array(v[iv:iv + 3], dt), np.array(v[iv + 3:iv + 6], dt)
            aJl, aJw = np.array(a[iv:iv + 3], dt), np.array(a[iv + 3:iv + 6], dt)
        vl[i], vw[i] = v_ + vJl, w_ + vJw
        # a_i = X a_p + S qdd + v_i x (S qd)
        al[i] = al_ + aJl + np.cross(vw[i], vJl) + np.cross(vl[i], vJw)
        aw[i] = aw_ + aJw + np.cross(vw[i], vJw)
        m, c, I = M["body"][i]
        # spatial inertia about the joint-frame origin
        hl = m * (vl[i] + np.cross(vw[i], c))
        hn = I @ vw[i] + np.cross(c, hl)
        f_l = m * (al[i] + np.cross(aw[i], c))
        f_n = I @ aw[i] + np.cross(c, f_l)
        fl[i] = f_l + np.cross(vw[i], hl)
        fn[i] = f_n + np.cross(vw[i], hn) + np.cross(vl[i], hl)
        if fext is not None:
            fl[i] = fl[i] - fext[i][0:3]
            fn[i] = fn[i] - fext[i][3:6]
    tau = np.zeros(M["nv"], dt)
    for i in range(n - 1, -1, -1):
        iv = M["idx_v"][i]
        if M["jtype"][i] == 0:
            tau[iv] = M["axis"][i] @ fn[i]
        else:
            tau[iv:iv + 3], tau[iv + 3:iv + 6] = fl[i], fn[i]
        pa = M["parent"][i]
        if pa >= 0:
            Rf = Rs[i] @ fl[i]
            fl[pa] = fl[pa] + Rf
            fn[pa] = fn[pa] + Rs[i] @ fn[i] + np.cross(ps[i], Rf)
    return tau


def rnea_derivatives(M, q, v, a, fext=None, gravity=True):
    h = 1e-30
    nv = M["nv"]
    dq, dv, da = np.zeros((nv, nv)), np.zeros((nv, nv)), np.zeros((nv, nv))
    for k in range(nv):
        e = np.zeros(nv, complex)
        e[k] = 1j * h
        dq[:, k] = rnea(M, q, v, a, fext, dq_seed=e, gravity=gravity).imag / h
        dv[:, k] = rnea(M, q, v + e, a, fext, gravity=gravity).imag / h
        da[:, k] = rnea(M, q, v, a + e, fext, gravity=gravity).imag / h
    return dq, dv, da


# -------------------------------------------------------------- samples ----
def random_q(M, rng):
    q = np.zeros(M["nq"])
    for i in range(M["njoints"]):
        iq = M["idx_q"][i]
        if M["jtype"][i] == 0:
            q[iq] = rng.uniform(-1.5, 1.5)
        else:
            q[iq:iq + 3] = rng.uniform(-1, 1, 3)
            quat = rng.normal(size=4)
            q[iq + 3:iq + 7] = quat / np.linalg.norm(quat)
    return q


def model_to_json(M):
    return dict(njoints=M["njoints"], nq=M["nq"], nv=M["nv"], floating=M["floating"],
                parent=M["parent"], jtype=M["jtype"], idx_q=M["idx_q"], idx_v=M["idx_v"],
                axis=[a.tolist() for a in M["axis"]],
                plc_R=[R.reshape(-1).tolist() for R in M["plc_R"]],
                plc_p=[p.tolist() for p in M["plc_p"]],
                mass=[b[0] for b in M["body"]], com=[b[1].tolist() for b in M["body"]],
                inertia=[b[2].reshape(-1).tolist() for b in M["body"]],
                q_min=M["q_min"], q_max=M["q_max"], v_max=M["v_max"], u_max=M["u_max"],
                frame_names=[f[0] for f in M["frames"]],
                contacts=[dict(frame=c[0], joint=c[1], R=c[2].reshape(-1).tolist(), p=c[3].tolist())
                          for c in M["contacts"]])


def generate(name, urdf, contact_frames, nsamples, seed):
    M = load_model(urdf, contact_frames)
    rng = np.random.default_rng(seed)
    samples = []
    for s in range(nsamples):
        q = random_q(M, rng)
        v = rng.uniform(-1, 1, M["nv"])
        a = rng.uniform(-1, 1, M["nv"])
        fext = None
        rec = dict(q=q.tolist(), v=v.tolist(), a=a.tolist())
        if contact_frames:
            # contact forces expressed in the LOCAL contact frame, mapped to the
            # parent joint frame (PointContact::computeJointForceFromContactForce)
            fc = rng.uniform(-20, 40, (len(contact_frames), 3))
            fext = np.zeros((M["njoints"], 6))
            for c, (fid, jid, Rc, pc) in enumerate(M["contacts"]):
                fl = Rc @ fc[c]
                fext[jid, 0:3] += fl
                fext[jid, 3:6] += np.cross(pc, fl)
            rec["f"] = fc.tolist()
        tau = rnea(M, q, v, a, fext)
        dq, dv, da = rnea_derivatives(M, q, v, a, fext)
        rec.update(tau=tau.tolist(), dtau_dq=dq.tolist(), dtau_dv=dv.tolist(), dtau_da=da.tolist())
        # impulse model (zero gravity, v = 0): Robot::RNEAImpulse(+Derivatives)
        if contact_frames:
            z = np.zeros(M["nv"])
            rec["tau_impulse"] = rnea(M, q, z, a, fext, gravity=False).tolist()
            dqi, _, dai = rnea_derivatives(M, q, z, a, fext, gravity=False)
            rec["dimp_dq"], rec["dimp_ddv"] = dqi.tolist(), dai.tolist()
        samples.append(rec)
    out = dict(robot=name, model=model_to_json(M), samples=samples,
               note="generated by tests/golden/gen_golden_rbd.py (numpy body-frame RNEA + complex step)")
    with open(os.path.join(HERE, "rbd_%s.json" % name), "w") as f:
        json.dump(out, f)
    print(name, "njoints", M["njoints"], "nq", M["nq"], "nv", M["nv"], "samples", nsamples)
    return M




# ===================================================================== contact
# Frame kinematics + Baumgarte constraint + Lie-group operations of the floating
# base, for the contact path.  Exact pieces come
# from the complex step; the Baumgarte derivative is then ASSEMBLED with the
# reference's own formula (include/idocp/robot/point_contact.hxx:91-144).
def frame_kinematics(M, q, v, a, dq_seed=None):
    """Per contact frame: world position p, world rotation R, LOCAL spatial
    velocity (lin, ang) and LOCAL spatial acceleration (no gravity)."""
    n = M["njoints"]
    cplx = dq_seed is not None or np.iscomplexobj(v) or np.iscomplexobj(a)
    dt = complex if cplx else float
    Rw, pw, vl, vw, al, aw = ([None] * n for _ in range(6))
    for i in range(n):
        R, p = joint_transform(M, i, q, dq_seed)
        pa = M["parent"][i]
        iv = M["idx_v"][i]
        if pa < 0:
            Rw[i], pw[i] = R, p
            pvl, pvw, pal, paw = (np.zeros(3, dt) for _ in range(4))
        else:
            Rw[i], pw[i] = Rw[pa] @ R, pw[pa] + Rw[pa] @ p
            pvl, pvw, pal, paw = vl[pa], vw[pa], al[pa], aw[pa]
        w_ = R.T @ pvw
        v_ = R.T @ (pvl + np.cross(pvw, p))
        aw_ = R.T @ paw
        al_ = R.T @ (pal + np.cross(paw, p))
        if M["jtype"][i] == 0:
            ax = M["axis"][i]
            vJl, vJw = np.zeros(3, dt), ax * v[iv]
            aJl, aJw = np.zeros(3, dt), ax * a[iv]
        else:
            vJl, vJw = np.array(v[iv:iv + 3], dt), np.array(v[iv + 3:iv + 6], dt)
            aJl, aJw = np.array(a[iv:iv + 3], dt), np.array(a[iv + 3:iv + 6], dt)
        vl[i], vw[i] = v_ + vJl, w_ + vJw
        al[i] = al_ + aJl + np.cross(vw[i], vJl) + np.cross(vl[i], vJw)
        aw[i] = aw_ + aJw + np.cross(vw[i], vJw)
    out = []
    for (fid, jid, Rc, pc) in M["contacts"]:
        Rf = Rw[jid] @ Rc
        pf = pw[jid] + Rw[jid] @ pc
        # motion transform joint frame -> contact frame
        fw = Rc.T @ vw[jid]
        fv = Rc.T @ (vl[jid] + np.cross(vw[jid], pc))
        faw = Rc.T @ aw[jid]
        fal = Rc.T @ (al[jid] + np.cross(aw[jid], pc))
        out.append(dict(p=pf, R=Rf, v=np.concatenate([fv, fw]), a=np.concatenate([fal, faw])))
    return out


def frame_derivatives(M, q, v, a):
    """frame_v_partial_dq, a_partial_dq, a_partial_dv, a_partial_da (6 x nv, LOCAL)
    and d p_world / dq (3 x nv) per contact, by complex step."""
    h, nv, nc = 1e-30, M["nv"], len(M["contacts"])
    vdq, adq, adv, ada = (np.zeros((nc, 6, nv)) for _ in range(4))
    pdq = np.zeros((nc, 3, nv))
    for k in range(nv):
        e = np.zeros(nv, complex)
        e[k] = 1j * h
        fq = frame_kinematics(M, q, v, a, dq_seed=e)
        fv = frame_kinematics(M, q, v + e, a)
        fa = frame_kinematics(M, q, v, a + e)
        for c in range(nc):
            vdq[c, :, k] = fq[c]["v"].imag / h
            adq[c, :, k] = fq[c]["a"].imag / h
            pdq[c, :, k] = fq[c]["p"].imag / h
            adv[c, :, k] = fv[c]["a"].imag / h
            ada[c, :, k] = fa[c]["a"].imag / h
    return vdq, adq, adv, ada, pdq


def baumgarte(M, q, v, a, contact_points, time_step):
    """Residual and derivatives exactly as PointContact computes them
    (point_contact.hxx:67-87, 91-144)."""
    fk = frame_kinematics(M, q, v, a)
    vdq, adq, adv, ada, pdq = frame_derivatives(M, q, v, a)
    nc, nv = len(fk), M["nv"]
    C = np.zeros(3 * nc)
    dCdq, dCdv, dCda = (np.zeros((3 * nc, nv)) for _ in range(3))
    wv, wp = 2.0 / time_step, 1.0 / (time_step * time_step)
    for c in range(nc):
        vl, vw = fk[c]["v"][:3].real, fk[c]["v"][3:].real
        acl = fk[c]["a"][:3].real + np.cross(vw, vl)            # classical acceleration, LOCAL
        C[3 * c:3 * c + 3] = acl + wv * vl + wp * (fk[c]["p"].real - contact_points[c])
        J = ada[c]                                              # LOCAL frame Jacobian
        dq = adq[c, :3] + skew(vw) @ vdq[c, :3] + skew(vl) @ vdq[c, 3:]
        dv = adv[c, :3] + skew(vw) @ J[:3] + skew(vl) @ J[3:]
        dq = dq + wv * vdq[c, :3]
        dv = dv + wv * ada[c, :3]
        dq = dq + wp * (fk[c]["R"].real @ J[:3])
        dCdq[3 * c:3 * c + 3], dCdv[3 * c:3 * c + 3], dCda[3 * c:3 * c + 3] = dq, dv, ada[c, :3]
    return C, dCdq, dCdv, dCda, fk, (vdq, adq, adv, ada, pdq)


# ---- SE(3) x R^n Lie operations (pinocchio::integrate / difference / dDifference)
def exp3(w):
    t = np.linalg.norm(w)
    K = skew(w)
    if t < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(t) / t * K + (1 - np.cos(t)) / (t * t) * K @ K


def log3(R):
    c = min(1.0, max(-1.0, (np.trace(R) - 1) / 2))
    t = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if t < 1e-10:
        return 0.5 * w
    return t / (2 * np.sin(t)) * w


def Vmat(w):
    t = np.linalg.norm(w)
    K = skew(w)
    if t < 1e-10:
        return np.eye(3) + 0.5 * K + K @ K / 6
    return np.eye(3) + (1 - np.cos(t)) / (t * t) * K + (t - np.sin(t)) / (t ** 3) * K @ K


def R_to_quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    z = np.sqrt(max(0.0, 1 - R[0, 0] - R[1, 1] + R[2, 2])) / 2
    x, y, z = np.copysign(x, R[2, 1] - R[1, 2]), np.copysign(y, R[0, 2] - R[2, 0]), np.copysign(z, R[1, 0] - R[0, 1])
    return np.array([x, y, z, w])


def integrate(M, q, dv):
    """q (+) dv with the free-flyer convention of pinocchio (SE3 exponential)."""
    out = np.array(q, float)
    for i in range(M["njoints"]):
        iq, iv = M["idx_q"][i], M["idx_v"][i]
        if M["jtype"][i] == 0:
            out[iq] = q[iq] + dv[iv]
        else:
            R = quat_to_R(*q[iq + 3:iq + 7])
            vl, w = dv[iv:iv + 3], dv[iv + 3:iv + 6]
            out[iq:iq + 3] = q[iq:iq + 3] + R @ (Vmat(w) @ vl)
            out[iq + 3:iq + 7] = R_to_quat(R @ exp3(w))
    return out


def difference(M, q0, q1):
    """q1 (-) q0 = log(q0^{-1} q1)  (tangent at q0)."""
    out = np.zeros(M["nv"])
    for i in range(M["njoints"]):
        iq, iv = M["idx_q"][i], M["idx_v"][i]
        if M["jtype"][i] == 0:
            out[iv] = q1[iq] - q0[iq]
        else:
            R0, R1 = quat_to_R(*q0[iq + 3:iq + 7]), quat_to_R(*q1[iq + 3:iq + 7])
            R = R0.T @ R1
            p = R0.T @ (q1[iq:iq + 3] - q0[iq:iq + 3])
            w = log3(R)
            out[iv:iv + 3] = np.linalg.solve(Vmat(w), p)
            out[iv + 3:iv + 6] = w
    return out


def ddifference(M, q0, q1, arg):
    """Jacobian of difference(q0, q1) w.r.t. a tangent perturbation of q0 (arg=0)
    or q1 (arg=1), by Richardson-extrapolated central differences."""
    nv = M["nv"]
    J = np.zeros((nv, nv))

    def f(e):
        return difference(M, integrate(M, q0, e), q1) if arg == 0 else difference(M, q0, integrate(M, q1, e))

    for k in range(nv):
        e = np.zeros(nv)
        d = []
        for hh in (1e-3, 5e-4):
            e[k] = hh
            fp = f(e)
            e[k] = -hh
            fm = f(e)
            d.append((fp - fm) / (2 * hh))
        J[:, k] = (4 * d[1] - d[0]) / 3
    return J


def generate_contact(name, urdf, contact_frames, nsamples, seed):
    M = load_model(urdf, contact_frames)
    rng = np.random.default_rng(seed)
    time_step = 0.05
    samples = []
    for s in range(nsamples):
        q = random_q(M, rng)
        v = rng.uniform(-1, 1, M["nv"])
        a = rng.uniform(-1, 1, M["nv"])
        cp = rng.uniform(-0.5, 0.5, (len(contact_frames), 3))
        C, dCdq, dCdv, dCda, fk, (vdq, adq, adv, ada, pdq) = baumgarte(M, q, v, a, cp, time_step)
        _, _, Mm = rnea_derivatives(M, q, v, a)
        KKT = np.block([[Mm, dCda.T], [dCda, np.zeros((dCda.shape[0],) * 2)]])
        q1 = integrate(M, q, rng.uniform(-0.6, 0.6, M["nv"]))
        dvv = rng.uniform(-0.5, 0.5, M["nv"])
        rec = dict(q=q.tolist(), v=v.tolist(), a=a.tolist(), contact_points=cp.tolist(), time_step=time_step,
                   frame_p=[f["p"].real.tolist() for f in fk], frame_R=[f["R"].real.reshape(-1).tolist() for f in fk],
                   frame_v=[f["v"].real.tolist() for f in fk], frame_a=[f["a"].real.tolist() for f in fk],
                   v_partial_dq=vdq.tolist(), a_partial_dq=adq.tolist(), a_partial_dv=adv.tolist(),
                   a_partial_da=ada.tolist(), p_partial_dq=pdq.tolist(),
                   C=C.tolist(), dCdq=dCdq.tolist(), dCdv=dCdv.tolist(), dCda=dCda.tolist(),
                   MJtJinv=np.linalg.inv(KKT).tolist(),
                   q1=q1.tolist(), dv=dvv.tolist(), q_plus_dv=integrate(M, q, dvv).tolist(),
                   q1_minus_q=difference(M, q, q1).tolist(),
                   dDiff_arg0=ddifference(M, q, q1, 0).tolist(), dDiff_arg1=ddifference(M, q, q1, 1).tolist())
        samples.append(rec)
    out = dict(robot=name, samples=samples,
               note="generated by tests/golden/gen_golden_rbd.py generate_contact (complex step / Richardson differences)")
    with open(os.path.join(HERE, "contact_%s.json" % name), "w") as f:
        json.dump(out, f)
    print("contact", name, "samples", nsamples)


if __name__ == "__main__":
    generate("iiwa14", os.path.join(HERE, "urdf", "iiwa14.urdf"), (), 6, 20240)
    generate("anymal", os.path.join(HERE, "urdf", "anymal.urdf"), (14, 24, 34, 44), 6, 20250)
    generate_contact("anymal", os.path.join(HERE, "urdf", "anymal.urdf"), (14, 24, 34, 44), 4, 20251)
