"""`.g2o` text graphs in the dialect the reference emits.

Writer of the reference: g2o/core/optimizable_graph.cpp:589-623 (parameters, vertices by id, edges),
tags from g2o/types/types_slam3d.cpp:37-48 and g2o/types/types_dyn_slam3d.cpp:44-51; dumps at
src/Optimizer.cc:806-808 and :1934-1936.  The reference streams numbers with 6 significant digits (lossy);
this writer uses 17 and the reader accepts both.  Robust kernels are not serialised: the Huber deltas are
arguments (1e-4 for every edge class in the reference, src/Optimizer.cc:213,1352).

    PARAMS_SE3OFFSET 0 x y z qx qy qz qw
    VERTEX_SE3:QUAT id x y z qx qy qz qw            camera pose T_wc or object motion H
    VERTEX_TRACKXYZ id x y z
    EDGE_SE3_PRIOR id paramId x y z qx qy qz qw <21 upper-triangular information>
    EDGE_SE3:QUAT i j x y z qx qy qz qw <21>
    EDGE_SE3_TRACKXYZ camId ptId paramId mx my mz i00 i01 i02 i11 i12 i22
    EDGE_SE3_MOTION p1 p2 H mx my mz i00 i01 i02 i11 i12 i22
"""
import numpy as np

from .synth import BAGraph, HUBER_DELTA


def _quat_from_R(R):
    """Eigen::Quaterniond(Matrix3d) (Shepperd), returned as (qx, qy, qz, qw) with qw >= 0 as g2o writes it."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _R_from_quat(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _se3_fields(p12):
    R = np.asarray(p12[:9], float).reshape(3, 3)
    return list(p12[9:12]) + list(_quat_from_R(R))


def _pose12(v7):
    return np.concatenate([_R_from_quat(np.asarray(v7[3:7], float)).ravel(), np.asarray(v7[:3], float)])


def _upper(M, n):
    M = np.asarray(M, float).reshape(n, n)
    return [M[i, j] for i in range(n) for j in range(i, n)]


def _full(vals, n):
    M = np.zeros((n, n)); k = 0
    for i in range(n):
        for j in range(i, n):
            M[i, j] = M[j, i] = float(vals[k]); k += 1
    return M


def _fmt(v, digits):
    return " ".join(f"{float(x):.{digits}g}" for x in v)


def write_g2o(path, g: BAGraph, pose=None, point=None, digits=17):
    """Write ``g`` (optionally with other estimates, e.g. after optimisation).  Vertex ids: poses 0..P-1, points P..P+L-1."""
    pose = g.pose if pose is None else np.asarray(pose); point = g.point if point is None else np.asarray(point)
    P = g.n_pose
    with open(path, "w") as f:
        f.write("PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1\n")
        for i in range(P):
            f.write(f"VERTEX_SE3:QUAT {i} {_fmt(_se3_fields(pose[i]), digits)}\n")
        for j in range(g.n_point):
            f.write(f"VERTEX_TRACKXYZ {P + j} {_fmt(point[j], digits)}\n")
        for e in range(g.n_prior):
            f.write(f"EDGE_SE3_PRIOR {int(g.pr_pose[e])} 0 {_fmt(_se3_fields(g.pr_z[e]), digits)} {_fmt(_upper(g.pr_info[e], 6), digits)}\n")
        for e in range(g.n_ep):
            f.write(f"EDGE_SE3:QUAT {int(g.ep_i[e])} {int(g.ep_j[e])} {_fmt(_se3_fields(g.ep_z[e]), digits)} {_fmt(_upper(g.ep_info[e], 6), digits)}\n")
        for e in range(g.n_eb):
            w = g.eb_w[e]
            f.write(f"EDGE_SE3_TRACKXYZ {int(g.eb_pose[e])} {P + int(g.eb_point[e])} 0 {_fmt(g.eb_z[:, e], digits)} {_fmt([w, 0, 0, w, 0, w], digits)}\n")
        for e in range(g.n_et):
            w = g.et_w[e]
            f.write(f"EDGE_SE3_MOTION {P + int(g.et_p1[e])} {P + int(g.et_p2[e])} {int(g.et_pose[e])} {_fmt(g.et_z[:, e], digits)} {_fmt([w, 0, 0, w, 0, w], digits)}\n")


def read_g2o(path, huber_eb=HUBER_DELTA, huber_et=HUBER_DELTA, huber_ep=HUBER_DELTA) -> BAGraph:
    """Read a graph in the reference's dialect.  SE3 vertices become poses and TRACKXYZ vertices points, each in
    ascending id order; only isotropic point-edge information (w * I3, as the reference builds it) is supported."""
    se3, xyz = {}, {}
    pr, ep, eb, et = [], [], [], []
    for line in open(path):
        t = line.split()
        if not t or t[0] in ("PARAMS_SE3OFFSET", "FIX"):
            continue
        tag, v = t[0], t[1:]
        if tag == "VERTEX_SE3:QUAT":
            se3[int(v[0])] = _pose12([float(x) for x in v[1:8]])
        elif tag == "VERTEX_TRACKXYZ":
            xyz[int(v[0])] = np.array([float(x) for x in v[1:4]])
        elif tag == "EDGE_SE3_PRIOR":
            pr.append((int(v[0]), _pose12([float(x) for x in v[2:9]]), _full(v[9:30], 6)))
        elif tag == "EDGE_SE3:QUAT":
            ep.append((int(v[0]), int(v[1]), _pose12([float(x) for x in v[2:9]]), _full(v[9:30], 6)))
        elif tag in ("EDGE_SE3_TRACKXYZ", "EDGE_SE3_MOTION"):
            info = _full(v[6:12], 3)
            if not (np.allclose(info, info[0, 0] * np.eye(3), rtol=1e-5, atol=0)):
                raise ValueError(f"{tag}: only isotropic information is supported")
            rec = (int(v[0]), int(v[1]), int(v[2]), np.array([float(x) for x in v[3:6]]), info[0, 0])
            (eb if tag == "EDGE_SE3_TRACKXYZ" else et).append(rec)
        else:
            raise ValueError(f"unknown tag {tag}")
    pid = {k: i for i, k in enumerate(sorted(se3))}
    lid = {k: i for i, k in enumerate(sorted(xyz))}
    I = lambda a: np.array(a, np.int32).reshape(-1)
    D = lambda a, shape: np.array(a, float).reshape(shape)
    return BAGraph(
        pose=D([se3[k] for k in sorted(se3)], (-1, 12)), point=D([xyz[k] for k in sorted(xyz)], (-1, 3)),
        eb_pose=I([pid[r[0]] for r in eb]), eb_point=I([lid[r[1]] for r in eb]),
        eb_z=np.ascontiguousarray(D([r[3] for r in eb], (-1, 3)).T), eb_w=D([r[4] for r in eb], (-1,)),
        et_p1=I([lid[r[0]] for r in et]), et_p2=I([lid[r[1]] for r in et]), et_pose=I([pid[r[2]] for r in et]),
        et_z=np.ascontiguousarray(D([r[3] for r in et], (-1, 3)).T), et_w=D([r[4] for r in et], (-1,)),
        ep_i=I([pid[r[0]] for r in ep]), ep_j=I([pid[r[1]] for r in ep]), ep_z=D([r[2] for r in ep], (-1, 12)), ep_info=D([r[3].ravel() for r in ep], (-1, 36)),
        pr_pose=I([pid[r[0]] for r in pr]), pr_z=D([r[1] for r in pr], (-1, 12)), pr_info=D([r[2].ravel() for r in pr], (-1, 36)),
        huber_eb=huber_eb, huber_et=huber_et, huber_ep=huber_ep)
