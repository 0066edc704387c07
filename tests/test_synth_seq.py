"""The synthetic sequence generator (vdo_slam_amd/synth_seq.py) is self-consistent: the rendered flow of every object pixel
is the projection of the object's rigid motion (rotation included), objects exist only inside their frame range."""
import numpy as np

from vdo_slam_amd import synth_seq as SQ
from vdo_slam_amd.synth import KITTI_K


def test_turning_boxes_flow_is_their_rigid_motion_and_lifetimes_are_respected():
    Ts = SQ.camera_poses(8)
    objs = SQ.survey_objects(leave_at=3, enter_at=5)
    fx, fy, cx, cy = KITTI_K
    seen = {}
    for k in (0, 2, 3, 5, 6):
        fr = SQ.render_frame(k, Ts, objs, w=621, h=188, K4=(fx / 2, fy / 2, cx / 2, cy / 2))
        labels = set(np.unique(fr["mask"]).tolist()) - {0}
        seen[k] = labels
        for lab in labels:
            m = fr["mask"] == lab
            vv, uu = np.nonzero(m)
            z = fr["depth_true"][m]
            Xc = np.c_[(uu - cx / 2) * z / (fx / 2), (vv - cy / 2) * z / (fy / 2), z]
            Xw = Xc @ Ts[k][:3, :3].T + Ts[k][:3, 3]
            Hm = SQ.object_motion(objs[lab - 1], k)
            Xn = Xw @ Hm[:3, :3].T + Hm[:3, 3]
            T1 = np.linalg.inv(Ts[k + 1])
            Xc1 = Xn @ T1[:3, :3].T + T1[:3, 3]
            un = fx / 2 * Xc1[:, 0] / Xc1[:, 2] + cx / 2; vn = fy / 2 * Xc1[:, 1] / Xc1[:, 2] + cy / 2
            assert np.abs(un - uu - fr["flow"][m][:, 0]).max() < 1e-4 and np.abs(vn - vv - fr["flow"][m][:, 1]).max() < 1e-4
    assert 2 in seen[0] and 2 in seen[2] and 2 not in seen[3] and 2 not in seen[5]
    assert 5 not in seen[3] and 5 in seen[5] and 5 in seen[6]
    H = SQ.object_motion(objs[4], 6)
    assert abs(np.arccos((np.trace(H[:3, :3]) - 1) / 2) - 0.05) < 1e-12            # the largest yaw rate of SURVEY 8d
    assert np.allclose(SQ.object_motion(objs[2], 1)[:3, :3], np.eye(3))               # one object does not turn
