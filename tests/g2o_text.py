"""A few lines that read g2o's text format into plain numpy literals - the tests' own reader, independent of the
product's svi_ba_load_g2o, used to build the ORACLE's graph from the same file through its API.
Formats (g2o slam3d types as registered upstream; the acceleration edge is the reference's,
edge_se3_linear_acceleration.cpp:35-103):
  VERTEX_SE3:QUAT id x y z qx qy qz qw | VERTEX_TRACKXYZ id x y z | FIX ids... | PARAMS_SE3OFFSET id x y z qx qy qz qw
  PARAMS_CAMERACALIB id x y z qx qy qz qw fx fy cx cy | EDGE_SE3:QUAT i j x y z qx qy qz qw + 21
  EDGE_SE3_TRACKXYZ / EDGE_PROJECT_DEPTH / EDGE_PROJECT_DISPARITY p l pid m0 m1 m2 + 6
  EDGE_POINTXYZ i j x y z + 6 | EDGE_SE3_LINEAR_ACCELERATION p pid ax ay az + 6"""
import numpy as np


def quat_R(x, y, z, w):
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose12(v7):
    return np.concatenate([quat_R(*v7[3:7]).reshape(9), v7[:3]])


def read(path):
    g = dict(offsets={}, cams={}, poses=[], lms=[], fixed=[], se3=[], proj=[], lmlm=[], accel=[])
    kind = {"EDGE_SE3_TRACKXYZ": 0, "EDGE_PROJECT_DEPTH": 1, "EDGE_PROJECT_DISPARITY": 2}
    for line in open(path):
        w = line.split()
        if not w or w[0].startswith("#"):
            continue
        tag, v = w[0], w[1:]
        f = lambda a: np.array([float(x) for x in a])  # noqa: E731
        if tag == "PARAMS_SE3OFFSET":
            g["offsets"][int(v[0])] = pose12(f(v[1:8]))
        elif tag == "PARAMS_CAMERACALIB":
            g["cams"][int(v[0])] = f(v[8:12])
        elif tag == "VERTEX_SE3:QUAT":
            g["poses"].append((int(v[0]), pose12(f(v[1:8]))))
        elif tag == "VERTEX_TRACKXYZ":
            g["lms"].append((int(v[0]), f(v[1:4])))
        elif tag == "FIX":
            g["fixed"] += [int(x) for x in v]
        elif tag == "EDGE_SE3:QUAT":
            g["se3"].append((int(v[0]), int(v[1]), pose12(f(v[2:9])), f(v[9:30])))
        elif tag in kind:
            g["proj"].append((kind[tag], int(v[0]), int(v[1]), int(v[2]), f(v[3:6]), f(v[6:12])))
        elif tag == "EDGE_POINTXYZ":
            g["lmlm"].append((int(v[0]), int(v[1]), f(v[2:5]), f(v[5:11])))
        elif tag == "EDGE_SE3_LINEAR_ACCELERATION":
            g["accel"].append((int(v[0]), int(v[1]), f(v[2:5]), f(v[5:11])))
        else:
            raise ValueError("unknown tag " + tag)
    return g


def build(ba, g):
    """the graph of `g` through the BundleAdjuster / OracleBA method surface, edges in file order per kind; landmark
    edges robust (g2o does not serialise kernels, the reference attaches Cauchy to all of them), pose edges not"""
    fixed = set(g["fixed"])
    if 3 in g["offsets"]:
        ba.set_imu_offset(g["offsets"][3])
    for pid, T in g["poses"]:
        ba.add_pose(pid, T, pid in fixed)
    for lid, p in g["lms"]:
        ba.add_landmark(lid, p, lid in fixed)
    for i, j, Z, info in g["se3"]:
        ba.add_edge_se3(i, j, Z, info, False)
    for t, p, l, pid, z, info in g["proj"]:
        ba.add_edges_bulk([t], [p], [l], z[None], info[None], [1])
    for i, j, z, info in g["lmlm"]:
        ba.add_edge_lm_lm(i, j, z, info, True)
    for p, pid, a, info in g["accel"]:
        ba.add_edge_accel(p, a, g["offsets"].get(pid), info)
    return ba
