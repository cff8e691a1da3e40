"""Synthetic KITTI label / detection sets and a runner for the reference's evaluator binary (test infrastructure)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "evaluate_object")
TYPES = ["Car", "Pedestrian", "Cyclist", "Van", "Person_sitting", "DontCare", "Truck", "Misc"]


def make_dataset(seed, n_images):
    """-> (gts, dets): per image lists of dicts with KITTI fields.  Detections are jittered copies of a subset of the
    boxes (so every difficulty sees TPs, FPs, misses, neighbour classes and DontCare hits) plus random clutter.
    Coordinates carry 2 decimals and scores 3, exactly what the results writer emits."""
    rs = np.random.RandomState(seed)
    gts, dets = [], []
    for _ in range(n_images):
        g, d = [], []
        for _ in range(rs.randint(2, 9)):
            t = TYPES[rs.choice(len(TYPES), p=[.35, .2, .12, .08, .05, .1, .05, .05])]
            h = float(np.exp(rs.uniform(np.log(12), np.log(200)))); w = h * rs.uniform(0.4, 2.2)
            x1 = rs.uniform(0, 1200 - w); y1 = rs.uniform(100, 370 - min(h, 250))
            box = [round(x1, 2), round(y1, 2), round(x1 + w, 2), round(y1 + h, 2)]
            g.append({"type": t, "trunc": round(float(rs.choice([0, 0, 0.1, 0.2, 0.4, 0.7])), 2), "occ": int(rs.choice([0, 0, 1, 2, 3])),
                      "alpha": round(float(rs.uniform(-3, 3)), 2), "box": box})
            if rs.rand() < 0.85:
                jit = rs.normal(0, 0.02 * min(w, h), 4)
                cls = t if t in TYPES[:3] and rs.rand() < 0.97 else TYPES[rs.randint(0, 3)]
                b = [round(float(box[k] + jit[k]), 2) for k in range(4)]
                if b[2] > b[0] + 1 and b[3] > b[1] + 1:
                    d.append({"type": cls.lower(), "box": b, "score": round(float(rs.beta(6, 1.5)), 3)})
        for _ in range(rs.randint(0, 4)):
            h = float(np.exp(rs.uniform(np.log(15), np.log(150)))); w = h * rs.uniform(0.5, 2.0)
            x1 = rs.uniform(0, 1200 - w); y1 = rs.uniform(100, 200)
            d.append({"type": TYPES[rs.randint(0, 3)].lower(), "box": [round(x1, 2), round(y1, 2), round(x1 + w, 2), round(y1 + h, 2)],
                      "score": round(float(rs.beta(1.5, 6)), 3)})
        gts.append(g); dets.append(d)
    return gts, dets


def write_dataset(root, gts, dets):
    """KITTI layout: <root>/training/label_2/<id>.txt, <root>/set.txt, <root>/results/data/<id>.txt"""
    lab = os.path.join(root, "training", "label_2"); res = os.path.join(root, "results", "data")
    os.makedirs(lab); os.makedirs(res)
    ids = ["%06d" % i for i in range(len(gts))]
    with open(os.path.join(root, "set.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    for i, g, d in zip(ids, gts, dets):
        with open(os.path.join(lab, i + ".txt"), "w") as f:
            for o in g:
                f.write("{} {:.2f} {} {:.2f} {:.2f} {:.2f} {:.2f} {:.2f} 1.5 1.6 3.9 1.0 1.5 20.0 0.1\n".format(
                    o["type"], o["trunc"], o["occ"], o["alpha"], *o["box"]))
        with open(os.path.join(res, i + ".txt"), "w") as f:
            for o in d:
                f.write("{} -1 -1 0 {:.2f} {:.2f} {:.2f} {:.2f} 0 0 0 0 0 0 0 {:.3f}\n".format(o["type"], *o["box"], o["score"]))
    return ids


def run_reference_binary(root, n):
    """-> (ap [3,3] with zeros for unevaluated classes, prec11 [3,3,11]) parsed from the stats files the binary writes."""
    cmd = [REF_BIN, os.path.join(root, "training"), os.path.join(root, "set.txt"), os.path.join(root, "results"), str(n)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root)
    ap = np.zeros((3, 3)); p11 = np.zeros((3, 3, 11))
    for c, name in enumerate(("car", "pedestrian", "cyclist")):
        p = os.path.join(root, "results", f"stats_{name}_ap.txt")
        if os.path.exists(p):
            ap[c] = [float(l.split("=")[1]) for l in open(p).read().split()]
            rows = [l.split() for l in open(os.path.join(root, "results", f"stats_{name}_detection.txt")).read().strip().split("\n")]
            p11[c] = np.array(rows, dtype=np.float64)
    return ap, p11
