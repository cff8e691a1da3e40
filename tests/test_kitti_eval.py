"""Results writer + KITTI AP: the native evaluator (sqd_kitti_ap behind results.evaluate) against
the REFERENCE's own evaluate_object binary -- live when oracle/_ref/evaluate_object exists (built by
oracle/ref_build/Makefile from the sources under /root/reference), and always against the committed golden vectors
that binary produced (tests/golden/make_golden_kitti_eval.py)."""
import os

import numpy as np
import pytest

from kitti_eval_util import REF_BIN, make_dataset, run_reference_binary, write_dataset
from squeezedet_pytorch_amd import results as R


def _mine(root, n):
    aps = R.evaluate(os.path.join(root, "results"), os.path.join(root, "training", "label_2"), os.path.join(root, "set.txt"))
    ap = np.array([[aps[f"{c}_{d}"] for d in ("easy", "moderate", "hard")] for c in R.KITTI_CLASS_NAMES])
    return aps, ap


@pytest.mark.parametrize("seed,n", [(0, 300), (1, 120)])
def test_ap_matches_reference_golden(tmp_path, golden_dir, seed, n):
    g = np.load(os.path.join(golden_dir, "kitti_eval.npz"))
    gts, dets = make_dataset(seed, n)
    write_dataset(str(tmp_path), gts, dets)
    aps, ap = _mine(str(tmp_path), n)
    assert np.array_equal(ap, g[f"ap{seed}"])                      # same 6 significant digits the reference prints
    assert aps["mAP"] == pytest.approx(g[f"ap{seed}"].mean(), rel=1e-12)
    # the 11 sampled precisions the binary wrote with %f
    mine11 = np.array([[l.split() for l in open(tmp_path / "results" / f"stats_{c}_detection.txt").read().strip().split("\n")]
                       for c in ("car", "pedestrian", "cyclist")], dtype=np.float64)
    assert np.array_equal(mine11, g[f"prec11_{seed}"])
    assert ap.max() > 0.5 and ap.min() >= 0.0                      # the fixture exercises real PR curves


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="reference evaluator not built (make -C oracle/ref_build)")
@pytest.mark.parametrize("seed,n", [(7, 200), (8, 12), (9, 450)])
def test_ap_matches_reference_binary_live(tmp_path, seed, n):
    gts, dets = make_dataset(seed, n)
    if seed == 8:                                                  # no cyclist detections at all: class not evaluated
        dets = [[o for o in d if o["type"] != "cyclist"] for d in dets]
    write_dataset(str(tmp_path), gts, dets)
    ap_ref, p11_ref = run_reference_binary(str(tmp_path), n)
    for f in os.listdir(tmp_path / "results"):                      # drop the binary's stats so ours are the ones compared
        if f.startswith("stats_"):
            os.remove(tmp_path / "results" / f)
    aps, ap = _mine(str(tmp_path), n)
    assert np.array_equal(ap, ap_ref)
    if seed == 8:
        assert aps["Cyclist_easy"] == 0.0 and not os.path.exists(tmp_path / "results" / "stats_cyclist_ap.txt")


def test_save_results_format_and_roundtrip(tmp_path):
    res = [{"image_meta": {"image_id": "000007"}, "class_ids": np.array([0, 2]), "scores": np.array([0.91234, 0.5], np.float32),
            "boxes": np.array([[10.126, 20.5, 110.0, 220.994], [0, 1, 2, 3]], np.float32)},
           {"image_meta": {"image_id": "000008"}}]
    R.save_results(res, str(tmp_path))
    txt = open(tmp_path / "data" / "000007.txt").read().split("\n")
    assert txt[0] == "car -1 -1 0 10.13 20.50 110.00 220.99 0 0 0 0 0 0 0 0.912"
    assert txt[1] == "cyclist -1 -1 0 0.00 1.00 2.00 3.00 0 0 0 0 0 0 0 0.500"
    assert open(tmp_path / "data" / "000008.txt").read() == ""
    cls, box, score = R.read_result_file(str(tmp_path / "data" / "000007.txt"))
    assert cls.tolist() == [0, 2] and score.tolist() == [0.912, 0.5] and box[0].tolist() == [10.13, 20.5, 110.0, 220.99]


def test_perfect_and_empty_detections():
    kind = np.array([0, 1, 5], np.int32)
    box = np.array([[100, 100, 200, 180], [300, 120, 340, 220], [500, 100, 600, 200]], np.float64)
    gt = (kind, box, np.zeros(3), np.zeros(3, np.int32))
    perfect = (np.array([0, 1], np.int32), box[:2].copy(), np.array([0.9, 0.8]))
    ap, prec, ev = R.kitti_ap([gt] * 60, [perfect] * 60)
    assert ev.tolist() == [True, True, False]
    assert np.allclose(ap[0], 1.0) and np.allclose(ap[1], 1.0) and np.all(ap[2] == 0)
    # a false positive inside a DontCare region is forgiven; outside it costs precision
    fp_in_dc = (np.array([0, 0], np.int32), np.array([[100, 100, 200, 180], [510, 110, 590, 190]], np.float64), np.array([0.9, 0.95]))
    ap2, _, _ = R.kitti_ap([gt] * 60, [fp_in_dc] * 60)
    assert np.allclose(ap2[0], 1.0)
    fp_out = (np.array([0, 0], np.int32), np.array([[100, 100, 200, 180], [700, 110, 790, 190]], np.float64), np.array([0.9, 0.95]))
    ap3, _, _ = R.kitti_ap([gt] * 60, [fp_out] * 60)
    assert np.all(ap3[0] < 0.75)
    ap4, _, ev4 = R.kitti_ap([gt], [(np.zeros(0, np.int32), np.zeros((0, 4)), np.zeros(0))])
    assert not ev4.any() and np.all(ap4 == 0)
