#!/usr/bin/env python
"""Golden vectors for the KITTI AP evaluator: run the REFERENCE's own evaluate_object binary (built from
/root/reference/src/utils/kitti-eval/cpp by oracle/ref_build/Makefile into oracle/_ref/) on a seeded synthetic
label/detection set and store inputs + the AP values / precision samples it wrote -> tests/golden/kitti_eval.npz.

    make -C oracle/ref_build && python tests/golden/make_golden_kitti_eval.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kitti_eval_util import make_dataset, run_reference_binary, write_dataset  # noqa: E402


def main():
    out = {}
    for seed, n in ((0, 300), (1, 120)):
        gts, dets = make_dataset(seed, n)
        with tempfile.TemporaryDirectory() as tmp:
            write_dataset(tmp, gts, dets)
            ap, det11 = run_reference_binary(tmp, n)
        out[f"ap{seed}"] = ap; out[f"prec11_{seed}"] = det11; out[f"n{seed}"] = np.array([n])
        print(f"seed {seed}: AP\n", ap)
    np.savez_compressed(os.path.join(HERE, "kitti_eval.npz"), **out)


if __name__ == "__main__":
    main()
