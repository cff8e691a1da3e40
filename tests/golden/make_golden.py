#!/usr/bin/env python
"""Generate the committed golden vectors in tests/golden/*.npz by running the REFERENCE
(hazenai/SqueezeDet-PyTorch, imported read-only from /root/reference/src) on seeded
synthetic inputs.  Runs only in the build container (the reference never travels); the
resulting small .npz files are data (inputs are regenerable from seeds, outputs stored).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference: ``model.squeezedet`` (SqueezeDetBase, SqueezeDet,
SqueezeDetWithLoss, PredictionResolver, Loss), ``model.modules``, ``utils.boxes``
(generate_anchors, compute_deltas, boxes_postprocess -- needs a ``cv2`` module object to
exist at import time; cv2 is never called by these functions), ``engine.detector``
(Detector.filter -- needs ``torchvision.ops.nms``, which is absent in the container: it is
bound to ``oracle.nms`` so that the reference's own filter control flow runs around the
restated NMS; NMS arithmetic itself therefore stays "parity unpinned", see oracle/__init__.py).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import oracle  # noqa: E402
import squeezedet_pytorch_amd as sqd  # noqa: E402
from squeezedet_pytorch_amd import synthetic  # noqa: E402

REF = "/root/reference/src"


def import_reference():
    sys.path.insert(0, REF)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")

    def _nms(boxes, scores, thr):
        keep = oracle.nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), thr)
        return torch.from_numpy(keep)

    tv_ops.nms = _nms
    tv.ops = tv_ops
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", tv_ops)
    from model import squeezedet as ref_model
    from model import modules as ref_modules
    from utils import boxes as ref_boxes
    from engine import detector as ref_detector
    return ref_model, ref_modules, ref_boxes, ref_detector


def ref_cfg(arch, input_size, dropout_prob=0.0):
    cfg = sqd.make_cfg(arch=arch, input_size=input_size, device="cpu", dropout_prob=dropout_prob)
    return cfg


def main():
    ref_model, ref_modules, ref_boxes, ref_detector = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- 1. anchors (src/utils/boxes.py:37-67) -------------------------------------------
    a_ref = ref_boxes.generate_anchors((24, 78), (384, 1248), oracle.KITTI_ANCHOR_SEED)
    a_small = ref_boxes.generate_anchors((4, 6), (64, 96), oracle.KITTI_ANCHOR_SEED)
    np.savez_compressed(os.path.join(HERE, "anchors.npz"), kitti=a_ref, small=a_small)

    # ---- 2. backbone forward, small input, both architectures ----------------------------
    small = (64, 96)
    out = {}
    for arch in ("squeezedet", "squeezedetplus"):
        cfg = ref_cfg(arch, small)
        m = ref_model.SqueezeDet(cfg).eval()
        sd = synthetic.make_state_dict(arch, seed=1234)
        m.load_state_dict(sd, strict=True)
        x = synthetic.make_images(2, small, seed=3)
        with torch.no_grad():
            pred = m.base(x)
            det = m({"image": x})
            # per-layer taps of the reference's nn.Sequential
            taps = {}
            y = x
            for i, layer in enumerate(m.base.features):
                y = layer(y)
                if i in (0, 2, 3, 5, 14):
                    taps[i] = y.clone()
        out[f"{arch}_pred"] = pred.numpy()
        out[f"{arch}_class_ids"] = det["class_ids"].numpy()
        out[f"{arch}_scores"] = det["scores"].numpy()
        out[f"{arch}_boxes"] = det["boxes"].numpy()
        for i, t in taps.items():
            out[f"{arch}_feat{i}_sum"] = np.array([t.double().sum().item(), t.double().abs().sum().item()])
        out[f"{arch}_feat3"] = taps[3].numpy()[:, ::8]          # a slice of Fire-1 output
    np.savez_compressed(os.path.join(HERE, "backbone_small.npz"), **out)

    # ---- 3. full-size KITTI forward: sampled rows + per-layer checksums -------------------
    cfg = ref_cfg("squeezedet", (384, 1248))
    m = ref_model.SqueezeDet(cfg).eval()
    m.load_state_dict(synthetic.make_state_dict("squeezedet", seed=1234), strict=True)
    x = synthetic.make_images(1, (384, 1248), seed=0)
    with torch.no_grad():
        pred = m.base(x)
        det = m({"image": x})
        sums = []
        y = x
        for layer in m.base.features:
            y = layer(y)
            sums.append([y.double().sum().item(), y.double().abs().sum().item()])
    sc = det["scores"][0].numpy()
    top = np.argsort(-sc, kind="stable")[:256]
    np.savez_compressed(os.path.join(HERE, "kitti_full.npz"),
                        pred_rows=pred[0, ::257].numpy(), layer_sums=np.array(sums),
                        top_idx=top, top_scores=sc[top], top_class_ids=det["class_ids"][0].numpy()[top],
                        top_boxes=det["boxes"][0].numpy()[top],
                        pred_top=pred[0].numpy()[top])

    # ---- 4. decode / inference head on synthetic pred (rows I, J) -------------------------
    rs = np.random.RandomState(11)
    pred = torch.from_numpy((rs.standard_normal((2, 16848, 8)) * np.array([2, 2, 2, 2, .4, .4, .4, .4])).astype(np.float32))
    res = ref_model.PredictionResolver(cfg, log_softmax=True)
    with torch.no_grad():
        probs, logp, scores, deltas, boxes = res(pred)
        p2 = probs * scores
        ids = torch.argmax(p2, dim=2)
        best = torch.max(p2, dim=2)[0]
    sel = np.arange(0, 16848, 13)
    np.savez_compressed(os.path.join(HERE, "decode.npz"), seed=11, sel=sel,
                        probs=probs.numpy()[:, sel], logp=logp.numpy()[:, sel], scores=scores.numpy()[:, sel],
                        boxes=boxes.numpy()[:, sel], class_ids=ids.numpy()[:, sel], best=best.numpy()[:, sel],
                        best_sum=np.array([best.double().sum().item()]), box_sum=np.array([boxes.double().sum().item()]))

    # ---- 5. Detector.filter (row K): reference control flow around oracle.nms --------------
    det_cfg = ref_cfg("squeezedet", (384, 1248))
    filt = ref_detector.Detector.filter
    fake_self = types.SimpleNamespace(cfg=det_cfg)
    cases = {}
    for b in range(2):
        d = filt(fake_self, {"class_ids": ids[b], "scores": best[b], "boxes": boxes[b]})
        cases[f"syn{b}"] = d
    # the full-size KITTI forward of step 3
    cases["kitti"] = filt(fake_self, {k: v[0] for k, v in det.items()})
    # everything below threshold -> None
    lo = {"class_ids": ids[0], "scores": best[0] * 0.2, "boxes": boxes[0]}
    cases["allbelow"] = filt(fake_self, lo)
    save = {}
    for k, d in cases.items():
        if d is None:
            save[k + "_none"] = np.array([1])
        else:
            save[k + "_class_ids"] = d["class_ids"].numpy()
            save[k + "_scores"] = d["scores"].numpy()
            save[k + "_boxes"] = d["boxes"].numpy()
    np.savez_compressed(os.path.join(HERE, "filter.npz"), **save)

    # ---- 6. GT encoding (row P) + loss fwd/bwd (rows L, M) --------------------------------
    anchors = cfg.anchors
    rs = np.random.RandomState(5)
    gts, raw = [], {}
    for b in range(2):
        n = 3 + b * 2
        x1 = rs.uniform(0, 1100, n); y1 = rs.uniform(0, 300, n)
        bw = rs.uniform(20, 300, n); bh = rs.uniform(20, 150, n)
        bx = np.stack([x1, y1, np.minimum(x1 + bw, 1247), np.minimum(y1 + bh, 383)], 1).astype(np.float32)
        cls = rs.randint(0, 3, n)
        deltas_ref, idx_ref = ref_boxes.compute_deltas(bx, anchors)
        gt = np.zeros((16848, 12), np.float32)
        gt[idx_ref, 0] = 1.; gt[idx_ref, 1:5] = bx; gt[idx_ref, 5:9] = deltas_ref; gt[idx_ref, 9 + cls] = 1.
        gts.append(gt)
        raw[f"gtboxes{b}"] = bx; raw[f"gtcls{b}"] = cls; raw[f"gtidx{b}"] = idx_ref; raw[f"gtdeltas{b}"] = deltas_ref
    gt = torch.from_numpy(np.stack(gts))
    loss_mod = ref_model.Loss(cfg)
    p = pred.clone().requires_grad_(True)
    loss, stats = loss_mod(p, gt)
    loss.mean().backward()
    g = p.grad.numpy()
    pos = [np.nonzero(gts[b][:, 0])[0] for b in range(2)]
    neg = np.arange(5, 16848, 523)
    np.savez_compressed(os.path.join(HERE, "loss.npz"), pred_seed=11,
                        loss=loss.detach().numpy(), class_loss=stats["class_loss"].detach().numpy(),
                        score_loss=stats["score_loss"].detach().numpy(), bbox_loss=stats["bbox_loss"].detach().numpy(),
                        grad_pos0=g[0][pos[0]], grad_pos1=g[1][pos[1]], neg=neg, grad_neg=g[:, neg],
                        grad_abs_sum=np.array([np.abs(g.astype(np.float64)).sum()]), **raw)

    # ---- 7. one full train step, small input, dropout off (rows L-N) -----------------------
    small = (64, 96)
    for arch in ("squeezedet",):
        cfg_s = ref_cfg(arch, small)
        m = ref_model.SqueezeDetWithLoss(cfg_s).train()
        sd = synthetic.make_state_dict(arch, seed=1234)
        m.load_state_dict(sd, strict=True)
        x = synthetic.make_images(2, small, seed=3)
        gt_s = synthetic.make_gt(2, cfg_s.anchors, small, seed=2, min_boxes=2, max_boxes=3)
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        loss, stats = m({"image": x, "gt": gt_s})
        loss = loss.mean()
        opt.zero_grad()
        loss.backward()
        gnorms = {k: float(p_.grad.double().norm()) for k, p_ in m.named_parameters()}
        gsums = {k: float(p_.grad.double().sum()) for k, p_ in m.named_parameters()}
        total = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0))
        opt.step()
        names = [k for k, _ in m.named_parameters()]
        np.savez_compressed(os.path.join(HERE, "train_step_small.npz"),
                            names=np.array(names), loss=np.array([loss.item()]),
                            loss_vec=stats["loss"].detach().numpy(),
                            grad_norms=np.array([gnorms[k] for k in names]),
                            grad_sums=np.array([gsums[k] for k in names]), total_norm=np.array([total]),
                            new_param_sums=np.array([float(p_.double().sum()) for _, p_ in m.named_parameters()]),
                            new_param_abs=np.array([float(p_.double().abs().sum()) for _, p_ in m.named_parameters()]),
                            convdet_bias_grad=m.base.convdet.bias.grad.numpy(),
                            stem_w_grad=m.base.features[0].weight.grad.numpy())
    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
