// KITTI 2D object-detection average precision -- host code, no GPU work.
//
// Restates the evaluation the reference shells out to (src/datasets/kitti.py:99-124 runs the binary built from
// src/utils/kitti-eval/cpp/evaluate_object.cpp): per class and difficulty (easy / moderate / hard), ground truth is
// split into counted / ignored / foreign boxes (cleanData :281-352), a first matching pass collects the scores of
// the true positives (computeStatistics :354-497 with compute_fp = false), 41 recall-spaced score thresholds are
// derived from them (getThresholds :245-279), a second pass counts TP / FP / FN at every threshold with greedy
// best-overlap matching and DontCare suppression, precision is made monotone from the right and AP is the mean of
// the 11 samples i = 0, 4, ..., 40 (eval_class :503-571, saveStats :169-194).
// Peculiarities of THIS reference version that are kept: detections are never height-filtered (its ignored_det is
// only 0 / -1), a class is evaluated only if some detection of it exists, ties keep the first candidate.
// Orientation similarity (AOS) does not influence AP and is not computed.
//
// Checked against the reference's own binary (oracle/_ref/evaluate_object, built by oracle/ref_build/Makefile) in
// tests/test_kitti_eval.py.
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#include "sqd_common.h"

namespace {

constexpr int kRecallPts = 41;
const int kMinHeight[3] = {40, 25, 25};
const int kMaxOcclusion[3] = {0, 1, 2};
const double kMaxTruncation[3] = {0.15, 0.3, 0.5};
const double kMinOverlap[3] = {0.7, 0.5, 0.5};           // car, pedestrian, cyclist

enum GtKind { kCar = 0, kPedestrian = 1, kCyclist = 2, kVan = 3, kPersonSitting = 4, kDontCare = 5, kOtherKind = 6 };

struct Box { double x1, y1, x2, y2; };

// mode -1: intersection over union; mode 0: intersection over area of a
double overlap(const Box& a, const Box& b, int mode) {
  const double w = std::min(a.x2, b.x2) - std::max(a.x1, b.x1);
  const double h = std::min(a.y2, b.y2) - std::max(a.y1, b.y1);
  if (w <= 0 || h <= 0) return 0;
  const double inter = w * h;
  const double aa = (a.x2 - a.x1) * (a.y2 - a.y1), ab = (b.x2 - b.x1) * (b.y2 - b.y1);
  return mode == 0 ? inter / aa : inter / (aa + ab - inter);
}

struct Frame {                      // one image, already specialised to (class, difficulty)
  std::vector<Box> gt, det, dontcare;
  std::vector<int> gt_state;        // 0 counted, 1 ignored (neighbour class / too hard), -1 foreign
  std::vector<char> det_of_class;
  std::vector<double> det_score;
};

struct Counts { int tp = 0, fp = 0, fn = 0; };

// One matching pass over a frame.  with_fp = false: best SCORE candidate per ground truth, scores of the true positives
// appended to tp_scores.  with_fp = true: detections below `thresh` are invisible, best OVERLAP candidate wins, false
// positives are counted and those lying (by > min overlap of their own area) in a DontCare region are forgiven.
Counts match(const Frame& f, int cls, bool with_fp, double thresh, std::vector<double>* tp_scores) {
  Counts c;
  const size_t nd = f.det.size();
  std::vector<char> taken(nd, 0), hidden(nd, 0);
  if (with_fp)
    for (size_t j = 0; j < nd; ++j) hidden[j] = f.det_score[j] < thresh;
  for (size_t i = 0; i < f.gt.size(); ++i) {
    if (f.gt_state[i] == -1) continue;
    int pick = -1;
    double best_score = -10000000, best_ov = 0;
    for (size_t j = 0; j < nd; ++j) {
      if (!f.det_of_class[j] || taken[j] || hidden[j]) continue;
      const double ov = overlap(f.det[j], f.gt[i], -1);
      if (!(ov > kMinOverlap[cls])) continue;
      if (!with_fp) {
        if (f.det_score[j] > best_score) { pick = (int)j; best_score = f.det_score[j]; }
      } else if (ov > best_ov) {
        pick = (int)j; best_ov = ov;
      }
    }
    if (pick < 0) {
      if (f.gt_state[i] == 0) ++c.fn;
    } else if (f.gt_state[i] == 1) {
      taken[pick] = 1;                                   // matched to an ignored box: neither TP nor FP
    } else {
      ++c.tp;
      if (tp_scores) tp_scores->push_back(f.det_score[pick]);
      taken[pick] = 1;
    }
  }
  if (with_fp) {
    for (size_t j = 0; j < nd; ++j)
      if (f.det_of_class[j] && !taken[j] && !hidden[j]) ++c.fp;
    int forgiven = 0;
    for (const Box& dc : f.dontcare)
      for (size_t j = 0; j < nd; ++j) {
        if (!f.det_of_class[j] || taken[j] || hidden[j]) continue;
        if (overlap(f.det[j], dc, 0) > kMinOverlap[cls]) { taken[j] = 1; ++forgiven; }
      }
    c.fp -= forgiven;
  }
  return c;
}

// scores at (approximately) linearly spaced recall: a score is skipped when the next one lands closer to the target
std::vector<double> recall_thresholds(std::vector<double> scores, double n_gt) {
  std::sort(scores.begin(), scores.end(), std::greater<double>());
  std::vector<double> t;
  double target = 0;
  const size_t n = scores.size();
  for (size_t i = 0; i < n; ++i) {
    const double left = (double)(i + 1) / n_gt;
    const double right = (i + 1 < n) ? (double)(i + 2) / n_gt : left;
    if ((right - target) < (target - left) && i + 1 < n) continue;
    t.push_back(scores[i]);
    target += 1.0 / (kRecallPts - 1.0);
  }
  return t;
}

}  // namespace

// n_images frames.  Ground truth: gt_off [n+1], gt_kind [G] (0 car, 1 pedestrian, 2 cyclist, 3 van, 4 person_sitting,
// 5 DontCare, 6 other), gt_box [G][4] (x1,y1,x2,y2), gt_trunc [G], gt_occ [G].  Detections: det_off [n+1], det_cls [D]
// (0..2, anything else = not evaluated), det_box [D][4], det_score [D].  Outputs: ap [3][3] (class x difficulty),
// precision [3][3][41] (may be NULL), evaluated [3] (1 if the class had at least one detection, else its rows are 0).
extern "C" int sqd_kitti_ap(int n_images, const int* gt_off, const int* gt_kind, const double* gt_box,
                            const double* gt_trunc, const int* gt_occ, const int* det_off, const int* det_cls,
                            const double* det_box, const double* det_score, double* ap, double* precision,
                            int* evaluated) {
  SQD_CHECK_ARG(n_images >= 0 && gt_off && det_off && ap && evaluated);
  SQD_CHECK_ARG(n_images == 0 || (gt_off[n_images] == 0 || (gt_kind && gt_box && gt_trunc && gt_occ)));
  SQD_CHECK_ARG(n_images == 0 || (det_off[n_images] == 0 || (det_cls && det_box && det_score)));
  for (int i = 0; i < n_images; ++i) SQD_CHECK_ARG(gt_off[i] <= gt_off[i + 1] && det_off[i] <= det_off[i + 1]);
  const int total_det = n_images ? det_off[n_images] : 0;
  for (int cls = 0; cls < 3; ++cls) {
    evaluated[cls] = 0;
    for (int j = 0; j < total_det; ++j) if (det_cls[j] == cls) { evaluated[cls] = 1; break; }
    for (int d = 0; d < 3; ++d) {
      ap[cls * 3 + d] = 0;
      if (precision) std::fill(precision + (cls * 3 + d) * kRecallPts, precision + (cls * 3 + d + 1) * kRecallPts, 0.0);
    }
    if (!evaluated[cls]) continue;
    for (int d = 0; d < 3; ++d) {
      std::vector<Frame> frames((size_t)n_images);
      std::vector<double> tp_scores;
      int n_gt = 0;
      for (int im = 0; im < n_images; ++im) {
        Frame& f = frames[im];
        for (int g = gt_off[im]; g < gt_off[im + 1]; ++g) {
          const Box b{gt_box[4 * g], gt_box[4 * g + 1], gt_box[4 * g + 2], gt_box[4 * g + 3]};
          const int kind = gt_kind[g];
          int valid = -1;
          if (kind == cls) valid = 1;
          else if ((cls == kPedestrian && kind == kPersonSitting) || (cls == kCar && kind == kVan)) valid = 0;
          const bool too_hard = gt_occ[g] > kMaxOcclusion[d] || gt_trunc[g] > kMaxTruncation[d] || (b.y2 - b.y1) < kMinHeight[d];
          int state = -1;
          if (valid == 1 && !too_hard) { state = 0; ++n_gt; }
          else if (valid == 0 || (too_hard && valid == 1)) state = 1;
          f.gt.push_back(b); f.gt_state.push_back(state);
          if (kind == kDontCare) f.dontcare.push_back(b);
        }
        for (int j = det_off[im]; j < det_off[im + 1]; ++j) {
          f.det.push_back(Box{det_box[4 * j], det_box[4 * j + 1], det_box[4 * j + 2], det_box[4 * j + 3]});
          f.det_of_class.push_back(det_cls[j] == cls);
          f.det_score.push_back(det_score[j]);
        }
        match(f, cls, false, 0, &tp_scores);
      }
      std::vector<double> thr = recall_thresholds(tp_scores, (double)n_gt);
      if (thr.size() > (size_t)kRecallPts) thr.resize(kRecallPts);
      std::vector<Counts> acc(thr.size());
      for (const Frame& f : frames)
        for (size_t t = 0; t < thr.size(); ++t) {
          const Counts c = match(f, cls, true, thr[t], nullptr);
          acc[t].tp += c.tp; acc[t].fp += c.fp; acc[t].fn += c.fn;
        }
      double prec[kRecallPts];
      std::fill(prec, prec + kRecallPts, 0.0);
      for (size_t t = 0; t < thr.size(); ++t) prec[t] = acc[t].tp / (double)(acc[t].tp + acc[t].fp);
      for (size_t t = 0; t < thr.size(); ++t) {           // monotone from the right: first maximum of prec[t..40]
        double best = prec[t];
        for (int k = (int)t + 1; k < kRecallPts; ++k) if (best < prec[k]) best = prec[k];
        prec[t] = best;
      }
      double sum = 0;
      for (int i = 0; i < kRecallPts; i += 4) sum += prec[i];
      ap[cls * 3 + d] = sum / 11.0;
      if (precision) std::copy(prec, prec + kRecallPts, precision + (cls * 3 + d) * kRecallPts);
    }
  }
  return SQD_OK;
}
