// KITTI detection-AP bookkeeping on the host (the evaluation path).
//
// Restates the per-image greedy matching of the official KITTI devkit as the reference carries it:
// mmdet3d/core/evaluation/kitti_utils/eval.py:161-279 (compute_statistics_jit) and :291-338
// (fused_compute_statistics).  The reference JIT-compiles these loops with numba (absent here) and
// calls them once per image and per score threshold from Python; this entry walks all images and
// all thresholds of one (class, difficulty, min_overlap) cell in a single call.  The overlap
// matrices come from the caller (2D boxes: numpy; BEV / 3D: the rotated-overlap kernel of
// iou3d_nms.hip) — nothing here touches the GPU.
#include <cmath>
#include <vector>

#include "dm_common.h"

namespace {

struct KittiImage {
  const double *ov;      // (ndt, ngt) row-major: ov[j * ngt + i] = overlap(detection j, ground truth i)
  int ndt, ngt, ndc;
  const double *gt;      // (ngt, 5)  x1 y1 x2 y2 alpha
  const double *dt;      // (ndt, 6)  x1 y1 x2 y2 alpha score
  const double *dc;      // (ndc, 4)  "DontCare" regions
  const int64_t *ig;     // (ngt)  0 = counts, 1 = neighbouring class / too hard, -1 = other class
  const int64_t *id;     // (ndt)  0 = counts, 1 = too small, -1 = other class
};

struct KittiCounts {
  long tp = 0, fp = 0, fn = 0;
  double similarity = 0.0;
};

// One image, one score threshold.  collect != nullptr: the first pass (compute_fp = false), which only
// gathers the scores of the matched detections.
KittiCounts match_image(const KittiImage &v, int metric, double min_overlap, double thresh,
                        bool compute_fp, bool compute_aos, std::vector<double> *collect) {
  const long NO_DETECTION = -10000000;
  std::vector<char> assigned(v.ndt, 0), below(v.ndt, 0);
  if (compute_fp)
    for (int j = 0; j < v.ndt; ++j) below[j] = v.dt[j * 6 + 5] < thresh;
  KittiCounts c;
  std::vector<double> delta;
  for (int i = 0; i < v.ngt; ++i) {
    if (v.ig[i] == -1) continue;
    int det = -1;
    double valid = (double)NO_DETECTION, best = 0.0;
    bool took_ignored = false;
    for (int j = 0; j < v.ndt; ++j) {
      if (v.id[j] == -1 || assigned[j] || below[j]) continue;
      const double o = v.ov[(size_t)j * v.ngt + i];
      const double s = v.dt[j * 6 + 5];
      if (!compute_fp && o > min_overlap && s > valid) {          // highest score wins
        det = j;
        valid = s;
      } else if (compute_fp && o > min_overlap && (o > best || took_ignored) && v.id[j] == 0) {
        best = o;                                                // highest overlap wins
        det = j;
        valid = 1;
        took_ignored = false;
      } else if (compute_fp && o > min_overlap && valid == (double)NO_DETECTION && v.id[j] == 1) {
        det = j;                                                 // a too-small detection, kept only if nothing else
        valid = 1;
        took_ignored = true;
      }
    }
    const bool none = valid == (double)NO_DETECTION;
    if (none && v.ig[i] == 0) {
      ++c.fn;
    } else if (!none && (v.ig[i] == 1 || v.id[det] == 1)) {
      assigned[det] = 1;
    } else if (!none) {
      ++c.tp;
      if (collect) collect->push_back(v.dt[det * 6 + 5]);
      if (compute_aos) delta.push_back(v.gt[i * 5 + 4] - v.dt[det * 6 + 4]);
      assigned[det] = 1;
    }
  }
  if (compute_fp) {
    for (int j = 0; j < v.ndt; ++j)
      if (!(assigned[j] || v.id[j] == -1 || v.id[j] == 1 || below[j])) ++c.fp;
    long nstuff = 0;
    if (metric == 0) {
      // detections lying in a DontCare region (intersection / detection area) are not false positives
      for (int d = 0; d < v.ndc; ++d) {
        const double *q = v.dc + d * 4;
        for (int j = 0; j < v.ndt; ++j) {
          if (assigned[j] || v.id[j] == -1 || v.id[j] == 1 || below[j]) continue;
          const double *b = v.dt + j * 6;
          double o = 0.0;
          const double iw = std::fmin(b[2], q[2]) - std::fmax(b[0], q[0]);
          if (iw > 0) {
            const double ih = std::fmin(b[3], q[3]) - std::fmax(b[1], q[1]);
            if (ih > 0) o = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
          }
          if (o > min_overlap) {
            assigned[j] = 1;
            ++nstuff;
          }
        }
      }
    }
    c.fp -= nstuff;
    if (compute_aos) {
      double sum = 0.0;
      for (double d : delta) sum += (1.0 + std::cos(d)) / 2.0;
      c.similarity = (c.tp > 0 || c.fp > 0) ? sum : -1.0;
    }
  }
  return c;
}

void slice_image(KittiImage &v, const double *&ov, const double *&gt, const double *&dt,
                 const double *&dc, const int64_t *&ig, const int64_t *&id, int ngt, int ndt, int ndc) {
  v.ov = ov, v.gt = gt, v.dt = dt, v.dc = dc, v.ig = ig, v.id = id;
  v.ngt = ngt, v.ndt = ndt, v.ndc = ndc;
  ov += (size_t)ngt * ndt;
  gt += (size_t)ngt * 5;
  dt += (size_t)ndt * 6;
  dc += (size_t)ndc * 4;
  ig += ngt;
  id += ndt;
}

}  // namespace

// First pass (eval.py:498-512): scores of the true positives over all images at threshold 0.
// tp_scores: capacity = total number of ground truths.  Returns the number written, < 0 on bad input.
extern "C" long long dm_kitti_tp_scores_host(const double *overlaps, const int64_t *gt_nums,
                                             const int64_t *dt_nums, const int64_t *dc_nums,
                                             int n_images, const double *gt_datas,
                                             const double *dt_datas, const double *dontcares,
                                             const int64_t *ignored_gts, const int64_t *ignored_dets,
                                             int metric, double min_overlap, double *tp_scores) {
  if (n_images < 0 || !gt_nums || !dt_nums || !dc_nums || !tp_scores) return -1;
  std::vector<double> out;
  KittiImage v;
  for (int n = 0; n < n_images; ++n) {
    slice_image(v, overlaps, gt_datas, dt_datas, dontcares, ignored_gts, ignored_dets, (int)gt_nums[n],
                (int)dt_nums[n], (int)dc_nums[n]);
    match_image(v, metric, min_overlap, 0.0, false, false, &out);
  }
  for (size_t i = 0; i < out.size(); ++i) tp_scores[i] = out[i];
  return (long long)out.size();
}

// Second pass (eval.py:291-338): pr[t] += (tp, fp, fn, similarity) of every image at thresholds[t].
extern "C" int dm_kitti_pr_host(const double *overlaps, const int64_t *gt_nums, const int64_t *dt_nums,
                                const int64_t *dc_nums, int n_images, const double *gt_datas,
                                const double *dt_datas, const double *dontcares,
                                const int64_t *ignored_gts, const int64_t *ignored_dets, int metric,
                                double min_overlap, const double *thresholds, int n_thresholds,
                                int compute_aos, double *pr) {
  if (n_images < 0 || n_thresholds < 0 || !gt_nums || !dt_nums || !dc_nums || !pr) return DM_ERR_INVALID_ARG;
  KittiImage v;
  for (int n = 0; n < n_images; ++n) {
    slice_image(v, overlaps, gt_datas, dt_datas, dontcares, ignored_gts, ignored_dets, (int)gt_nums[n],
                (int)dt_nums[n], (int)dc_nums[n]);
    for (int t = 0; t < n_thresholds; ++t) {
      const KittiCounts c = match_image(v, metric, min_overlap, thresholds[t], true, compute_aos != 0, nullptr);
      pr[t * 4 + 0] += (double)c.tp;
      pr[t * 4 + 1] += (double)c.fp;
      pr[t * 4 + 2] += (double)c.fn;
      if (c.similarity != -1.0) pr[t * 4 + 3] += c.similarity;
    }
  }
  return DM_OK;
}
