// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Sequential restatement of the Tracking-side
// bookkeeping that surrounds the gathers of tracking_oracle.cpp, written the way the reference runs it:
//   DynObjTracking (per-label acceptance + label association)      src/Tracking.cc:1366-1612
//   RenewFrameInfo, object part                                     src/Tracking.cc:2806-2995
//   UpdateMask (per-label majority vote + conditional mask warp)    src/Tracking.cc:2997-3068
//   GetStaticTrack / GetDynamicTrackNew (rebuilt from frame 0)      src/Tracking.cc:2201-2421
// std::sort(…, SortPairInt) on the (label,count) pairs of a std::map: libstdc++ uses insertion sort
// below 16 elements, which is stable, so ties keep the map's ascending-label order — restated as
// "largest count, smallest label" (parity unpinned for >16 distinct labels).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "vdo_oracle.h"

namespace {
int majority_label(const std::vector<int>& v) {
  std::map<int, int> dups;
  for (int k : v) ++dups[k];
  int best = 0, cnt = -1;
  for (auto& kv : dups) if (kv.second > cnt) { cnt = kv.second; best = kv.first; }
  return best;
}
}  // namespace

// DynObjTracking.  Per-point inputs of the current frame (n): sem label, obj label (in/out: -1 outlier),
// key x/y, depth, 3-D flow; last-frame sem labels of the SAME points (they are carried index-aligned);
// last-frame object table (sem position, motion label, tracked flag).  Outputs: obj_label_inout updated,
// accepted objects as CSR (obj_off[n_obj+1], obj_idx), their sem label / motion label, max_id in/out.
extern "C" int vdo_oracle_dyn_obj_tracking(int n, const int32_t* sem_label, int32_t* obj_label_inout, const float* kx, const float* ky,
                                           const float* depth, const float* flow3d, const int32_t* last_sem_label,
                                           int n_last_obj, const int32_t* last_sem_pos, const int32_t* last_mod_label, const uint8_t* last_obj_stat,
                                           int img_w, int img_h, int shrink_row, int shrink_col, float sf_mg_thres, float sf_ds_thres, float th_depth_obj,
                                           int f_id, int32_t* max_id_inout,
                                           int32_t* obj_off, int32_t* obj_idx, int32_t* obj_sem, int32_t* obj_mod) {
  std::vector<int> uni(sem_label, sem_label + n);
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  std::vector<std::vector<int>> posi(uni.size());
  for (int i = 0; i < n; ++i) {
    if (obj_label_inout[i] == -1) continue;
    for (size_t j = 0; j < uni.size(); ++j) if (sem_label[i] == uni[j]) { posi[j].push_back(i); break; }
  }
  std::vector<std::vector<int>> ObjId;
  std::vector<int> sem_posi;
  for (size_t i = 0; i < posi.size(); ++i) {
    float count = 0;
    const float count_thres = 0.5f;
    for (int id : posi[i]) {
      const float u = kx[id], v = ky[id];
      if (v < shrink_row || v > (img_h - shrink_row) || u < shrink_col || u > (img_w - shrink_col)) count = count + 1;
    }
    if (count / posi[i].size() > count_thres) { for (int id : posi[i]) obj_label_inout[id] = -1; continue; }   // NB 0/0 = NaN > 0.5 is false: empty groups pass
    ObjId.push_back(posi[i]);
    sem_posi.push_back(uni[i]);
  }
  std::vector<std::vector<int>> ObjIdNew;
  std::vector<int> SemPosNew;
  for (size_t i = 0; i < ObjId.size(); ++i) {
    float obj_center_depth = 0, sf_count = 0;
    for (int id : ObjId[i]) {
      obj_center_depth = obj_center_depth + depth[id];
      const float fx = flow3d[3 * id], fz = flow3d[3 * id + 2];
      const float sf_norm = std::sqrt(fx * fx + fz * fz);
      if (sf_norm < sf_mg_thres) sf_count = sf_count + 1;
    }
    if (sf_count / ObjId[i].size() > sf_ds_thres) { for (int id : ObjId[i]) obj_label_inout[id] = 0; continue; }
    else if (obj_center_depth / ObjId[i].size() > th_depth_obj || ObjId[i].size() < 150) { for (int id : ObjId[i]) obj_label_inout[id] = -1; continue; }
    ObjIdNew.push_back(ObjId[i]);
    SemPosNew.push_back(sem_posi[i]);
  }
  int max_id = *max_id_inout;
  if (f_id == 1) max_id = 1;
  int off = 0;
  obj_off[0] = 0;
  for (size_t i = 0; i < ObjIdNew.size(); ++i) {
    std::vector<int> lb_last;
    for (int id : ObjIdNew[i]) lb_last.push_back(last_sem_label[id]);
    const int new_lab = majority_label(lb_last);
    int lab;
    if (max_id == 1) { lab = max_id; max_id = max_id + 1; }
    else {
      bool exist = false;
      lab = 0;
      for (int k = 0; k < n_last_obj; ++k)
        if (last_sem_pos[k] == new_lab && last_obj_stat[k]) { lab = last_mod_label[k]; exist = true; break; }
      if (!exist) { lab = max_id; max_id = max_id + 1; }
    }
    for (int id : ObjIdNew[i]) { obj_label_inout[id] = lab; obj_idx[off++] = id; }
    obj_off[i + 1] = off;
    obj_sem[i] = SemPosNew[i];
    obj_mod[i] = lab;
  }
  *max_id_inout = max_id;
  return (int)ObjIdNew.size();
}

// RenewFrameInfo, object part.  Objects of the current frame: inlier sets (CSR over the current object
// points), tracked flag, sem position, motion label.  cur_* = current frame's object points (keys, labels);
// tmp_* = the semi-dense sampling of the NEW image made by the Frame constructor (K10 outputs).
extern "C" int vdo_oracle_renew_object(int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                                       const int32_t* sem_pos, const int32_t* mod_label,
                                       const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                                       int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                                       const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                                       const int32_t* mask, const float* depth, const float* flow, int w, int h, int max_num_obj, int cap,
                                       float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                                       float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out) {
  int m = 0;
  auto push = [&](float x, float y, float d, int sl, float fx, float fy, float cx, float cy, int inl, int ol) -> bool {
    if (m >= cap) return false;
    key_x[m] = x; key_y[m] = y; depth_out[m] = d; sem_out[m] = sl; flow_x[m] = fx; flow_y[m] = fy; corr_x[m] = cx; corr_y[m] = cy;
    dyn_inlier_id[m] = inl; obj_label_out[m] = ol; ++m;
    return true;
  };
  std::vector<int> fea_count(n_obj);
  for (int i = 0; i < n_obj; ++i) {
    if (!obj_stat[i]) { fea_count[i] = -1; continue; }
    int count = 0;
    for (int q = inl_off[i]; q < inl_off[i + 1]; ++q) {
      const int id = inl_idx[q];
      const int x = (int)cur_x[id], y = (int)cur_y[id];
      if (x >= w || y >= h || x <= 0 || y <= 0) continue;
      const size_t o = (size_t)y * w + x;
      if (mask[o] != 0 && depth[o] < 25 && depth[o] > 0) {
        const float fx = flow[2 * o], fy = flow[2 * o + 1];
        if (x + fx < w && y + fy < h && x + fx > 0 && y + fy > 0) {
          if (!push((float)x, (float)y, depth[o], mask[o], fx, fy, x + fx, y + fy, id, cur_obj_label[id])) return -1;
          count = count + 1;
        }
      }
    }
    fea_count[i] = count;
  }
  const int n_check = m;
  for (int i = 0; i < n_obj; ++i) {
    if (!obj_stat[i]) continue;
    const int SemLabel = sem_pos[i];
    int tot_num = fea_count[i];
    int start_id = 0;
    const int step = 15;
    while (tot_num < max_num_obj) {
      if (start_id == step) break;
      for (int j = start_id; j < n_tmp; j = j + step) {
        if (tmp_label[j] != SemLabel) continue;
        float min_dist = 100;
        bool used = false;
        for (int k = 0; k < n_check; ++k) {
          const float cur = std::sqrt((key_x[k] - tmp_x[j]) * (key_x[k] - tmp_x[j]) + (key_y[k] - tmp_y[j]) * (key_y[k] - tmp_y[j]));
          if (cur < min_dist) min_dist = cur;
          if (min_dist < 1.0) { used = true; break; }
        }
        if (used) continue;
        if (!push(tmp_x[j], tmp_y[j], tmp_depth[j], tmp_label[j], tmp_flow_x[j], tmp_flow_y[j], tmp_corr_x[j], tmp_corr_y[j], -1, mod_label[i])) return -1;
        tot_num = tot_num + 1;
        if (tot_num >= max_num_obj) break;
      }
      start_id = start_id + 1;
    }
  }
  std::vector<int> uni(tmp_label, tmp_label + n_tmp);
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  std::vector<bool> known(uni.size(), false);
  for (int i = 0; i < n_obj; ++i)
    for (size_t j = 0; j < uni.size(); ++j)
      if (uni[j] == sem_pos[i] && obj_stat[i]) { known[j] = true; break; }
  for (size_t i = 0; i < uni.size(); ++i) {
    if (known[i]) continue;
    for (int j = 0; j < n_tmp; ++j)
      if (uni[i] == tmp_label[j])
        if (!push(tmp_x[j], tmp_y[j], tmp_depth[j], tmp_label[j], tmp_flow_x[j], tmp_flow_y[j], tmp_corr_x[j], tmp_corr_y[j], -1, -2)) return -1;
  }
  return m;
}

// UpdateMask: mask_cur is updated in place; returns the number of labels whose mask was recovered.
extern "C" int vdo_oracle_update_mask(int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                                      const int32_t* mask_last, const float* flow_last, int w, int h, int32_t* mask_cur) {
  std::vector<int> uni(last_sem_label, last_sem_label + n);
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  int recovered = 0;
  for (size_t i = 0; i < uni.size(); ++i) {
    std::vector<int> lab;
    for (int k = 0; k < n; ++k) {
      if (last_sem_label[k] != uni[i]) continue;
      const int u = (int)last_corr_x[k], v = (int)last_corr_y[k];
      if (u < w && u > 0 && v < h && v > 0) lab.push_back(mask_cur[(size_t)v * w + u]);
    }
    if (lab.size() < 100) continue;
    if (majority_label(lab) == 0) {
      vdo_oracle_mask_warp(mask_last, flow_last, w, h, uni[i], mask_cur);
      ++recovered;
    }
  }
  return recovered;
}

// GetStaticTrack / GetDynamicTrackNew: rebuilt from frame 0 every call, exactly like the reference.
// asso: CSR over frames (asso_off[n_frames+1]) of the index of the matched feature in the previous frame (-1: none).
// Outputs: tracklets as CSR of (frame, feature) pairs; obj_id per track when feat_label != NULL.
extern "C" int vdo_oracle_build_tracks(int n_frames, const int32_t* asso_off, const int32_t* asso, const int32_t* feat_label,
                                       int cap_tracks, int cap_pairs, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id) {
  std::vector<std::vector<std::pair<int, int>>> tracks;
  std::vector<int> object_id, pre;
  int id_so_far = 0;
  for (int i = 0; i < n_frames; ++i) {
    const int nf = asso_off[i + 1] - asso_off[i];
    const int32_t* tm = asso + asso_off[i];
    std::vector<int> cur(nf, -1);
    for (int j = 0; j < nf; ++j) {
      if (tm[j] == -1) continue;
      if (i > 0 && pre[tm[j]] != -1) {
        tracks[pre[tm[j]]].push_back(std::make_pair(i + 1, j));
        cur[j] = pre[tm[j]];
      } else {
        std::vector<std::pair<int, int>> t(2);
        t[0] = std::make_pair(i, tm[j]);
        t[1] = std::make_pair(i + 1, j);
        tracks.push_back(t);
        if (feat_label) object_id.push_back(feat_label[asso_off[i] + j]);
        cur[j] = id_so_far;
        id_so_far = id_so_far + 1;
      }
    }
    pre = cur;
  }
  if ((int)tracks.size() > cap_tracks) return -1;
  int off = 0;
  track_off[0] = 0;
  for (size_t t = 0; t < tracks.size(); ++t) {
    for (auto& pr : tracks[t]) {
      if (off >= cap_pairs) return -1;
      pair_frame[off] = pr.first; pair_feat[off] = pr.second; ++off;
    }
    track_off[t + 1] = off;
    if (feat_label && obj_id) obj_id[t] = object_id[t];
  }
  return (int)tracks.size();
}
