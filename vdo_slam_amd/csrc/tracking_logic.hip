// Tracking-side bookkeeping around the K11-K15 gathers (tracking.hip), as the C-ABI serves it:
//   vdo_dyn_obj_tracking   Tracking::DynObjTracking            reference src/Tracking.cc:1366-1612
//   vdo_renew_object       Tracking::RenewFrameInfo, objects    :2806-2995
//   vdo_update_mask        Tracking::UpdateMask                 :2997-3068
//   vdo_tracks_*           GetStaticTrack / GetDynamicTrackNew  :2201-2421 (incremental instead of from frame 0)
// Split of work: everything that touches an image or is O(n*m) runs on the GPU (gathers at the
// carried / flowed positions, the "is there a carried point within 1 px" test, the per-label vote
// and the conditional mask warp — the latter two without any host round trip); the order-dependent
// selections (first-come truncation, stride-15 interleave, id hand-out) stay on the host and consume
// flags.  Grouping by label uses one pass over label-slot tables instead of the reference's nested
// searches; float accumulations keep the reference's per-label index order.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"
#include "frame_images.hpp"
#include "near_flags.hpp"
#include "arena.hpp"
#include "tracking_shared.hpp"

namespace vdo {

// object-carry predicate at INT-truncated coordinates (:2832-2856): ok, sem label, depth, flow
__global__ void k_renew_obj_pred(int n, const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ mask,
                                 const float* __restrict__ depth, const float* __restrict__ flow, int w, int h,
                                 int32_t* __restrict__ ok, int32_t* __restrict__ sem, float* __restrict__ dout, float* __restrict__ fx, float* __restrict__ fy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)px[i], y = (int)py[i];
  int good = 0, sl = 0;
  float d = 0, fxe = 0, fye = 0;
  if (!(x >= w || y >= h || x <= 0 || y <= 0)) {
    const size_t o = (size_t)y * w + x;
    sl = mask[o]; d = depth[o];
    if (sl != 0 && d < 25 && d > 0) {
      fxe = flow[2 * o]; fye = flow[2 * o + 1];
      if (x + fxe < w && y + fye < h && x + fxe > 0 && y + fye > 0) good = 1;
    }
  }
  ok[i] = good; sem[i] = sl; dout[i] = d; fx[i] = fxe; fy[i] = fye;
}

// RenewFrameInfo (objects) + mvObj3DPoint in ONE launch: thread i judges carried candidate i (k_renew_obj_pred) and back-projects it from its
// truncated position with the depth it has just read; thread j back-projects sample j of the new image and clears its "within 1 px of a carried
// point" flag for the k_near_flags_sel launch that follows (one launch instead of predicate + fill + two back-projections).
__global__ void k_renew_obj_all(int n, const float* __restrict__ px, const float* __restrict__ py, const int32_t* __restrict__ mask,
                                const float* __restrict__ depth, const float* __restrict__ flow, int w, int h,
                                int32_t* __restrict__ ok, int32_t* __restrict__ sem, float* __restrict__ dout, float* __restrict__ fx, float* __restrict__ fy,
                                Cam cam, float* __restrict__ xyz_c,
                                int n_tmp, const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qd, float* __restrict__ xyz_t,
                                int32_t* __restrict__ used) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int x = (int)px[i], y = (int)py[i];
    int good = 0, sl = 0;
    float d = 0, fxe = 0, fye = 0;
    if (!(x >= w || y >= h || x <= 0 || y <= 0)) {
      const size_t o = (size_t)y * w + x;
      sl = mask[o]; d = depth[o];
      if (sl != 0 && d < 25 && d > 0) {
        fxe = flow[2 * o]; fye = flow[2 * o + 1];
        if (x + fxe < w && y + fye < h && x + fxe > 0 && y + fye > 0) good = 1;
      }
    }
    ok[i] = good; sem[i] = sl; dout[i] = d; fx[i] = fxe; fy[i] = fye;
    float o3[3];
    backproject(cam, (float)x, (float)y, d, o3);
    xyz_c[3 * i] = o3[0]; xyz_c[3 * i + 1] = o3[1]; xyz_c[3 * i + 2] = o3[2];
  }
  if (i < n_tmp) {
    used[i] = 0;
    float o3[3];
    backproject(cam, qx[i], qy[i], qd[i], o3);
    xyz_t[3 * i] = o3[0]; xyz_t[3 * i + 1] = o3[1]; xyz_t[3 * i + 2] = o3[2];
  }
}

constexpr int kVoteBins = 1024;    // instance labels of a mask are small non-negative integers

// One workgroup per call: vote of the current mask at the flowed positions of one last-frame label.
// flag[0] = 1 when >= 100 positions are inside the image and the most frequent label (smallest label on
// ties) is the background 0; flag[1] = 1 if a label fell outside the histogram (host reports an error).
__global__ __launch_bounds__(256) void k_label_vote(int n, const float* __restrict__ cx, const float* __restrict__ cy,
                                                    const int32_t* __restrict__ mask, int w, int h, int32_t* __restrict__ flag) {
  __shared__ int hist[kVoteBins];
  __shared__ int s_valid, s_bad;
  for (int i = threadIdx.x; i < kVoteBins; i += 256) hist[i] = 0;
  if (threadIdx.x == 0) { s_valid = 0; s_bad = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int u = (int)cx[i], v = (int)cy[i];
    if (u < w && u > 0 && v < h && v > 0) {
      const int l = mask[(size_t)v * w + u];
      if (l < 0 || l >= kVoteBins) atomicOr(&s_bad, 1);
      else { atomicAdd(&hist[l], 1); atomicAdd(&s_valid, 1); }
    }
  }
  __syncthreads();
  // most frequent label, smallest on ties: every thread scans 4 bins, then a (count desc, label asc) reduction
  __shared__ int s_cnt[256], s_lab[256];
  {
    int best = threadIdx.x * 4, cnt = hist[best];
    for (int l = best + 1; l < threadIdx.x * 4 + 4; ++l) if (hist[l] > cnt) { cnt = hist[l]; best = l; }
    s_cnt[threadIdx.x] = cnt; s_lab[threadIdx.x] = best;
  }
  __syncthreads();
  for (int step = 128; step > 0; step >>= 1) {
    if (threadIdx.x < step) {
      const int c2 = s_cnt[threadIdx.x + step], l2 = s_lab[threadIdx.x + step];
      if (c2 > s_cnt[threadIdx.x] || (c2 == s_cnt[threadIdx.x] && l2 < s_lab[threadIdx.x])) { s_cnt[threadIdx.x] = c2; s_lab[threadIdx.x] = l2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    flag[0] = (s_valid >= 100 && s_lab[0] == 0 && !s_bad) ? 1 : 0;
    flag[1] = s_bad;
  }
}

// mask warp of one label, executed only if the vote said so (flag on the device: no host round trip)
__global__ void k_mask_warp_if(const int32_t* __restrict__ flag, const int32_t* __restrict__ mask_last, const float* __restrict__ flow_last,
                               int w, int h, int32_t lab, int32_t* __restrict__ mask_cur) {
  if (!flag[0]) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (k >= w) return;
  const size_t o = (size_t)j * w + k;
  if (mask_last[o] != lab) return;
  const int fx = (int)flow_last[2 * o], fy = (int)flow_last[2 * o + 1];
  if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) mask_cur[(size_t)(j + fy) * w + (k + fx)] = lab;
}

// ---- UpdateMask in three launches, whatever the number of labels (the label-after-label form above costs two launches per
// label).  The reference walks the labels in ascending order; a label is "recovered" when >= 100 of its flowed samples are
// inside the image and the most frequent CURRENT label under them is the background, and then its whole last-frame mask is
// warped into the current mask - visible to the votes of the labels after it.  Equivalent without the sequential image
// passes:
//   1. k_warp_candidates: ONE pass over the last frame: cand[q] gets bit s for every label slot s whose warp lands on q;
//   2. k_votes_par (one workgroup per label, the last one to finish re-votes sequentially what an earlier recovery can change): the label seen under a sample q is the recovered label of the highest
//      earlier slot with its bit in cand[q] (a later warp overwrites an earlier one), else the untouched current mask;
//   3. k_apply_warps: every pixel with candidate bits takes the highest recovered slot's label; cand is cleared on the way.
// Up to 64 labels per frame (one bit each); more fall back to the label-after-label launches.
struct LabelSlots { int n; int32_t lab[64]; };

__global__ __launch_bounds__(256) void k_warp_candidates(const int32_t* __restrict__ mask_last, const float* __restrict__ flow_last, int w, int h, LabelSlots T,
                                                         unsigned long long* __restrict__ cand) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (k >= w) return;
  const size_t o = (size_t)j * w + k;
  const int32_t lab = mask_last[o];
  int slot = -1;
  for (int s = 0; s < T.n; ++s) if (T.lab[s] == lab) slot = s;
  if (slot < 0) return;
  const int fx = (int)flow_last[2 * o], fy = (int)flow_last[2 * o + 1];
  if (k + fx < w && k + fx > 0 && j + fy < h && j + fy > 0) atomicOr(&cand[(size_t)(j + fy) * w + (k + fx)], 1ull << slot);
}

// The vote of ONE label slot s by one workgroup: the labels seen under its flowed samples - the recovered label of the highest slot in
// `rec` whose warp lands on the pixel, else the untouched current mask -, their most frequent value (smallest on ties), and the verdict
// "recovered" (>= 100 samples inside the image and the background wins).  Writes flag[2s] / flag[2s+1]; every thread returns the verdict.
struct VoteLds { int hist[kVoteBins]; int valid, bad; int cnt[256], lab[256]; int verdict; };
__device__ __forceinline__ int vote_label(VoteLds& L, const LabelSlots& T, int s, unsigned long long rec, const int32_t* __restrict__ off, const float* __restrict__ cx,
                                          const float* __restrict__ cy, const int32_t* __restrict__ mask, const unsigned long long* __restrict__ cand, int w, int h,
                                          int32_t* __restrict__ flag) {
  for (int i = threadIdx.x; i < kVoteBins; i += 256) L.hist[i] = 0;
  if (threadIdx.x == 0) { L.valid = 0; L.bad = 0; }
  __syncthreads();
  // (almost every sample of a label sees the same value: the lanes of a wave that agree send ONE LDS atomic with their count instead of
  //  64 same-address ones; whole waves run the loop so that the ballots are complete)
  const int lo = off[s], hi = off[s + 1];
  // Round 5: FOUR rounds of samples per trip, their loads hoisted - the positions of all four first, then the four mask (and candidate) words - so that a label of
  // ~600 samples pays two dependent trips to memory instead of six (this kernel heads the frame's object chain on an otherwise idle device: ~4 us per trip).
  for (int base = lo + (int)(threadIdx.x & ~63u); base < hi; base += 4 * 256) {
    float px[4], py[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = min(base + k * 256 + (int)(threadIdx.x & 63u), hi - 1); px[k] = cx[i]; py[k] = cy[i]; }
    int ml[4]; unsigned long long mc[4]; bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * 256 + (int)(threadIdx.x & 63u);
      const int u = (int)px[k], v = (int)py[k];
      in[k] = i < hi && u < w && u > 0 && v < h && v > 0;
      const size_t q = in[k] ? (size_t)v * w + u : 0;
      ml[k] = mask[q];
      mc[k] = rec ? cand[q] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + k * 256 >= hi) break;                      // (uniform over the wave: the ballots below are complete)
      const bool inside = in[k];
      const unsigned long long m = mc[k] & rec;
      const int l = inside ? (m ? T.lab[63 - __clzll((long long)m)] : ml[k]) : -1;
      const bool bad = inside && (l < 0 || l >= kVoteBins);
      if (__ballot(bad) && (threadIdx.x & 63u) == 0) atomicOr(&L.bad, 1);
      bool todo = inside && !bad;
      const unsigned long long ok = __ballot(todo);
      if (ok && (threadIdx.x & 63u) == 0) atomicAdd(&L.valid, (int)__popcll(ok));
      for (unsigned long long act = ok; act;) {
        const int leader = (int)__ffsll((long long)act) - 1;
        const int l0 = __shfl(l, leader);
        const unsigned long long same = __ballot(todo && l == l0);
        if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&L.hist[l0], (int)__popcll(same));
        if (l == l0) todo = false;
        act &= ~same;
      }
    }
  }
  __syncthreads();
  {
    int best = threadIdx.x * 4, cnt = L.hist[best];
    for (int l = best + 1; l < threadIdx.x * 4 + 4; ++l) if (L.hist[l] > cnt) { cnt = L.hist[l]; best = l; }
    L.cnt[threadIdx.x] = cnt; L.lab[threadIdx.x] = best;
  }
  __syncthreads();
  for (int step = 128; step > 0; step >>= 1) {
    if (threadIdx.x < step) {
      const int c2 = L.cnt[threadIdx.x + step], l2 = L.lab[threadIdx.x + step];
      if (c2 > L.cnt[threadIdx.x] || (c2 == L.cnt[threadIdx.x] && l2 < L.lab[threadIdx.x])) { L.cnt[threadIdx.x] = c2; L.lab[threadIdx.x] = l2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int r = (L.valid >= 100 && L.lab[0] == 0 && !L.bad) ? 1 : 0;
    flag[2 * s] = r; flag[2 * s + 1] = L.bad;
    L.verdict = r;
  }
  __syncthreads();
  return L.verdict;
}

// off[s] .. off[s+1]: samples (flowed positions) of label slot s.  flag[2s] = recovered, flag[2s+1] = a label fell outside the
// histogram; rec_out = bit mask of the recovered slots.
// One workgroup PER LABEL votes under the hypothesis that no earlier label was recovered (what it sees is then the untouched current mask:
// true in almost every frame - a mask goes missing now and then).  The workgroup that finishes last (a ticket) looks at the verdicts: the
// labels up to and including the FIRST recovered one voted under a true hypothesis; only the ones after it are voted again, label after
// label, with the recovered set growing - the single-workgroup sequential walk this kernel used to be for every label (23-28 us per frame
// on the object chain; ~5 us now).  ticket: one int, zero between launches (the last workgroup resets it).
__global__ __launch_bounds__(256) void k_votes_par(LabelSlots T, const int32_t* __restrict__ off, const float* __restrict__ cx, const float* __restrict__ cy,
                                                   const int32_t* __restrict__ mask, const unsigned long long* __restrict__ cand, int w, int h,
                                                   int32_t* __restrict__ flag, unsigned long long* __restrict__ rec_out, int* __restrict__ ticket) {
  __shared__ VoteLds L;
  __shared__ int s_last, s_first;
  vote_label(L, T, (int)blockIdx.x, 0ull, off, cx, cy, mask, cand, w, h, flag);
  if (threadIdx.x == 0) {
    __threadfence();                                           // the verdict is visible device-wide before the ticket is drawn
    s_last = (atomicAdd(ticket, 1) == T.n - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) { __threadfence(); s_first = 64; *ticket = 0; }
  __syncthreads();
  if ((int)threadIdx.x < T.n && __hip_atomic_load(&flag[2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&s_first, (int)threadIdx.x);   // one round of loads
  __syncthreads();
  if (threadIdx.x == 0 && s_first == 64) s_first = -1;
  __syncthreads();
  const int first = s_first;
  unsigned long long rec = first >= 0 ? (1ull << first) : 0ull;
  for (int s = first + 1; first >= 0 && s < T.n; ++s)
    if (vote_label(L, T, s, rec, off, cx, cy, mask, cand, w, h, flag)) rec |= 1ull << s;
  if (threadIdx.x == 0) *rec_out = rec;
}

__global__ void k_apply_warps(unsigned long long* __restrict__ cand, const unsigned long long* __restrict__ rec, LabelSlots T, int64_t n, int32_t* __restrict__ mask_cur) {
  const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned long long c = cand[q];
  if (!c) return;
  const unsigned long long m = c & *rec;
  if (m) mask_cur[q] = T.lab[63 - __clzll((long long)m)];
  cand[q] = 0ull;
}

// the UpdateMask launches for the label slots uni[0..L) with samples dx/dy grouped by slot (off on the host); doff / dflag /
// drec: device scratch of the caller's arena (L+1 ints, 2L ints, one 64-bit word)
static void launch_update_mask(vdo_frame_images* cur, vdo_frame_images* last, const std::vector<int32_t>& uni, const std::vector<int>& off, const float* dx, const float* dy,
                               const int32_t* doff, int32_t* dflag, unsigned long long* drec, hipStream_t st) {
  const int L = (int)uni.size();
  if (L >= 1 && L <= 64 && cur->d_cand && cur->d_ticket) {
    LabelSlots T{};
    T.n = L;
    for (int s = 0; s < L; ++s) T.lab[s] = uni[s];
    hipLaunchKernelGGL(k_warp_candidates, dim3((cur->w + 255) / 256, cur->h), dim3(256), 0, st, (const int32_t*)last->d_mask, (const float*)last->d_flow, cur->w, cur->h, T, cur->d_cand);
    hipLaunchKernelGGL(k_votes_par, dim3(L), dim3(256), 0, st, T, doff, dx, dy, (const int32_t*)cur->d_mask, (const unsigned long long*)cur->d_cand, cur->w, cur->h, dflag, drec, cur->d_ticket);
    const int64_t np = (int64_t)cur->w * cur->h;
    hipLaunchKernelGGL(k_apply_warps, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, cur->d_cand, (const unsigned long long*)drec, T, np, cur->d_mask);
    return;
  }
  for (int s = 0; s < L; ++s) {      // label after label (a recovered mask is visible to the next label's vote, as in the reference)
    const int ns = off[s + 1] - off[s];
    hipLaunchKernelGGL(k_label_vote, dim3(1), dim3(256), 0, st, ns, dx + off[s], dy + off[s], (const int32_t*)cur->d_mask, cur->w, cur->h, dflag + 2 * s);
    hipLaunchKernelGGL(k_mask_warp_if, dim3((cur->w + 255) / 256, cur->h), dim3(256), 0, st, (const int32_t*)(dflag + 2 * s), (const int32_t*)last->d_mask,
                       (const float*)last->d_flow, cur->w, cur->h, uni[s], cur->d_mask);
  }
}

// sorted distinct labels + slot of every element.  Mask labels are small integers: a presence table over
// [min, max] gives both in O(n) (a sort + binary searches of ~5k labels cost ~0.1 ms per call); wide label
// ranges fall back to sort + lower_bound.
static void label_slots(int n, const int32_t* lab, std::vector<int32_t>& uni, std::vector<int32_t>& slot) {
  uni.clear();
  slot.resize(n);
  if (n == 0) return;
  int32_t lo = lab[0], hi = lab[0];
  for (int i = 1; i < n; ++i) { lo = std::min(lo, lab[i]); hi = std::max(hi, lab[i]); }
  const int64_t range = (int64_t)hi - lo + 1;
  if (range <= 65536) {
    static thread_local std::vector<int32_t> table;
    table.assign((size_t)range, -1);
    for (int i = 0; i < n; ++i) table[lab[i] - lo] = 0;
    for (int64_t v = 0; v < range; ++v) if (table[v] == 0) { table[v] = (int32_t)uni.size(); uni.push_back((int32_t)(lo + v)); }
    for (int i = 0; i < n; ++i) slot[i] = table[lab[i] - lo];
    return;
  }
  uni.assign(lab, lab + n);
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  for (int i = 0; i < n; ++i) slot[i] = (int32_t)(std::lower_bound(uni.begin(), uni.end(), lab[i]) - uni.begin());
}

// most frequent value, smallest on ties (std::sort of <16 map entries is a stable insertion sort)
static int majority(std::vector<int32_t>& v) {
  if (v.empty()) return 0;
  int32_t lo = v[0], hi = v[0];
  for (int32_t x : v) { lo = std::min(lo, x); hi = std::max(hi, x); }
  const int64_t range = (int64_t)hi - lo + 1;
  if (range <= 65536) {
    static thread_local std::vector<int32_t> cntv;
    cntv.assign((size_t)range, 0);
    for (int32_t x : v) ++cntv[x - lo];
    int best = lo, cnt = -1;
    for (int64_t k = 0; k < range; ++k) if (cntv[k] > cnt) { cnt = cntv[k]; best = (int)(lo + k); }
    return best;
  }
  std::sort(v.begin(), v.end());
  int best = 0, cnt = -1;
  for (size_t a = 0; a < v.size();) {
    size_t b = a;
    while (b < v.size() && v[b] == v[a]) ++b;
    if ((int)(b - a) > cnt) { cnt = (int)(b - a); best = v[a]; }
    a = b;
  }
  return best;
}

}  // namespace vdo

using namespace vdo;

extern "C" int vdo_dyn_obj_tracking(const vdo_dyn_obj_params* prm, int n, const int32_t* sem_label, int32_t* obj_label_inout,
                                    const float* key_x, const float* key_y, const float* depth, const float* flow3d, const int32_t* last_sem_label,
                                    int n_last_obj, const int32_t* last_sem_pos, const int32_t* last_mod_label, const uint8_t* last_obj_stat,
                                    int32_t* max_id_inout, int32_t* obj_off, int32_t* obj_idx, int32_t* obj_sem, int32_t* obj_mod, int* n_obj_out) {
  if (!prm || !n_obj_out || !max_id_inout || n < 0 || (n > 0 && (!sem_label || !obj_label_inout || !key_x || !key_y || !depth || !flow3d || !last_sem_label)))
    return set_error(VDO_ERR_INVALID, "vdo_dyn_obj_tracking: bad argument");
  std::vector<int32_t> uni, slot;
  label_slots(n, sem_label, uni, slot);
  const int L = (int)uni.size();
  // one pass: members per label (index order), border votes
  std::vector<int> cnt(L, 0);
  std::vector<float> border(L, 0.f);
  for (int i = 0; i < n; ++i) {
    if (obj_label_inout[i] == -1) continue;
    const int s = slot[i];
    ++cnt[s];
    const float u = key_x[i], v = key_y[i];
    if (v < prm->shrink_row || v > (prm->img_h - prm->shrink_row) || u < prm->shrink_col || u > (prm->img_w - prm->shrink_col)) border[s] += 1.f;
  }
  // state per label: 0 candidate, 1 border-dropped
  std::vector<int> state(L, 0);
  for (int s = 0; s < L; ++s) if (border[s] / (float)cnt[s] > 0.5f) state[s] = 1;     // 0/0 -> NaN -> kept, as in the reference
  // second pass: depth sum / slow-flow votes of the surviving labels (float, index order)
  std::vector<float> dsum(L, 0.f), slow(L, 0.f);
  for (int i = 0; i < n; ++i) {
    if (obj_label_inout[i] == -1) continue;
    const int s = slot[i];
    if (state[s] == 1) { obj_label_inout[i] = -1; continue; }
    dsum[s] = dsum[s] + depth[i];
    const float fx = flow3d[3 * i], fz = flow3d[3 * i + 2];
    if (std::sqrt(fx * fx + fz * fz) < prm->sf_mg_thres) slow[s] = slow[s] + 1.f;
  }
  for (int s = 0; s < L; ++s) {
    if (state[s]) continue;
    if (slow[s] / (float)cnt[s] > prm->sf_ds_thres) state[s] = 2;                                    // static object -> label 0
    else if (dsum[s] / (float)cnt[s] > prm->th_depth_obj || cnt[s] < 150) state[s] = 3;              // too far / too small -> -1
  }
  // members of the accepted labels, CSR in label order
  std::vector<int> acc_of(L, -1);
  int n_acc = 0;
  obj_off[0] = 0;
  for (int s = 0; s < L; ++s) if (state[s] == 0) { acc_of[s] = n_acc; obj_off[n_acc + 1] = obj_off[n_acc] + cnt[s]; obj_sem[n_acc] = uni[s]; ++n_acc; }
  std::vector<int> cur(n_acc + 1);
  for (int a = 0; a <= n_acc; ++a) cur[a] = obj_off[a];
  for (int i = 0; i < n; ++i) {
    if (obj_label_inout[i] == -1) continue;       // includes the border-dropped ones set above
    const int s = slot[i];
    if (state[s] == 2) obj_label_inout[i] = 0;
    else if (state[s] == 3) obj_label_inout[i] = -1;
    else if (state[s] == 0) obj_idx[cur[acc_of[s]]++] = i;
  }
  // association with the last frame's objects
  int max_id = *max_id_inout;
  if (prm->f_id == 1) max_id = 1;
  std::vector<int32_t> votes;
  for (int a = 0; a < n_acc; ++a) {
    votes.clear();
    for (int q = obj_off[a]; q < obj_off[a + 1]; ++q) votes.push_back(last_sem_label[obj_idx[q]]);
    const int new_lab = majority(votes);
    int lab = -1;
    if (max_id != 1)
      for (int k = 0; k < n_last_obj; ++k)
        if (last_sem_pos[k] == new_lab && last_obj_stat[k]) { lab = last_mod_label[k]; break; }
    if (lab == -1) { lab = max_id; max_id = max_id + 1; }
    // NB a found label can legitimately be any value the previous frame handed out (>= 1), never -1
    for (int q = obj_off[a]; q < obj_off[a + 1]; ++q) obj_label_inout[obj_idx[q]] = lab;
    obj_mod[a] = lab;
  }
  *max_id_inout = max_id;
  *n_obj_out = n_acc;
  return VDO_OK;
}

// RenewFrameInfo (objects) + the 3-D points of the new set (mvObj3DPoint) in ONE pass over the device: the carried candidates
// are judged, the "within 1 px of a carried key" test of the sampled points runs against the device-side verdicts, carried
// and sampled points are all back-projected - one copy back, one synchronisation.  xyz_out may be NULL (K4 / Twc unused then).
extern "C" int vdo_renew_object_world(vdo_frame_images* f, int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                                      const int32_t* sem_pos, const int32_t* mod_label,
                                      const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                                      int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                                      const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                                      int max_num_obj, int cap, const float K4[4], const float Twc[16],
                                      float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                                      float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out, float* xyz_out, int* n_out) {
  if (!f || !n_out || n_obj < 0 || n_tmp < 0 || cap < 0 || (xyz_out && (!K4 || !Twc))) return set_error(VDO_ERR_INVALID, "vdo_renew_object: bad argument");
  int rc = ctx_bind(f->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(f->ctx);
  if (!S.reserve(Arena::bytes_for(24 * ((size_t)(n_obj ? inl_off[n_obj] : 0) + (size_t)n_tmp + 64)))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  // ---- carried candidates: inliers of the tracked objects, object-major (the reference's visiting order)
  static thread_local std::vector<float> cx, cy, dd, fx, fy, xyz_c, xyz_t;      // (per-thread scratch: no allocation in steady state)
  static thread_local std::vector<int32_t> cid, cobj, ok, sem, used;
  cx.clear(); cy.clear(); cid.clear(); cobj.clear();
  for (int i = 0; i < n_obj; ++i) {
    if (!obj_stat[i]) continue;
    for (int q = inl_off[i]; q < inl_off[i + 1]; ++q) { const int id = inl_idx[q]; cx.push_back(cur_x[id]); cy.push_back(cur_y[id]); cid.push_back(id); cobj.push_back(i); }
  }
  const int nc = (int)cx.size();
  ok.resize(nc); sem.resize(nc); dd.resize(nc); fx.resize(nc); fy.resize(nc);
  used.assign(n_tmp, 0);
  xyz_c.resize(xyz_out ? 3 * (size_t)nc : 0); xyz_t.resize(xyz_out ? 3 * (size_t)n_tmp : 0);
  if (nc || (n_tmp && xyz_out)) {
    float *dx = S.up(cx.data(), nc), *dy = S.up(cy.data(), nc);
    float *dqx = S.up(tmp_x, n_tmp), *dqy = S.up(tmp_y, n_tmp), *dqd = S.up(xyz_out ? tmp_depth : nullptr, xyz_out ? n_tmp : 0);
    int32_t *dok = S.up<int32_t>(nullptr, nc), *dsem = S.up<int32_t>(nullptr, nc);
    float *ddd = S.up<float>(nullptr, nc), *dfx = S.up<float>(nullptr, nc), *dfy = S.up<float>(nullptr, nc);
    int32_t* dused = S.up<int32_t>(nullptr, n_tmp);
    float *dxc = S.up<float>(nullptr, xyz_out ? 3 * (size_t)nc : 0), *dxt = S.up<float>(nullptr, xyz_out ? 3 * (size_t)n_tmp : 0);
    if (!dxt || !dxc || !dused || !dfy) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
    hipStream_t st = S.stream();
    const Cam cam = xyz_out ? make_cam_Twc(K4, Twc) : Cam{};
    if (xyz_out) {                         // (what FramePipeline calls: one fused launch, then the 1-px test)
      const int nmax = std::max(nc, n_tmp);
      hipLaunchKernelGGL(k_renew_obj_all, dim3((nmax + 255) / 256), dim3(256), 0, st, nc, (const float*)dx, (const float*)dy, (const int32_t*)f->d_mask,
                         (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, dok, dsem, ddd, dfx, dfy, cam, dxc,
                         n_tmp, (const float*)dqx, (const float*)dqy, (const float*)dqd, dxt, dused);
      // top-up from the semi-dense sampling of the new image: "is a carried point within 1 px" against the verdicts above
      if (nc && n_tmp) launch_near_flags_sel(st, n_tmp, dqx, dqy, nc, dx, dy, dok, 1, dused, false);
    } else if (nc) {
      hipLaunchKernelGGL(k_renew_obj_pred, dim3((nc + 255) / 256), dim3(256), 0, st, nc, (const float*)dx, (const float*)dy, (const int32_t*)f->d_mask,
                         (const float*)f->d_depth, (const float*)f->d_flow, f->w, f->h, dok, dsem, ddd, dfx, dfy);
      if (n_tmp) launch_near_flags_sel(st, n_tmp, dqx, dqy, nc, dx, dy, dok, 1, dused);
    }
    S.down(ok.data(), dok, nc); S.down(sem.data(), dsem, nc); S.down(dd.data(), ddd, nc); S.down(fx.data(), dfx, nc); S.down(fy.data(), dfy, nc);
    if (nc && n_tmp) S.down(used.data(), dused, n_tmp);
    if (xyz_out) { S.down(xyz_c.data(), dxc, 3 * (size_t)nc); S.down(xyz_t.data(), dxt, 3 * (size_t)n_tmp); }
    rc = S.finish("vdo_renew_object");
    if (rc != VDO_OK) return rc;
  }
  int m = 0;
  auto push = [&](float x, float y, float d, int sl, float flx, float fly, float crx, float cry, int inl, int ol, const float* p3) -> bool {
    if (m >= cap) return false;
    key_x[m] = x; key_y[m] = y; depth_out[m] = d; sem_out[m] = sl; flow_x[m] = flx; flow_y[m] = fly; corr_x[m] = crx; corr_y[m] = cry;
    dyn_inlier_id[m] = inl; obj_label_out[m] = ol;
    if (xyz_out) { xyz_out[3 * m] = p3[0]; xyz_out[3 * m + 1] = p3[1]; xyz_out[3 * m + 2] = p3[2]; }
    ++m;
    return true;
  };
  std::vector<int> fea_count(n_obj, -1);
  for (int i = 0; i < n_obj; ++i) if (obj_stat[i]) fea_count[i] = 0;
  for (int k = 0; k < nc; ++k) {
    if (!ok[k]) continue;
    const int x = (int)cx[k], y = (int)cy[k];
    if (!push((float)x, (float)y, dd[k], sem[k], fx[k], fy[k], x + fx[k], y + fy[k], cid[k], cur_obj_label[cid[k]], xyz_c.data() + 3 * (size_t)k)) return set_error(VDO_ERR_INVALID, "vdo_renew_object: output capacity %d too small", cap);
    ++fea_count[cobj[k]];
  }
  // ---- top-up from the semi-dense sampling of the new image (used[]: a carried point within 1 px)
  for (int i = 0; i < n_obj; ++i) {
    if (!obj_stat[i]) continue;
    int tot = fea_count[i];
    for (int start = 0; start < 15 && tot < max_num_obj; ++start) {
      for (int j = start; j < n_tmp; j += 15) {
        if (tmp_label[j] != sem_pos[i] || used[j]) continue;
        if (!push(tmp_x[j], tmp_y[j], tmp_depth[j], tmp_label[j], tmp_flow_x[j], tmp_flow_y[j], tmp_corr_x[j], tmp_corr_y[j], -1, mod_label[i], xyz_t.data() + 3 * (size_t)j))
          return set_error(VDO_ERR_INVALID, "vdo_renew_object: output capacity %d too small", cap);
        if (++tot >= max_num_obj) break;
      }
    }
  }
  // ---- labels the tracker does not follow yet: all their sampled points, object label -2
  std::vector<int32_t> uni, slot;
  label_slots(n_tmp, tmp_label, uni, slot);
  std::vector<char> known(uni.size(), 0);
  for (int i = 0; i < n_obj; ++i) {
    if (!obj_stat[i]) continue;
    auto it = std::lower_bound(uni.begin(), uni.end(), sem_pos[i]);
    if (it != uni.end() && *it == sem_pos[i]) known[it - uni.begin()] = 1;
  }
  for (size_t s = 0; s < uni.size(); ++s) {
    if (known[s]) continue;
    for (int j = 0; j < n_tmp; ++j)
      if (slot[j] == (int)s && !push(tmp_x[j], tmp_y[j], tmp_depth[j], tmp_label[j], tmp_flow_x[j], tmp_flow_y[j], tmp_corr_x[j], tmp_corr_y[j], -1, -2, xyz_t.data() + 3 * (size_t)j))
        return set_error(VDO_ERR_INVALID, "vdo_renew_object: output capacity %d too small", cap);
  }
  *n_out = m;
  return VDO_OK;
}

extern "C" int vdo_renew_object(vdo_frame_images* f, int n_obj, const int32_t* inl_off, const int32_t* inl_idx, const uint8_t* obj_stat,
                                const int32_t* sem_pos, const int32_t* mod_label,
                                const float* cur_x, const float* cur_y, const int32_t* cur_obj_label,
                                int n_tmp, const float* tmp_x, const float* tmp_y, const float* tmp_depth, const int32_t* tmp_label,
                                const float* tmp_flow_x, const float* tmp_flow_y, const float* tmp_corr_x, const float* tmp_corr_y,
                                int max_num_obj, int cap,
                                float* key_x, float* key_y, float* depth_out, int32_t* sem_out, float* flow_x, float* flow_y,
                                float* corr_x, float* corr_y, int32_t* dyn_inlier_id, int32_t* obj_label_out, int* n_out) {
  return vdo_renew_object_world(f, n_obj, inl_off, inl_idx, obj_stat, sem_pos, mod_label, cur_x, cur_y, cur_obj_label, n_tmp, tmp_x, tmp_y, tmp_depth, tmp_label,
                                tmp_flow_x, tmp_flow_y, tmp_corr_x, tmp_corr_y, max_num_obj, cap, nullptr, nullptr,
                                key_x, key_y, depth_out, sem_out, flow_x, flow_y, corr_x, corr_y, dyn_inlier_id, obj_label_out, nullptr, n_out);
}

extern "C" int vdo_update_mask(vdo_frame_images* cur, vdo_frame_images* last, int n, const int32_t* last_sem_label,
                               const float* last_corr_x, const float* last_corr_y, int* n_recovered) {
  if (!cur || !last || cur->w != last->w || cur->h != last->h || n < 0) return set_error(VDO_ERR_INVALID, "vdo_update_mask: bad argument");
  if (n_recovered) *n_recovered = 0;
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(cur->ctx);
  if (rc != VDO_OK) return rc;
  Arena S(cur->ctx);
  if (!S.reserve(Arena::bytes_for(4 * (size_t)n + 512))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  // group the flowed positions by last-frame label (ascending labels, index order inside a label)
  std::vector<int32_t> uni, slot;
  label_slots(n, last_sem_label, uni, slot);
  const int L = (int)uni.size();
  std::vector<int> off(L + 1, 0);
  for (int i = 0; i < n; ++i) off[slot[i] + 1]++;
  for (int s = 0; s < L; ++s) off[s + 1] += off[s];
  std::vector<float> gx(n), gy(n);
  {
    std::vector<int> c(off.begin(), off.end() - 1);
    for (int i = 0; i < n; ++i) { const int p = c[slot[i]]++; gx[p] = last_corr_x[i]; gy[p] = last_corr_y[i]; }
  }
  float *dx = S.up(gx.data(), n), *dy = S.up(gy.data(), n);
  std::vector<int32_t> off32(off.begin(), off.end());
  int32_t* doff = S.up(off32.data(), off32.size());
  int32_t* dflag = S.up<int32_t>(nullptr, 2 * (size_t)L);
  unsigned long long* drec = S.up<unsigned long long>(nullptr, 1);
  if (!dflag || !doff || !drec) return set_error(VDO_ERR_OOM, "hipMalloc failed");
  launch_update_mask(cur, last, uni, off, dx, dy, doff, dflag, drec, S.stream());          // stream-ordered, no host sync inside
  std::vector<int32_t> flag(2 * (size_t)L);
  S.down(flag.data(), dflag, flag.size());
  rc = S.finish("vdo_update_mask");
  if (rc != VDO_OK) return rc;
  int rec = 0;
  for (int s = 0; s < L; ++s) {
    if (flag[2 * s + 1]) return set_error(VDO_ERR_UNSUPPORTED, "vdo_update_mask: mask label outside [0,%d)", kVoteBins);
    rec += flag[2 * s];
  }
  if (n_recovered) *n_recovered = rec;
  return VDO_OK;
}

// UpdateMask (K15) -> object propagation (K11) -> GetSceneFlowObj (K13) as ONE call: the three steps of the object chain between
// the renewed object set of the last frame and DynObjTracking.  Same kernels, same order on the stream as vdo_update_mask +
// vdo_propagate_object + vdo_scene_flow (which remain), but one staged upload, one download, one synchronisation instead of three.
// K11 (objects) + K13 in one launch: depth / label of the updated mask under the correspondence (k_gather mode 1), then the scene flow of the
// point from those two values (k_scene_flow) - same arithmetic, one launch and one pass over the inputs less
static __global__ void k_gather_scene_flow(int n, const float* __restrict__ kx, const float* __restrict__ ky, const float* __restrict__ depth, const int32_t* __restrict__ mask,
                                           int w, int h, float th, float* __restrict__ dout, int32_t* __restrict__ lout, Cam cc,
                                           const float* __restrict__ lx, const float* __restrict__ ly, const float* __restrict__ ld, const int32_t* __restrict__ ll, Cam lc,
                                           float* __restrict__ flow3d, int32_t* __restrict__ objlab, const int32_t* __restrict__ olab_init,
                                           const int32_t* __restrict__ flag_in, int32_t* __restrict__ flag_out, int nflag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nflag) flag_out[i] = flag_in[i];              // (the UpdateMask verdicts of the launches before this one, into the caller's - mapped - output block)
  if (i >= n) return;
  const float xf = kx[i], yf = ky[i];
  const int u = (int)xf, v = (int)yf;
  float o = 0.1f; int l = 0;
  {
    const bool inside = u < (w - 1) && u > 0 && v < (h - 1) && v > 0;
    const size_t q = inside ? (size_t)v * w + u : 0;
    const float d = depth[q];                              // (both words requested together, unconditionally: one trip to memory behind the position instead of two)
    const int ml = mask[q];
    if (inside && d < th && d > 0) { o = d; l = ml; }
  }
  dout[i] = o; lout[i] = l;
  if (l <= 0 || ll[i] <= 0) { objlab[i] = -1; flow3d[3 * i] = 0; flow3d[3 * i + 1] = 0; flow3d[3 * i + 2] = 0; return; }
  float p[3], c[3];
  backproject(lc, lx[i], ly[i], ld[i], p);
  backproject(cc, xf, yf, o, c);
  flow3d[3 * i] = c[0] - p[0]; flow3d[3 * i + 1] = c[1] - p[1]; flow3d[3 * i + 2] = c[2] - p[2];
  objlab[i] = olab_init ? olab_init[i] : -2;           // (the label DynObjTracking starts from; rounds 1-4 uploaded a buffer of -2 for the kernel to leave alone)
}

// The inputs of the object chain that belong to the LAST frame - its object set: labels, correspondences, key points, depths - are known when that frame's
// object stage ends, a whole frame before the chain runs.  vdo_object_chain_prestage groups them by label and sends them to a block of the context that only the
// chain reads (asynchronously, on the context's stream: the next frame's kernels are ordered behind it); vdo_object_chain finds them there - after checking, value
// by value against the pinned mirror, that they are what it was called with - and starts with its first kernel instead of a grouping pass, a staged copy and the
// copy's ~15 us of stream time at the head of the frame.  Optional: without it (or when anything differs) the chain stages its inputs itself, as before.
namespace {
struct ChainLayout { size_t gx, gy, cx, cy, lx, ly, ld, ll, off, total; };
ChainLayout chain_layout(int n) {
  auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
  ChainLayout q{};
  size_t o = 0;
  const size_t b = al(4 * (size_t)n);
  q.gx = o; o += b; q.gy = o; o += b; q.cx = o; o += b; q.cy = o; o += b; q.lx = o; o += b; q.ly = o; o += b; q.ld = o; o += b; q.ll = o; o += b;
  q.off = o; o += al(4 * 65);
  q.total = o;
  return q;
}
// label slots of the n samples + their (x, y) grouped by slot; returns the number of labels L (uni, off: L and L + 1 entries)
int group_by_label(int n, const int32_t* lab, const float* px, const float* py, std::vector<int32_t>& uni, std::vector<int>& off, float* gx, float* gy) {
  static thread_local std::vector<int32_t> slot;
  static thread_local std::vector<int> c;
  label_slots(n, lab, uni, slot);
  const int L = (int)uni.size();
  off.assign(L + 1, 0);
  for (int i = 0; i < n; ++i) off[slot[i] + 1]++;
  for (int s = 0; s < L; ++s) off[s + 1] += off[s];
  c.assign(off.begin(), off.end() - 1);
  for (int i = 0; i < n; ++i) { const int p = c[slot[i]]++; gx[p] = px[i]; gy[p] = py[i]; }
  return L;
}
}  // namespace

extern "C" int vdo_object_chain_prestage(vdo_ctx* ctx, int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                                         const float* last_x, const float* last_y, const float* last_d) {
  if (!ctx || n < 0) return set_error(VDO_ERR_INVALID, "vdo_object_chain_prestage: bad argument");
  ctx->stage_n = -1;
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  const ChainLayout q = chain_layout(n);
  if (q.total > ctx->stage_cap) {
    hipStreamSynchronize(ctx->stream);
    if (ctx->d_stage) hipFree(ctx->d_stage);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    ctx->d_stage = ctx->h_stage = nullptr; ctx->stage_cap = 0;
    const size_t want = std::max<size_t>(2 * q.total, size_t(1) << 20);
    if (hipMalloc((void**)&ctx->d_stage, want) != hipSuccess || hipHostMalloc((void**)&ctx->h_stage, want) != hipSuccess) return set_error(VDO_ERR_OOM, "vdo_object_chain_prestage: allocation failed");
    ctx->stage_cap = want;
  }
  static thread_local std::vector<int32_t> uni;
  static thread_local std::vector<int> off;
  char* h = ctx->h_stage;
  const int L = group_by_label(n, last_sem_label, last_corr_x, last_corr_y, uni, off, (float*)(h + q.gx), (float*)(h + q.gy));
  if (L > 64) return VDO_OK;                              // (more labels than the one-launch form of UpdateMask takes: the chain stages for itself)
  std::memcpy(h + q.cx, last_corr_x, 4 * (size_t)n); std::memcpy(h + q.cy, last_corr_y, 4 * (size_t)n);
  std::memcpy(h + q.lx, last_x, 4 * (size_t)n); std::memcpy(h + q.ly, last_y, 4 * (size_t)n); std::memcpy(h + q.ld, last_d, 4 * (size_t)n);
  std::memcpy(h + q.ll, last_sem_label, 4 * (size_t)n);
  int32_t* ho = (int32_t*)(h + q.off);
  for (int s = 0; s <= L; ++s) { ho[s] = off[s]; ctx->stage_off[s] = off[s]; }
  for (int s = 0; s < L; ++s) ctx->stage_uni[s] = uni[s];
  if (hipMemcpyAsync(ctx->d_stage, h, q.total, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "vdo_object_chain_prestage: copy failed");
  ctx->stage_L = L; ctx->stage_n = n;
  return VDO_OK;
}

extern "C" int vdo_object_chain(vdo_frame_images* cur, vdo_frame_images* last, int n, const int32_t* last_sem_label, const float* last_corr_x, const float* last_corr_y,
                                float th_depth_obj, const float Tcw_cur[16], const float* last_x, const float* last_y, const float* last_d, const float Tcw_last[16],
                                const float K4[4], int* n_recovered, float* depth_out, int32_t* sem_out, float* flow3d_out, int32_t* obj_label_out) {
  if (!cur || !last || cur->w != last->w || cur->h != last->h || n < 0) return set_error(VDO_ERR_INVALID, "vdo_object_chain: bad argument");
  if (n_recovered) *n_recovered = 0;
  if (n == 0) return VDO_OK;
  int rc = ctx_bind(cur->ctx);
  if (rc != VDO_OK) return rc;
  vdo_ctx* ctx = cur->ctx;
  // VDO_CHAIN_TRACE=1 (debugging): where the wall time of this call goes, every 20 calls on stderr
  static const bool tr_on = std::getenv("VDO_CHAIN_TRACE") != nullptr;
  static thread_local double tr_acc[6] = {0, 0, 0, 0, 0, 0}; static thread_local int tr_n = 0;
  auto tr_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tr0 = tr_on ? tr_now() : 0.0;
  Arena S(ctx);
  if (!S.reserve(Arena::bytes_for(16 * (size_t)n + 512))) return set_error(VDO_ERR_OOM, "scratch arena: allocation failed");
  static thread_local std::vector<int32_t> uni, flag;       // (per-thread scratch: no allocation in steady state)
  static thread_local std::vector<int> off;
  static thread_local std::vector<float> gx, gy;
  const float *dgx, *dgy, *dcx, *dcy, *dlx, *dly, *dld;
  const int32_t *dll, *doff;
  int L;
  // ---- the last frame's half of the inputs: staged ahead (and still what the caller passes), or staged now
  const ChainLayout q = chain_layout(n);
  bool staged = ctx->stage_n == n && ctx->d_stage && std::getenv("VDO_PIPE_NO_CHAIN_PRESTAGE") == nullptr;
  if (staged) {
    const char* h = ctx->h_stage;
    staged = std::memcmp(h + q.ll, last_sem_label, 4 * (size_t)n) == 0 && std::memcmp(h + q.cx, last_corr_x, 4 * (size_t)n) == 0 && std::memcmp(h + q.cy, last_corr_y, 4 * (size_t)n) == 0 &&
             std::memcmp(h + q.lx, last_x, 4 * (size_t)n) == 0 && std::memcmp(h + q.ly, last_y, 4 * (size_t)n) == 0 && std::memcmp(h + q.ld, last_d, 4 * (size_t)n) == 0;
  }
  ctx->stage_n = -1;                                       // (one use)
  if (staged) {
    const char* dv = ctx->d_stage;
    dgx = (const float*)(dv + q.gx); dgy = (const float*)(dv + q.gy); dcx = (const float*)(dv + q.cx); dcy = (const float*)(dv + q.cy);
    dlx = (const float*)(dv + q.lx); dly = (const float*)(dv + q.ly); dld = (const float*)(dv + q.ld); dll = (const int32_t*)(dv + q.ll); doff = (const int32_t*)(dv + q.off);
    L = ctx->stage_L;
    uni.assign(ctx->stage_uni, ctx->stage_uni + L); off.assign(ctx->stage_off, ctx->stage_off + L + 1);
  } else {
    // K15: the flowed positions grouped by last-frame label (ascending labels, index order inside a label); every input in one run of staged buffers -> one H2D copy
    gx.resize(n); gy.resize(n);
    L = group_by_label(n, last_sem_label, last_corr_x, last_corr_y, uni, off, gx.data(), gy.data());
    static thread_local std::vector<int32_t> off32;
    off32.assign(off.begin(), off.end());
    dgx = S.up(gx.data(), n); dgy = S.up(gy.data(), n);
    dcx = S.up(last_corr_x, n); dcy = S.up(last_corr_y, n);
    dlx = S.up(last_x, n); dly = S.up(last_y, n); dld = S.up(last_d, n);
    dll = S.up(last_sem_label, n);
    doff = S.up(off32.data(), off32.size());
    if (!dgx || !dgy || !dcx || !dcy || !dlx || !dly || !dld || !dll || !doff) return set_error(VDO_ERR_OOM, "scratch arena exhausted");
  }
  unsigned long long* drec = S.up<unsigned long long>(nullptr, 1);
  int32_t* dflag = S.up<int32_t>(nullptr, 2 * (size_t)L);  // (read by other workgroups of the votes: device memory; the last kernel copies it out)
  // outputs: written once by the last kernel, straight into the pinned block (Arena::out) - no device -> host copy
  int32_t* oflag = S.out<int32_t>(2 * (size_t)L);
  float* ddep = S.out<float>(n);
  int32_t* dsem = S.out<int32_t>(n);
  float* dfl = S.out<float>(3 * (size_t)n);
  int32_t* dol = S.out<int32_t>(n);
  if (!drec || !dflag || !oflag || !ddep || !dsem || !dfl || !dol) return set_error(VDO_ERR_OOM, "scratch arena exhausted");
  const double tr1 = tr_on ? tr_now() : 0.0;
  launch_update_mask(cur, last, uni, off, dgx, dgy, doff, dflag, drec, S.stream());
  // K11 (objects) on the updated mask, K13 on its outputs
  const int nth = std::max(n, 2 * L);
  hipLaunchKernelGGL(k_gather_scene_flow, dim3((nth + 255) / 256), dim3(256), 0, S.stream(), n, dcx, dcy, (const float*)cur->d_depth, (const int32_t*)cur->d_mask,
                     cur->w, cur->h, th_depth_obj, ddep, dsem, make_cam_Tcw(K4, Tcw_cur),
                     dlx, dly, dld, dll, make_cam_Tcw(K4, Tcw_last), dfl, dol, (const int32_t*)nullptr, (const int32_t*)dflag, oflag, 2 * L);
  flag.assign(2 * (size_t)L, 0);
  double tr2 = 0.0, tr3 = 0.0;
  if (tr_on) { tr2 = tr_now(); hipStreamSynchronize(ctx->stream); tr3 = tr_now(); }
  S.down(flag.data(), oflag, flag.size()); S.down(depth_out, ddep, n); S.down(sem_out, dsem, n); S.down(flow3d_out, dfl, 3 * (size_t)n); S.down(obj_label_out, dol, n);
  rc = S.finish("vdo_object_chain");
  if (tr_on) { const double tr4 = tr_now(); tr_acc[0] += tr1 - tr0; tr_acc[1] += tr2 - tr1; tr_acc[2] += tr3 - tr2; tr_acc[3] += tr4 - tr3; tr_acc[4] += staged ? 1 : 0;
    if (++tr_n % 20 == 0) { std::fprintf(stderr, "vdo_object_chain: n %d L %d staged %.0f%% | prep %.1f us, launches enqueued %.1f, wait for the kernels %.1f, copy out %.1f\n", n, L, 100 * tr_acc[4] / 20, tr_acc[0] / 20, tr_acc[1] / 20, tr_acc[2] / 20, tr_acc[3] / 20); for (double& a : tr_acc) a = 0; } }
  if (rc != VDO_OK) return rc;
  int rec = 0;
  for (int s = 0; s < L; ++s) {
    if (flag[2 * s + 1]) return set_error(VDO_ERR_UNSUPPORTED, "vdo_object_chain: mask label outside [0,%d)", kVoteBins);
    rec += flag[2 * s];
  }
  if (n_recovered) *n_recovered = rec;
  return VDO_OK;
}

// ---- tracklets, incrementally: each frame only looks at its own association vector --------------------
struct vdo_tracks {
  bool with_label = false;
  int n_frames = 0;
  std::vector<int32_t> pre, cur;                  // track id of every feature of the last added frame (+ the buffer the next frame fills)
  // (frame, feature) pairs of every track as singly linked chains in ONE flat append-only pool: a frame adds a few thousand pairs
  // (one small heap vector per track made that a few thousand allocations per frame; three parallel pools three cache lines per pair)
  struct Pair { int32_t frame, feat, next; };
  struct Track { int32_t first, last, len; };
  std::vector<Pair> pool;
  std::vector<Track> trk;
  std::vector<int32_t> obj_id;
  int64_t n_pairs = 0;
  void add_pair(int id, int frame, int feat) {
    const int k = (int)pool.size();
    pool.push_back(Pair{frame, feat, -1});
    Track& t = trk[id];
    if (t.last >= 0) pool[t.last].next = k; else t.first = k;
    t.last = k; t.len += 1;
  }
};

extern "C" int vdo_tracks_create(int with_object_label, vdo_tracks** out) {
  if (!out) return set_error(VDO_ERR_INVALID, "null out");
  vdo_tracks* t = new vdo_tracks();
  t->with_label = with_object_label != 0;
  // address space up front (pages are touched as the pools fill): a doubling std::vector re-allocates and copies several MB now
  // and then - a multi-millisecond frame in the middle of a sequence
  t->pool.reserve((size_t)1 << 22); t->trk.reserve((size_t)1 << 20);
  if (t->with_label) t->obj_id.reserve((size_t)1 << 20);
  *out = t;
  return VDO_OK;
}
extern "C" int vdo_tracks_destroy(vdo_tracks* t) { delete t; return VDO_OK; }

// asso[j] = index, in the previous frame's feature list, of the feature matched to feature j (-1: none).
extern "C" int vdo_tracks_add_frame(vdo_tracks* t, int n, const int32_t* asso, const int32_t* feat_label) {
  if (!t || n < 0 || (n > 0 && !asso) || (t->with_label && n > 0 && !feat_label)) return set_error(VDO_ERR_INVALID, "vdo_tracks_add_frame: bad argument");
  const int i = t->n_frames;
  std::vector<int32_t>& cur = t->cur;
  cur.assign(n, -1);
  for (int j = 0; j < n; ++j) {
    const int a = asso[j];
    if (a == -1) continue;
    if (i > 0 && (a < 0 || a >= (int)t->pre.size())) return set_error(VDO_ERR_INVALID, "frame %d feature %d: association %d out of range", i, j, a);
    if (i > 0 && t->pre[a] != -1) {
      const int id = t->pre[a];
      t->add_pair(id, i + 1, j);
      cur[j] = id; t->n_pairs += 1;
    } else {
      const int id = (int)t->trk.size();
      t->trk.push_back(vdo_tracks::Track{-1, -1, 0});
      t->add_pair(id, i, a); t->add_pair(id, i + 1, j);
      if (t->with_label) t->obj_id.push_back(feat_label[j]);
      cur[j] = id; t->n_pairs += 2;
    }
  }
  t->pre.swap(cur);
  t->n_frames = i + 1;
  return VDO_OK;
}

extern "C" int vdo_tracks_size(vdo_tracks* t, int* n_tracks, int64_t* n_pairs) {
  if (!t) return set_error(VDO_ERR_INVALID, "null handle");
  if (n_tracks) *n_tracks = (int)t->trk.size();
  if (n_pairs) *n_pairs = t->n_pairs;
  return VDO_OK;
}

// The tracks that are still observed in frame `first_frame` or later, whole (every pair since their start) and in creation order: what a WINDOW of the sequence needs
// (PartialBatchOptimization: labels of the window's features, position inside the track, the observation before) without walking the chains of all the tracks that ended
// before it - the cost follows the window, not the length of the sequence.  Buffers sized as for vdo_tracks_get; the counts actually written come back.
extern "C" int vdo_tracks_get_since(vdo_tracks* t, int first_frame, int* n_tracks, int64_t* n_pairs, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id) {
  if (!t || !n_tracks || !n_pairs || !track_off || !pair_frame || !pair_feat) return set_error(VDO_ERR_INVALID, "null argument");
  int off = 0, nt = 0;
  track_off[0] = 0;
  for (size_t k = 0; k < t->trk.size(); ++k) {
    const vdo_tracks::Track& tk = t->trk[k];
    if (tk.last < 0 || t->pool[tk.last].frame < first_frame) continue;
    for (int q = tk.first; q >= 0; q = t->pool[q].next) { pair_frame[off] = t->pool[q].frame; pair_feat[off] = t->pool[q].feat; ++off; }
    if (obj_id && t->with_label) obj_id[nt] = t->obj_id[k];
    track_off[++nt] = off;
  }
  *n_tracks = nt; *n_pairs = off;
  return VDO_OK;
}

extern "C" int vdo_tracks_get(vdo_tracks* t, int32_t* track_off, int32_t* pair_frame, int32_t* pair_feat, int32_t* obj_id) {
  if (!t || !track_off || !pair_frame || !pair_feat) return set_error(VDO_ERR_INVALID, "null argument");
  int off = 0;
  track_off[0] = 0;
  for (size_t k = 0; k < t->trk.size(); ++k) {
    for (int q = t->trk[k].first; q >= 0; q = t->pool[q].next) { pair_frame[off] = t->pool[q].frame; pair_feat[off] = t->pool[q].feat; ++off; }
    track_off[k + 1] = off;
    if (obj_id && t->with_label) obj_id[k] = t->obj_id[k];
  }
  return VDO_OK;
}
