// Host-only: landmark-track partition of a batch-BA graph across ranks.
// A track = one static point, or the chain of per-frame points of one dynamic feature linked by
// LandmarkMotionTernaryEdges (reference graph builder src/Optimizer.cc:1704-1741).  Tracks are
// ordered by the first pose that observes them (the same key the tiling in capi_ba.hip uses, so a
// shard's tiles keep their pose locality) and cut into `world` contiguous runs of near-equal
// incidence count (the unit of work of the sweep and of every PCG mat-vec).
#include <algorithm>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"

using namespace vdo;

extern "C" int vdo_ba_partition(const vdo_ba_graph* g, int world, int32_t* owner) {
  if (!g || !owner || world < 1) return set_error(VDO_ERR_INVALID, "vdo_ba_partition: bad argument");
  const int P = g->n_pose, L = g->n_point, Eb = g->n_eb, Et = g->n_et;
  for (int e = 0; e < Eb; ++e)
    if ((unsigned)g->eb_pose[e] >= (unsigned)P || (unsigned)g->eb_point[e] >= (unsigned)L) return set_error(VDO_ERR_INVALID, "binary edge %d: index out of range", e);
  std::vector<int32_t> next_e(L, -1), prev_e(L, -1);
  for (int e = 0; e < Et; ++e) {
    if ((unsigned)g->et_p1[e] >= (unsigned)L || (unsigned)g->et_p2[e] >= (unsigned)L || (unsigned)g->et_pose[e] >= (unsigned)P)
      return set_error(VDO_ERR_INVALID, "ternary edge %d: index out of range", e);
    if (next_e[g->et_p1[e]] != -1 || prev_e[g->et_p2[e]] != -1) return set_error(VDO_ERR_UNSUPPORTED, "ternary edge %d: landmark tracks must be simple chains", e);
    next_e[g->et_p1[e]] = e;
    prev_e[g->et_p2[e]] = e;
  }
  std::vector<int32_t> deg(L, 0), first(L, P);
  for (int e = 0; e < Eb; ++e) { deg[g->eb_point[e]]++; first[g->eb_point[e]] = std::min(first[g->eb_point[e]], g->eb_pose[e]); }
  struct Track { int32_t head, key; int64_t w; };
  std::vector<Track> tracks;
  int visited = 0;
  int64_t total = 0;
  for (int l = 0; l < L; ++l) {
    if (prev_e[l] != -1) continue;
    Track t{l, P, 0};
    for (int c = l;;) {
      ++visited;
      t.w += deg[c] + 1;                       // +1: the point itself (Hll block, back-substitution)
      t.key = std::min(t.key, first[c]);
      const int e = next_e[c];
      if (e == -1) break;
      t.w += 2;
      c = g->et_p2[e];
    }
    total += t.w;
    tracks.push_back(t);
  }
  if (visited != L) return set_error(VDO_ERR_UNSUPPORTED, "ternary edges form a cycle");
  std::stable_sort(tracks.begin(), tracks.end(), [](const Track& a, const Track& b) { return a.key < b.key; });
  int64_t acc = 0;
  for (const Track& t : tracks) {
    // rank r owns the tracks whose running weight midpoint falls in [r, r+1) * total / world
    int r = total > 0 ? (int)(((acc + t.w / 2) * world) / total) : 0;
    r = std::min(r, world - 1);
    acc += t.w;
    for (int c = t.head;;) {
      owner[c] = r;
      const int e = next_e[c];
      if (e == -1) break;
      c = g->et_p2[e];
    }
  }
  return VDO_OK;
}
