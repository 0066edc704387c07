// C-ABI entry points (include/vdo_slam_hip.h) of the batch-BA path: graph upload with the
// host-side re-ordering into tiles (ba_dev.hpp), linearisation, system download, estimates.
// The Levenberg–Marquardt driver is in ba_lm.hip.
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include <chrono>

#include "ba_host.hpp"

namespace vdo {
constexpr size_t kBaSlabBytes = (size_t)48 << 20;      // the context's slab for small batch handles (a 20-frame window of KITTI takes ~6 MB)

// device memory of a handle: from the context's slab while the handle owns the pool and the block fits, else hipMalloc (freed by vdo_ba_destroy)
void* ba_device_alloc(vdo_ba* ba, size_t bytes) {
  vdo_ctx* c = ba->ctx;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (ba->pooled && c->ba_slab && c->ba_slab_used + need <= c->ba_slab_cap) {
    void* p = c->ba_slab + c->ba_slab_used;
    c->ba_slab_used += need;
    return p;
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  ba->allocs.push_back(p);
  return p;
}
}  // namespace vdo

using namespace vdo;

namespace {

template <class T>
int upload(vdo_ba* ba, T** dst, const T* src, size_t n, hipStream_t s) {
  if (n == 0) { *dst = nullptr; return VDO_OK; }
  *dst = (T*)ba_device_alloc(ba, n * sizeof(T));
  if (!*dst) return set_error(VDO_ERR_OOM, "hipMalloc(%zu) failed", n * sizeof(T));
  if (src) {
    if (hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "H2D copy failed");
  } else {
    hipMemsetAsync(*dst, 0, n * sizeof(T), s);
  }
  return VDO_OK;
}

constexpr int kSoftSlots = 64;     // normal tiles stay below this many pose slots
constexpr int kHardSlots = 512;    // a single long DYNAMIC track may use up to this many (a chain of n points touches n cameras + n - 1 motion vertices: n <= 256 = VDO_TILE_PTS): the tile kernels
                                   // stage slots in rounds of 256, and 511 slots are ~140 KB of LDS (ONE workgroup per CU - paid only by graphs that hold such a track).  Rounds 1-4: 100; round 5: 256.
constexpr int kStaticSlots = 256;  // a STATIC point beyond this many pose vertices is a hub landmark (ba_hub.hip: no LDS at all) instead of a tile of its own

}  // namespace

#define UP(field, src, n)                                                          \
  do {                                                                             \
    int rc_ = upload(ba, &ba->d.field, src, (size_t)(n), s);                       \
    if (rc_ != VDO_OK) { vdo_ba_destroy(ba); return rc_; }                          \
  } while (0)

extern "C" int vdo_ba_create(vdo_ctx* ctx, const vdo_ba_graph* g, vdo_ba** out) {
  if (!ctx || !g || !out) return set_error(VDO_ERR_INVALID, "vdo_ba_create: null argument");
  if (g->n_pose <= 0 || g->n_point < 0 || g->n_eb < 0 || g->n_et < 0 || g->n_ep < 0 || g->n_prior < 0)
    return set_error(VDO_ERR_INVALID, "vdo_ba_create: negative/empty sizes");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  const auto t_create0 = std::chrono::steady_clock::now();
  static const bool trace_create = std::getenv("VDO_BATCH_TRACE") != nullptr;
  double t_mark[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto mark_t = [&](int k) { if (trace_create) t_mark[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create0).count(); };
  const int P = g->n_pose, L = g->n_point, Eb = g->n_eb, Et = g->n_et, Ep = g->n_ep, Npr = g->n_prior;
  // ---- validate indices
  for (int e = 0; e < Eb; ++e)
    if ((unsigned)g->eb_pose[e] >= (unsigned)P || (unsigned)g->eb_point[e] >= (unsigned)L)
      return set_error(VDO_ERR_INVALID, "binary edge %d: index out of range", e);
  for (int e = 0; e < Et; ++e)
    if ((unsigned)g->et_pose[e] >= (unsigned)P || (unsigned)g->et_p1[e] >= (unsigned)L || (unsigned)g->et_p2[e] >= (unsigned)L || g->et_p1[e] == g->et_p2[e])
      return set_error(VDO_ERR_INVALID, "ternary edge %d: index out of range", e);
  for (int e = 0; e < Ep; ++e)
    if ((unsigned)g->ep_i[e] >= (unsigned)P || (unsigned)g->ep_j[e] >= (unsigned)P || g->ep_i[e] == g->ep_j[e])
      return set_error(VDO_ERR_INVALID, "pose-pose edge %d: index out of range", e);
  for (int e = 0; e < Npr; ++e)
    if ((unsigned)g->pr_pose[e] >= (unsigned)P) return set_error(VDO_ERR_INVALID, "prior %d: index out of range", e);

  // ---- robust-kernel widths: g2o keeps delta^2 in a FLOAT member (robust_kernel_impl.h:84); the tile kernels' Huber weight (se3_dev.hpp huber_dev) takes its
  // square root without range scaling and relies on e > dsqr being a NORMAL number: a positive delta whose float square underflows (delta below ~1.1e-19) is refused
  // here - no camera, depth or motion residual is measured in units where that is a width
  for (const double hd : {g->huber_eb, g->huber_et, g->huber_ep})
    if (hd > 0 && !((double)(float)(hd * hd) >= 1.1754943508222875e-38))
      return set_error(VDO_ERR_INVALID, "vdo_ba_create: Huber width %.3g: its square is not a normal float (RobustKernelHuber keeps it in one)", hd);

  mark_t(0);
  // ---- chains (dynamic tracks) from the ternary edges
  std::vector<int32_t> next_e(L, -1), prev_e(L, -1);
  for (int e = 0; e < Et; ++e) {
    if (next_e[g->et_p1[e]] != -1 || prev_e[g->et_p2[e]] != -1)
      return set_error(VDO_ERR_UNSUPPORTED, "ternary edge %d: landmark tracks must be simple chains", e);
    next_e[g->et_p1[e]] = e;
    prev_e[g->et_p2[e]] = e;
  }
  // point -> binary edges CSR
  std::vector<int32_t> pb_off(L + 1, 0), pb_idx(Eb);
  for (int e = 0; e < Eb; ++e) pb_off[g->eb_point[e] + 1]++;
  for (int l = 0; l < L; ++l) pb_off[l + 1] += pb_off[l];
  {
    std::vector<int32_t> fill(pb_off.begin(), pb_off.end() - 1);
    for (int e = 0; e < Eb; ++e) pb_idx[fill[g->eb_point[e]]++] = e;
  }
  struct ChainInfo { int32_t head; int32_t npts; int32_t ninc; int32_t key; int32_t nb; };
  std::vector<ChainInfo> chains;
  chains.reserve(L);
  int visited = 0;
  for (int l = 0; l < L; ++l) {
    if (prev_e[l] != -1) continue;
    ChainInfo ci{l, 0, 0, P, 0};
    for (int cur = l;;) {
      ++ci.npts; ++visited;
      ci.ninc += pb_off[cur + 1] - pb_off[cur];
      ci.nb += pb_off[cur + 1] - pb_off[cur];
      for (int k = pb_off[cur]; k < pb_off[cur + 1]; ++k) ci.key = std::min(ci.key, g->eb_pose[pb_idx[k]]);
      const int e = next_e[cur];
      if (e == -1) break;
      ci.ninc += 2;
      cur = g->et_p2[e];
    }
    chains.push_back(ci);
  }
  if (visited != L) return set_error(VDO_ERR_UNSUPPORTED, "ternary edges form a cycle");
  // Order of the tracks = order of the tiles' contents: by first observing frame; round 6: the DYNAMIC tracks (chains of several points) first, among themselves by first frame, then
  // the static points.  A dynamic track of n points brings 2 n - 1 pose vertices (its cameras and its motions) - more than the 64 slots a tile is closed at - and its neighbours in
  // time on the same object share nearly all of them; interleaved with the static points (rounds 1-5) every such track closed its tile behind itself and sat there alone: one lane
  // of 256 walking its chain in the solver's kernels, 70 points of 256 (the OMD-shaped graph: 11 k tiles of one track each).  VDO_BA_MIXED_ORDER=1: the old order (A/B).
  static const bool mixed_order = std::getenv("VDO_BA_MIXED_ORDER") != nullptr;
  std::stable_sort(chains.begin(), chains.end(), [](const ChainInfo& a, const ChainInfo& b) {
    if (!mixed_order) { const bool da = a.npts > 1, db = b.npts > 1; if (da != db) return da; }
    return a.key < b.key;
  });

  mark_t(1);
  // ---- greedy tiling
  vdo_ba* ba = new vdo_ba();
  ba->ctx = ctx;
  // a small graph takes the context's pool if nobody holds it (VDO_BA_NO_POOL: A/B switch); the slab itself is allocated on the first such graph
  static const bool pool_off = std::getenv("VDO_BA_NO_POOL") != nullptr;
  if (!pool_off && !ctx->ba_pool_busy && (int64_t)L + Eb + Et < 200000) {
    if (!ctx->ba_slab && hipMalloc((void**)&ctx->ba_slab, kBaSlabBytes) == hipSuccess) ctx->ba_slab_cap = kBaSlabBytes;
    if (ctx->ba_slab) { ctx->ba_pool_busy = true; ctx->ba_slab_used = 0; ba->pooled = true; }
    else (void)hipGetLastError();
  }
  std::vector<Tile> tiles;
  std::vector<int32_t> tile_pose;                  // per slot: global pose id
  std::vector<int32_t> chain_off{0};
  std::vector<int32_t> pt_old_of_new; pt_old_of_new.reserve(L);
  std::vector<int32_t> pt_new_of_old(L, -1);
  std::vector<int32_t> pose_stamp(P, -1);
  std::vector<int32_t> cur_poses;
  std::vector<int32_t> eb_old_of_new; eb_old_of_new.reserve(Eb);
  std::vector<int32_t> et_old_of_new; et_old_of_new.reserve(Et);
  std::vector<int32_t> eb_key, et_key(Et), et_slot(Et), inc_key;      // (eb_key / inc_key grow with the padded edge blocks of the tiles)
  eb_key.reserve((size_t)Eb + Eb / 8); inc_key.reserve((size_t)Eb + Eb / 8 + 2 * (size_t)Et);
  std::vector<int32_t> pt_prev_edge_new; pt_prev_edge_new.reserve(L);
  std::vector<int32_t> et_new_of_old(Et, -1);
  std::vector<int32_t> tile_eb, tile_et;           // original ids of the open tile
  std::vector<int32_t> slot_lut(std::max(P, 1), 0), sort_cnt, sort_slot, sort_tmp;      // close_tile: slot of a pose of the open tile, scratch of its counting sorts
  int max_slots = 1;
  Tile cur{};
  int cur_tile_id = 0, cur_npts = 0, cur_ninc = 0, cur_nb = 0;
  bool thr_overflow = false;
  int cur_need = 0;                                // threads the open tile needs: sum over its pose slots of ceil(edges / VDO_TILE_EPT)
  std::vector<int32_t> pose_cnt(P, 0), cnt_stamp(P, -1), chain_eb_poses;
  auto chain_poses = [&](const ChainInfo& ci, std::vector<int32_t>& outp) {
    outp.clear();
    for (int c = ci.head;;) {
      for (int k = pb_off[c]; k < pb_off[c + 1]; ++k) outp.push_back(g->eb_pose[pb_idx[k]]);
      const int e = next_e[c];
      if (e == -1) break;
      outp.push_back(g->et_pose[e]);
      c = g->et_p2[e];
    }
  };
  int inc_total = 0;
  // VDO_BA_PLACE=0: the edges of a slot's run in pose-sorted order (round 4) instead of the bank-aware placement below (A/B, tools/)
  // (round 6: a graph of a few tiles - the 20-frame windows: 8 k edges - is launch-bound whatever its bank conflicts are, and the placement was half of its 0.9 ms of tile building:
  //  below kPlaceMinInc incidences the edges keep their pose-sorted order)
  constexpr int64_t kPlaceMinInc = 32768;
  const int place_mode = std::getenv("VDO_BA_PLACE") ? std::atoi(std::getenv("VDO_BA_PLACE")) : ((int64_t)Eb + 2 * (int64_t)Et >= kPlaceMinInc ? 1 : 0);
  long long place_ways = 0, place_groups = 0;
  bool dense_tiles_ok = true;
  double t_close_ms = 0.0;
  auto close_tile = [&]() {
    if (cur_npts == 0) return;
    const auto t_close0 = std::chrono::steady_clock::now();
    // slots: sorted distinct poses
    std::sort(cur_poses.begin(), cur_poses.end());
    cur.slot_begin = (int32_t)tile_pose.size();
    for (int32_t p : cur_poses) tile_pose.push_back(p);
    cur.slot_end = (int32_t)tile_pose.size();
    max_slots = std::max(max_slots, cur.slot_end - cur.slot_begin);
    auto slot_of = [&](int32_t p) { return (int32_t)(std::lower_bound(cur_poses.begin(), cur_poses.end(), p) - cur_poses.begin()); };
    // the tile's edges in pose order, ties in the order they came (= std::stable_sort by pose; round 6: a counting sort by slot - the comparator's two random reads into
    // eb_pose per comparison made the sort 170 us per tile, 0.7 of the 0.9 s vdo_ba_create spent on the 1 M-point graph)
    {
      const int ns = (int)cur_poses.size();
      for (int k = 0; k < ns; ++k) slot_lut[cur_poses[k]] = k;
      auto by_slot = [&](std::vector<int32_t>& ids, const int32_t* pose_of) {
        if (ids.size() < 2) return;
        sort_cnt.assign((size_t)ns + 1, 0);
        sort_slot.resize(ids.size());
        for (size_t k = 0; k < ids.size(); ++k) { sort_slot[k] = slot_lut[pose_of[ids[k]]]; ++sort_cnt[(size_t)sort_slot[k] + 1]; }
        for (int k = 0; k < ns; ++k) sort_cnt[(size_t)k + 1] += sort_cnt[k];
        sort_tmp.resize(ids.size());
        for (size_t k = 0; k < ids.size(); ++k) sort_tmp[(size_t)sort_cnt[sort_slot[k]]++] = ids[k];
        ids.swap(sort_tmp);
      };
      by_slot(tile_eb, g->eb_pose);
      by_slot(tile_et, g->et_pose);
    }
    // EdgeSE3PointXYZ edges of the tile: a PADDED block of 256 x ept entries in thread-transposed order - entry j * 256 + t is the j-th edge of
    // thread t - so that the tile kernels need no thread table and every load of theirs is one contiguous 256-lane row (with the edges in
    // pose-sorted order and a table of first-edge indices, the six loads of a thread's edges touched the same 12 cache lines six times: the
    // head of a tile took 7 k cycles at six edges per thread).  Every thread takes <= ept consecutive edges (pose-sorted order) of ONE pose
    // slot: runs of equal slot are cut into pieces of <= ept, ept the smallest of 1 .. VDO_TILE_EPT that fits 256 threads; unused entries
    // carry key -1.
    cur.eb_begin = (int32_t)eb_old_of_new.size();
    cur.et_begin = (int32_t)et_old_of_new.size();
    cur.inc_begin = inc_total;
    const int nb_real = (int)tile_eb.size(), nt = (int)tile_et.size();
    int pb = 1;
    for (; pb < VDO_TILE_EPT; ++pb) {
      int need = 0;
      for (int j = 0; j < nb_real;) { int k = j; while (k < nb_real && g->eb_pose[tile_eb[k]] == g->eb_pose[tile_eb[j]]) ++k; need += (k - j + pb - 1) / pb; j = k; }
      if (need <= VDO_TILE_THREADS) break;
    }
    const int nb = nb_real ? VDO_TILE_THREADS * pb : 0;     // entries of the block
    cur.ept = nb_real ? pb : 0;
    eb_old_of_new.resize((size_t)cur.eb_begin + nb, -1);
    eb_key.resize((size_t)cur.eb_begin + nb, -1);
    inc_key.resize((size_t)inc_total + nb + 2 * (size_t)nt, -1);
    {
      // Which edge of a slot's run goes to which (thread, row) is free - a thread needs <= pb edges of ONE slot in rows 0 .. count - 1, nothing else -
      // and it decides the LDS bank conflicts of every tile kernel: row i of a wave is one LDS instruction per operand, 64 lanes at the local point
      // ids of their edges (point reads, the four landmark ds_add_f64 of the sweep, the factor reads of the solver's kernels).  The LDS serves
      // a wave in lane groups - 16 contiguous lanes for 64-bit stores / atomics (32 banks: point id mod 16), 32 for 64-bit reads (64 banks: id mod 32),
      // MI355X_MICROARCH.md LDS - and every extra distinct address on a bank costs a cycle: with the edges in pose-sorted order the ids of a group
      // are random (2.5 .. 3 addresses on the busiest bank; SQ_LDS_BANK_CONFLICT ~ SQ_ACTIVE_INST_LDS in profiles/r04_sweep_sq_counters.txt, the LDS
      // pipe busy ~85 % of the sweep).  So: rows are filled one after the other (thread counts stay balanced: ceil or floor of run / threads), and
      // every (thread, row) takes, of its slot's remaining edges, one whose point id collides with the fewest lanes already placed in its 16-lane
      // group and 32-lane half of that row.
      static thread_local std::vector<int> bucket[32];
      int occ16[VDO_TILE_THREADS / 64][VDO_TILE_EPT][4][16], occ32[VDO_TILE_THREADS / 64][VDO_TILE_EPT][2][32];
      std::memset(occ16, 0, sizeof occ16); std::memset(occ32, 0, sizeof occ32);
      int t = 0;
      for (int j = 0; j < nb_real;) {
        int k = j;
        while (k < nb_real && g->eb_pose[tile_eb[k]] == g->eb_pose[tile_eb[j]]) ++k;
        const int len = k - j, nthr = (len + pb - 1) / pb;
        if (t + nthr > VDO_TILE_THREADS) { thr_overflow = true; break; }
        for (int r = 0; r < 32; ++r) bucket[r].clear();
        uint32_t nonempty = 0;                               // (buckets that still hold an edge: a run of a dozen edges touches a dozen of the 32)
        for (int q = k - 1; q >= j; --q) { const int r = (pt_new_of_old[g->eb_point[tile_eb[q]]] - cur.pt_begin) & 31; bucket[r].push_back(tile_eb[q]); nonempty |= 1u << r; }     // (popped from the back: pose-sorted order among equals)
        const int32_t slot = slot_of(g->eb_pose[tile_eb[j]]);
        int left = len;
        for (int i = 0; i < pb && left > 0; ++i)
          for (int tau = 0; tau < nthr && left > 0; ++tau, --left) {
            const int T = t + tau, w = T >> 6, g16 = (T >> 4) & 3, h = (T >> 5) & 1;
            int best = -1, best_cost = 1 << 30;
            for (uint32_t m = nonempty; m; m &= m - 1) {    // (ascending bucket index, the first minimum wins: as the loop over all 32 did)
              const int r = __builtin_ctz(m);
              const int cost = place_mode ? 2 * occ16[w][i][g16][r & 15] + occ32[w][i][h][r] : 0;
              if (cost < best_cost) { best_cost = cost; best = r; if (cost == 0) break; }
            }
            const int e = bucket[best].back(); bucket[best].pop_back();
            if (bucket[best].empty()) nonempty &= ~(1u << best);
            ++occ16[w][i][g16][best & 15]; ++occ32[w][i][h][best];
            const int pos = i * VDO_TILE_THREADS + T;
            const int32_t key = (slot << 16) | (pt_new_of_old[g->eb_point[e]] - cur.pt_begin);
            eb_old_of_new[(size_t)cur.eb_begin + pos] = e;
            eb_key[(size_t)cur.eb_begin + pos] = key;
            inc_key[(size_t)inc_total + pos] = key;
          }
        t += nthr;
        j = k;
      }
      // refinement: the rows of ONE thread can be exchanged freely (same slot, same count) - a few passes of pairwise exchanges wherever that lowers
      // the collisions of the two group-rows involved
      if (place_mode == 1 && !thr_overflow) {        // (VDO_BA_PLACE=2: the greedy placement alone)
        const int nthr_used = t;
        auto lp_at = [&](int T, int i) { const int32_t key = eb_key[(size_t)cur.eb_begin + i * VDO_TILE_THREADS + T]; return key < 0 ? -1 : (key & 0xffff); };
        for (int pass = 0; pass < 3; ++pass) {
          int moved = 0;
          for (int T = 0; T < nthr_used; ++T) {
            const int w = T >> 6, g16 = (T >> 4) & 3, h = (T >> 5) & 1;
            int cnt = 0;
            while (cnt < pb && lp_at(T, cnt) >= 0) ++cnt;
            for (int a = 0; a < cnt; ++a) for (int b = a + 1; b < cnt; ++b) {
              const int la = lp_at(T, a), lb = lp_at(T, b);
              if ((la & 31) == (lb & 31)) continue;
              // cost of this thread's two entries where they are, and exchanged (occupancies without this thread's own entries)
              auto c16 = [&](int i, int l) { return occ16[w][i][g16][l & 15]; };
              auto c32 = [&](int i, int l) { return occ32[w][i][h][l & 31]; };
              const int now = 2 * (c16(a, la) - 1) + (c32(a, la) - 1) + 2 * (c16(b, lb) - 1) + (c32(b, lb) - 1);
              const int then = 2 * (c16(a, lb) - ((la & 15) == (lb & 15) ? 1 : 0)) + c32(a, lb) + 2 * (c16(b, la) - ((la & 15) == (lb & 15) ? 1 : 0)) + c32(b, la);
              if (then < now) {
                --occ16[w][a][g16][la & 15]; --occ32[w][a][h][la & 31]; --occ16[w][b][g16][lb & 15]; --occ32[w][b][h][lb & 31];
                ++occ16[w][a][g16][lb & 15]; ++occ32[w][a][h][lb & 31]; ++occ16[w][b][g16][la & 15]; ++occ32[w][b][h][la & 31];
                const size_t pa = (size_t)cur.eb_begin + a * VDO_TILE_THREADS + T, pbb = (size_t)cur.eb_begin + b * VDO_TILE_THREADS + T;
                std::swap(eb_old_of_new[pa], eb_old_of_new[pbb]); std::swap(eb_key[pa], eb_key[pbb]);
                std::swap(inc_key[(size_t)inc_total + a * VDO_TILE_THREADS + T], inc_key[(size_t)inc_total + b * VDO_TILE_THREADS + T]);
                ++moved;
              }
            }
          }
          if (!moved) break;
        }
      }
      for (int w = 0; w < VDO_TILE_THREADS / 64; ++w) for (int i = 0; i < pb; ++i) for (int g4 = 0; g4 < 4; ++g4) {      // (build statistics: the busiest bank of every 16-lane group-row)
        int mx = 0, any = 0;
        for (int r = 0; r < 16; ++r) { mx = std::max(mx, occ16[w][i][g4][r]); any += occ16[w][i][g4][r]; }
        if (any) { place_ways += mx; ++place_groups; }
      }
    }
    for (int j = 0; j < nt; ++j) {
      const int e = tile_et[j];
      const int en = (int)et_old_of_new.size();
      et_old_of_new.push_back(e);
      et_new_of_old[e] = en;
      const int32_t sl = slot_of(g->et_pose[e]);
      const int32_t l1 = pt_new_of_old[g->et_p1[e]] - cur.pt_begin, l2 = pt_new_of_old[g->et_p2[e]] - cur.pt_begin;
      et_key[en] = l1 | (l2 << 16);
      et_slot[en] = sl;
      inc_key[inc_total + nb + j] = (sl << 16) | l1;
      inc_key[inc_total + nb + nt + j] = (sl << 16) | l2;
    }
    inc_total += nb + 2 * nt;
    if (nb + 2 * nt > VDO_TILE_THREADS * (VDO_TILE_EPT + 2)) dense_tiles_ok = false;      // (the dense assembly keeps VDO_TILE_EPT + 2 incidences per thread - cannot happen: nb <= 1536, nt < 256; ba_lm.hip would refuse the solver)
    cur.eb_end = (int32_t)eb_old_of_new.size();
    cur.et_end = (int32_t)et_old_of_new.size();
    cur.pt_end = (int32_t)pt_old_of_new.size();
    cur.chain_end = (int32_t)chain_off.size() - 1;
    tiles.push_back(cur);
    ++cur_tile_id;
    cur_npts = 0; cur_ninc = 0; cur_nb = 0; cur_need = 0;
    cur_poses.clear(); tile_eb.clear(); tile_et.clear();
    if (trace_create) t_close_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_close0).count();
  };
  // Incidences a tile is CLOSED at (soft; a single track may still take up to VDO_TILE_INC): VDO_TILE_EPT per thread spreads what a tile costs
  // apart from its edges over more edges - right for graphs of many tiles; a small graph (the 60-frame window: 0.25 M incidences) would be left
  // with fewer tiles than the device has CUs, so it gets smaller tiles: about four tiles per CU of a 256-CU device, not fewer than 2 edges per thread.
  int soft_inc = VDO_TILE_INC;
  {
    const double total_inc = (double)Eb + 2.0 * (double)Et;
    int ept = (int)std::ceil(total_inc / (VDO_TILE_THREADS * 1024.0));
    if (const char* e = std::getenv("VDO_BA_TILE_EPT")) ept = std::atoi(e);
    soft_inc = VDO_TILE_THREADS * std::min(std::max(ept, 2), VDO_TILE_EPT);
    if (std::getenv("VDO_BA_TILE_EPT")) soft_inc = VDO_TILE_THREADS * std::min(std::max(ept, 1), VDO_TILE_EPT);
  }
  std::vector<int32_t> cposes;
  bool cur_dyn_only = true;                        // the open tile holds dynamic tracks only
  int dyn_slot_cap = 0;                            // distinct pose vertices of the graph's largest dynamic track
  if (!mixed_order)
    for (const ChainInfo& ci : chains) {
      if (ci.npts <= 1) break;                     // (dynamic tracks come first)
      chain_poses(ci, cposes);
      std::sort(cposes.begin(), cposes.end());
      dyn_slot_cap = std::max(dyn_slot_cap, (int)(std::unique(cposes.begin(), cposes.end()) - cposes.begin()));
    }
  // (packing beyond the largest track's slot count - 125 / 150 / 200 % probed in round 6 - gains nothing on the OMD-shaped graph and costs the sweep of the roofline graph 6 %:
  //  every tile kernel's LDS follows the largest tile)
  // HUB landmarks (ba_hub.hip): a STATIC point (no LandmarkMotionTernaryEdge) whose observations do not fit a tile - more than kStaticSlots distinct pose vertices, more than 256
  // per-pose pieces or more than VDO_TILE_INC edges - stays out of the tiles; a workgroup of its own walks its edges.  (A dynamic track beyond the envelope is still refused.)
  std::vector<int32_t> hubs;
  const bool hubs_off = std::getenv("VDO_BA_NO_HUBS") != nullptr;             // (the refusal of rounds 1-5, for the tests of the envelope's messages)
  for (const ChainInfo& ci : chains) {
    // (a static point of at most min(kStaticSlots, 256 threads) observations - nearly every point of every graph - passes all of the checks below by its count alone)
    const bool plain_static = ci.npts == 1 && next_e[ci.head] == -1 && ci.nb == ci.ninc && ci.ninc <= std::min(kStaticSlots, VDO_TILE_THREADS);
    if (!plain_static && !hubs_off && ci.npts == 1 && next_e[ci.head] == -1 && ci.nb == ci.ninc) {
      chain_poses(ci, cposes);
      std::vector<int32_t> u(cposes);
      std::sort(u.begin(), u.end());
      int pieces = 0;
      for (size_t j = 0; j < u.size();) { size_t k = j; while (k < u.size() && u[k] == u[j]) ++k; pieces += (int)((k - j + VDO_TILE_EPT - 1) / VDO_TILE_EPT); j = k; }
      const int distinct = (int)(std::unique(u.begin(), u.end()) - u.begin());
      if (distinct > kStaticSlots || pieces > VDO_TILE_THREADS || ci.ninc > VDO_TILE_INC) { hubs.push_back(ci.head); continue; }
    }
    if (ci.npts > VDO_TILE_PTS || ci.ninc > VDO_TILE_INC) {
      delete ba;
      return set_error(VDO_ERR_UNSUPPORTED, "landmark track with %d points / %d incidences exceeds the tile capacity (%d / %d)",
                       ci.npts, ci.ninc, VDO_TILE_PTS, VDO_TILE_INC);
    }
    chain_poses(ci, cposes);
    if (!plain_static) {   // the track on its own must fit a tile (checked here, before anything is built, with a message that names the track):
        // distinct pose vertices <= kHardSlots (LDS slots), and its EdgeSE3PointXYZ edges cut per pose into pieces of <= VDO_TILE_EPT must fit
        // the 256 threads of the sweep
      std::vector<int32_t> u(cposes);
      std::sort(u.begin(), u.end());
      const int distinct = (int)(std::unique(u.begin(), u.end()) - u.begin());
      const int slot_limit = (ci.npts == 1 && next_e[ci.head] == -1) ? kStaticSlots : kHardSlots;      // (a static point gets here only with VDO_BA_NO_HUBS)
      if (distinct > slot_limit) {
        delete ba;
        return set_error(VDO_ERR_UNSUPPORTED, "landmark track of %d point(s) / %d incidences touches %d distinct pose vertices (limit %d per track)",
                         ci.npts, ci.ninc, distinct, slot_limit);
      }
      chain_eb_poses.clear();
      for (int c = ci.head;;) {
        for (int k = pb_off[c]; k < pb_off[c + 1]; ++k) chain_eb_poses.push_back(g->eb_pose[pb_idx[k]]);
        const int e = next_e[c];
        if (e == -1) break;
        c = g->et_p2[e];
      }
      std::sort(chain_eb_poses.begin(), chain_eb_poses.end());
      int pieces = 0;
      for (size_t j = 0; j < chain_eb_poses.size();) { size_t k = j; while (k < chain_eb_poses.size() && chain_eb_poses[k] == chain_eb_poses[j]) ++k; pieces += (int)((k - j + VDO_TILE_EPT - 1) / VDO_TILE_EPT); j = k; }
      if (pieces > VDO_TILE_THREADS) {
        delete ba;
        return set_error(VDO_ERR_UNSUPPORTED, "landmark track of %d point(s) with %zu EdgeSE3PointXYZ observations needs %d per-pose pieces (limit %d per track)",
                         ci.npts, chain_eb_poses.size(), pieces, VDO_TILE_THREADS);
      }
    }
    int newp = 0;
    for (int32_t p : cposes) if (pose_stamp[p] != cur_tile_id) ++newp;   // upper bound (duplicates inside the chain counted once below)
    // (+ every thread of the sweep takes <= VDO_TILE_EPT edges of ONE pose slot: the sum over the slots of ceil(edges / VDO_TILE_EPT) must fit the 256 threads)
    auto pieces_with_chain = [&](bool commit) {
      int need = cur_need;
      chain_eb_poses.clear();
      for (int c = ci.head;;) {
        for (int k = pb_off[c]; k < pb_off[c + 1]; ++k) chain_eb_poses.push_back(g->eb_pose[pb_idx[k]]);
        const int e = next_e[c];
        if (e == -1) break;
        c = g->et_p2[e];
      }
      for (int32_t p : chain_eb_poses) {
        if (cnt_stamp[p] != cur_tile_id) { cnt_stamp[p] = cur_tile_id; pose_cnt[p] = 0; }
        if (pose_cnt[p] % VDO_TILE_EPT == 0) ++need;
        ++pose_cnt[p];
      }
      if (commit) cur_need = need;
      else for (int32_t p : chain_eb_poses) --pose_cnt[p];
      return need;
    };
    // slots a tile is closed at: kSoftSlots - but dynamic tracks are packed together up to the slot count the graph's largest track forces on every tile kernel's LDS anyway
    // (dyn_slot_cap: the tile kernels' LDS is sized by the largest tile, so packing up to it costs no occupancy; a tile that holds a static point keeps the soft limit)
    const bool dyn_chain = ci.npts > 1;
    const int slot_cap = (!mixed_order && dyn_chain && cur_dyn_only) ? std::max(kSoftSlots, dyn_slot_cap) : kSoftSlots;
    if (cur_npts > 0 && (cur_npts + ci.npts > VDO_TILE_PTS || cur_ninc + ci.ninc > soft_inc ||
                         (int)cur_poses.size() + newp > slot_cap || pieces_with_chain(false) > VDO_TILE_THREADS))
      close_tile();
    if (cur_npts == 0) cur_dyn_only = true;
    cur_dyn_only = cur_dyn_only && dyn_chain;
    if (cur_npts == 0) {
      cur = Tile{};
      cur.pt_begin = (int32_t)pt_old_of_new.size();
      cur.chain_begin = (int32_t)chain_off.size() - 1;
    }
    for (int32_t p : cposes)
      if (pose_stamp[p] != cur_tile_id) { pose_stamp[p] = cur_tile_id; cur_poses.push_back(p); }
    if ((int)cur_poses.size() > kHardSlots) {
      delete ba;
      return set_error(VDO_ERR_UNSUPPORTED, "landmark track touches %zu pose vertices (limit %d)", cur_poses.size(), kHardSlots);
    }
    for (int c = ci.head;;) {
      pt_new_of_old[c] = (int32_t)pt_old_of_new.size();
      pt_old_of_new.push_back(c);
      pt_prev_edge_new.push_back(prev_e[c]);    // original ternary id for now; remapped below
      for (int k = pb_off[c]; k < pb_off[c + 1]; ++k) tile_eb.push_back(pb_idx[k]);
      const int e = next_e[c];
      if (e == -1) break;
      tile_et.push_back(e);
      c = g->et_p2[e];
    }
    chain_off.push_back((int32_t)pt_old_of_new.size());
    cur_npts += ci.npts; cur_ninc += ci.ninc; cur_nb += ci.nb;
    pieces_with_chain(true);
  }
  close_tile();
  if (thr_overflow) { delete ba; return set_error(VDO_ERR_UNSUPPORTED, "a tile has more pose-slot pieces than threads"); }
  mark_t(2);
  // ---- hub landmarks: device points behind every tile's (each a chain of its own), one pose-major partial row ("slot") per edge behind every tile's slots
  const int NPS_tiles = (int)tile_pose.size();
  std::vector<int32_t> hub_off{0}, hub_point, hub_pose, hub_eb_old;
  for (int32_t c : hubs) {
    pt_new_of_old[c] = (int32_t)pt_old_of_new.size();
    hub_point.push_back((int32_t)pt_old_of_new.size());
    pt_old_of_new.push_back(c);
    pt_prev_edge_new.push_back(-1);
    chain_off.push_back((int32_t)pt_old_of_new.size());
    for (int k = pb_off[c]; k < pb_off[c + 1]; ++k) { const int o = pb_idx[k]; hub_eb_old.push_back(o); hub_pose.push_back(g->eb_pose[o]); tile_pose.push_back(g->eb_pose[o]); }
    hub_off.push_back((int32_t)hub_pose.size());
  }
  const int n_hubs = (int)hubs.size(), n_hub_edges = (int)hub_pose.size();
  if (n_hubs) dense_tiles_ok = false;      // (the dense assembly walks tiles only: graphs with hubs are solved by the PCG)
  if (std::getenv("VDO_BA_TILE_STATS") && place_groups)
    std::fprintf(stderr, "vdo_ba_create: %zu tiles, busiest LDS bank of a 16-lane group-row of EdgeSE3PointXYZ edges: %.3f addresses on average (placement %d)\n",
                 tiles.size(), (double)place_ways / (double)place_groups, place_mode);
  for (auto& pe : pt_prev_edge_new) if (pe >= 0) pe = et_new_of_old[pe];
  const int n_tiles = (int)tiles.size(), NPS = (int)tile_pose.size(), n_chains = (int)chain_off.size() - 1;

  mark_t(3);
  // ---- permuted edge / vertex data
  // (Ebp: entries of the padded edge blocks = the device's edge index space; entries without an edge keep zeros)
  const int Ebp = (int)eb_old_of_new.size();
  std::vector<double> point_new(3 * (size_t)L), eb_z(3 * (size_t)Ebp, 0.0), eb_w(Ebp, 0.0), et_z(3 * (size_t)Et), et_w(Et);
  std::vector<float> eb_zf_host;                   // (function scope: alive until the uploads have been synchronised)
  for (int l = 0; l < L; ++l) for (int k = 0; k < 3; ++k) point_new[3 * (size_t)l + k] = g->point[3 * (size_t)pt_old_of_new[l] + k];
  for (int e = 0; e < Ebp; ++e) {
    const int o = eb_old_of_new[e];
    if (o < 0) continue;
    for (int k = 0; k < 3; ++k) eb_z[(size_t)k * Ebp + e] = g->eb_z[(size_t)k * Eb + o];
    eb_w[e] = g->eb_w[o];
  }
  for (int e = 0; e < Et; ++e) {
    const int o = et_old_of_new[e];
    for (int k = 0; k < 3; ++k) et_z[(size_t)k * Et + e] = g->et_z[(size_t)k * Et + o];
    et_w[e] = g->et_w[o];
  }
  // pose -> slots CSR ; pose -> pose-pose edges CSR
  std::vector<int32_t> ps_off(P + 1, 0), ps_idx(NPS);
  for (int k = 0; k < NPS; ++k) ps_off[tile_pose[k] + 1]++;
  for (int p = 0; p < P; ++p) ps_off[p + 1] += ps_off[p];
  {
    std::vector<int32_t> fill(ps_off.begin(), ps_off.end() - 1);
    for (int k = 0; k < NPS; ++k) ps_idx[fill[tile_pose[k]]++] = k;
  }
  // pose-major rows of the sweep partials (ba_dev.hpp): slot s -> row slot_dst[s]; 16 sums per row unless a pose carries both edge kinds
  std::vector<int32_t> slot_dst((size_t)NPS + 1, 0), pose_kind(std::max(P, 1), 0);
  for (int k = 0; k < NPS; ++k) slot_dst[ps_idx[k]] = k;
  int ps_stride = 16;
  {
    std::vector<char> has_b(P, 0), has_t(P, 0);
    for (int e = 0; e < Eb; ++e) has_b[g->eb_pose[e]] = 1;
    for (int e = 0; e < Et; ++e) has_t[g->et_pose[e]] = 1;
    for (int p = 0; p < P; ++p) { pose_kind[p] = has_t[p] ? 1 : 0; if (has_b[p] && has_t[p]) ps_stride = 32; }
    if (std::getenv("VDO_BA_WIDE_PARTIALS")) ps_stride = 32;
  }
  std::vector<int32_t> pe_off(P + 1, 0), pe_idx(2 * (size_t)Ep);
  for (int e = 0; e < Ep; ++e) { pe_off[g->ep_i[e] + 1]++; pe_off[g->ep_j[e] + 1]++; }
  for (int p = 0; p < P; ++p) pe_off[p + 1] += pe_off[p];
  {
    std::vector<int32_t> fill(pe_off.begin(), pe_off.end() - 1);
    for (int e = 0; e < Ep; ++e) { pe_idx[fill[g->ep_i[e]]++] = (e << 1); pe_idx[fill[g->ep_j[e]]++] = (e << 1) | 1; }
  }
  std::vector<int32_t> pr_off(P + 1, 0), pr_idx(std::max(Npr, 1));
  for (int q = 0; q < Npr; ++q) pr_off[g->pr_pose[q] + 1]++;
  for (int p = 0; p < P; ++p) pr_off[p + 1] += pr_off[p];
  {
    std::vector<int32_t> fill(pr_off.begin(), pr_off.end() - 1);
    for (int q = 0; q < Npr; ++q) pr_idx[fill[g->pr_pose[q]]++] = q;
  }
  mark_t(4);
  // ---- pose chains for the block-tridiagonal preconditioner: connected components of the pose-pose
  // (EdgeSE3) graph that are simple paths - the odometry chain of the cameras, the smoothness chain of
  // every object's motions (src/Optimizer.cc:1590-1612, 1743-1766) - in path order; every other pose
  // (isolated, or part of a branching / cyclic component) is a chain of length 1 (plain block-Jacobi).
  // Round 5: a path of >= kTwistMin poses is stored in TWISTED order - first half p_0 .. p_{m-1}, then the second half BACKWARDS p_{n-1} .. p_{m+1}, then p_m
  // (the joint) - so that its block LDL^T is two independent recurrences of half the depth that meet in one step (k_pchain_factor runs them on two waves).
  // In that order position m (p_{n-1}) has no predecessor (pc_edge = -1: L = 0, the substitutions restart there by themselves) and the joint has two: position
  // n-2 (the ordinary link) and position m-1 - the chain's one FAR link (pc_far_pos / pc_far_edge, -1 for an untwisted chain).  No fill-in: an exact
  // factorisation of the same block-tridiagonal matrix, reordered.
  std::vector<int32_t> pc_off{0}, pc_pose, pc_edge, pc_far_pos, pc_far_edge;
  const int kTwistMin = std::getenv("VDO_BA_NO_TWIST") ? (1 << 30) : 16;
  std::vector<char> comp_ok_all;
  {
    std::vector<int> deg(P, 0);
    for (int e = 0; e < Ep; ++e) { deg[g->ep_i[e]]++; deg[g->ep_j[e]]++; }
    std::vector<int> comp(P, -1);
    std::vector<char> comp_ok;
    std::vector<int> stack;
    int ncomp = 0;
    for (int p0 = 0; p0 < P; ++p0) {
      if (comp[p0] != -1) continue;
      int nodes = 0, degsum = 0; bool ok = true;
      stack.assign(1, p0); comp[p0] = ncomp;
      while (!stack.empty()) {
        const int p = stack.back(); stack.pop_back();
        ++nodes; degsum += deg[p];
        if (deg[p] > 2) ok = false;
        for (int k = pe_off[p]; k < pe_off[p + 1]; ++k) {
          const int e = pe_idx[k] >> 1;
          const int q = (pe_idx[k] & 1) ? g->ep_i[e] : g->ep_j[e];
          if (comp[q] == -1) { comp[q] = ncomp; stack.push_back(q); }
        }
      }
      if (degsum / 2 != nodes - 1) ok = false;        // a tree with max degree 2 is a path; anything else has a cycle or a double edge
      comp_ok.push_back(ok ? 1 : 0);
      comp_ok_all.push_back(ok ? 1 : 0);
      ++ncomp;
    }
    std::vector<char> done(P, 0);
    for (int p0 = 0; p0 < P; ++p0) {
      if (done[p0]) continue;
      if (!comp_ok[comp[p0]] || deg[p0] == 0) {
        done[p0] = 1; pc_pose.push_back(p0); pc_edge.push_back(-1); pc_off.push_back((int32_t)pc_pose.size()); pc_far_pos.push_back(-1); pc_far_edge.push_back(-1);
        continue;
      }
      if (deg[p0] != 1) continue;                       // start paths at their lower-numbered end point
      std::vector<int32_t> nodes, via_of;               // the path, and for t >= 1 the link nodes[t-1] -> nodes[t] (edge << 1 | side)
      int prev = -1, cur = p0, via = -1;
      while (cur != -1) {
        done[cur] = 1; nodes.push_back(cur); via_of.push_back(via);
        int nxt = -1, nvia = -1;
        for (int k = pe_off[cur]; k < pe_off[cur + 1]; ++k) {
          const int e = pe_idx[k] >> 1, side = pe_idx[k] & 1;
          const int q = side ? g->ep_i[e] : g->ep_j[e];
          if (q != prev && !done[q]) { nxt = q; nvia = (e << 1) | side; }     // side 0: cur is i of the edge -> E(cur,next) = block(i,j); 1: transposed
        }
        prev = cur; cur = nxt; via = nvia;
      }
      const int n = (int)nodes.size(), base = (int)pc_pose.size();
      if (n < kTwistMin) {
        for (int t = 0; t < n; ++t) { pc_pose.push_back(nodes[t]); pc_edge.push_back(via_of[t]); }
        pc_far_pos.push_back(-1); pc_far_edge.push_back(-1);
      } else {
        const int m = n / 2;
        for (int t = 0; t < m; ++t) { pc_pose.push_back(nodes[t]); pc_edge.push_back(via_of[t]); }
        // second half backwards: position m + u holds nodes[n-1-u]; its predecessor position holds nodes[n-u], the link between them is via_of[n-u] walked the other way
        for (int u = 0; n - 1 - u > m; ++u) { pc_pose.push_back(nodes[n - 1 - u]); pc_edge.push_back(u == 0 ? -1 : (via_of[n - u] ^ 1)); }
        pc_pose.push_back(nodes[m]); pc_edge.push_back(via_of[m + 1] ^ 1);      // the joint: ordinary link from nodes[m+1] (position n-2) ...
        pc_far_pos.push_back(base + m - 1); pc_far_edge.push_back(via_of[m]);    // ... and the far link from nodes[m-1] (position m-1)
      }
      pc_off.push_back((int32_t)pc_pose.size());
    }
    for (int p = 0; p < P; ++p) if (!done[p]) {      // unreachable, defensive
      pc_pose.push_back(p); pc_edge.push_back(-1); pc_off.push_back((int32_t)pc_pose.size()); pc_far_pos.push_back(-1); pc_far_edge.push_back(-1);
    }
  }
  const int n_pchains = (int)pc_off.size() - 1;
  {   // every EdgeSE3 on a simple path? (else the auto solver choice goes to the dense Cholesky, ba_lm.hip)
    bool paths = true;
    for (size_t c = 0; c < comp_ok_all.size(); ++c) paths = paths && comp_ok_all[c];
    ba->pose_graph_is_paths = paths;
    ba->dense_tiles_ok = dense_tiles_ok;
  }
  // incidence index of every (new) edge, for the un-permuting download
  ba->inc_of_eb.resize(Ebp); ba->inc1_of_et.resize(Et); ba->inc2_of_et.resize(Et);
  ba->n_eb = Eb;
  for (const Tile& T : tiles) {
    const int nb = T.eb_end - T.eb_begin, nt = T.et_end - T.et_begin;
    for (int j = 0; j < nb; ++j) ba->inc_of_eb[T.eb_begin + j] = T.inc_begin + j;
    for (int j = 0; j < nt; ++j) { ba->inc1_of_et[T.et_begin + j] = T.inc_begin + nb + j; ba->inc2_of_et[T.et_begin + j] = T.inc_begin + nb + nt + j; }
  }
  ba->pt_old_of_new = pt_old_of_new; ba->pt_new_of_old = pt_new_of_old;
  ba->eb_old_of_new = eb_old_of_new; ba->et_old_of_new = et_old_of_new;

  hipStream_t s = ctx->stream;
  const auto t_built = std::chrono::steady_clock::now();
  BADev& d = ba->d;
  d.P = P; d.L = L; d.Eb = Ebp; d.Et = Et; d.Ep = Ep; d.Npr = Npr; d.Ninc = Ebp + 2 * Et;
  d.n_tiles = n_tiles; d.NPS = NPS; d.n_chains = n_chains; d.max_slots = max_slots;
  if (dense_tile_lds(d) > (size_t)VDO_LDS_MAX_BYTES) ba->dense_tiles_ok = false;      // (a tile of > ~200 pose slots: the dense assembly's workgroup no longer fits the LDS; PCG does - 76 KB at 256 slots)
  d.huber_eb = g->huber_eb; d.huber_et = g->huber_et; d.huber_ep = g->huber_ep;
  d.dsqr_eb = (double)(float)(g->huber_eb * g->huber_eb);   // float member, robust_kernel_impl.h:84
  d.dsqr_et = (double)(float)(g->huber_et * g->huber_et);
  d.dsqr_ep = (double)(float)(g->huber_ep * g->huber_ep);
  UP(pose[0], g->pose, 12 * (size_t)P); UP(pose[1], g->pose, 12 * (size_t)P);
  UP(point[0], point_new.data(), 3 * (size_t)L); UP(point[1], point_new.data(), 3 * (size_t)L);
  tile_pose.push_back(0);                          // (one entry of padding: the tile kernels read slot min(thread, slots - 1) unconditionally, also for a tile without slots)
  UP(tile_pose, tile_pose.data(), tile_pose.size());
  {
    // The device holds the tile descriptors in LAUNCH order (tiles with the longest landmark chain first: their serial solves would be the
    // tail of a launch; they are also the ones with ternary edges, ~1.5x the work in the sweep): workgroup b reads descriptor b straight from
    // its block id - no order array in front of it (the chain of dependent loads at the head of a tile was a quarter of its time).
    std::vector<int32_t> order(std::max(n_tiles, 1), 0), longest(std::max(n_tiles, 1), 0);
    for (int t = 0; t < n_tiles; ++t) { order[t] = t; for (int c = tiles[t].chain_begin; c < tiles[t].chain_end; ++c) longest[t] = std::max(longest[t], chain_off[c + 1] - chain_off[c]); }
    if (!std::getenv("VDO_BA_TILE_ORDER_IDENTITY")) std::stable_sort(order.begin(), order.begin() + n_tiles, [&](int a, int b) { return longest[a] > longest[b]; });
    for (int t = 0; t < n_tiles; ++t) if (longest[t] > 1 || tiles[t].et_end > tiles[t].et_begin) ++ba->d.n_dyn_tiles;
    if (std::getenv("VDO_BA_TILE_ORDER_IDENTITY")) ba->d.n_dyn_tiles = n_tiles;      // (debug order: no dynamic-first guarantee)
    std::vector<Tile> tiles_l(std::max(n_tiles, 1));
    for (int b = 0; b < n_tiles; ++b) tiles_l[b] = tiles[order[b]];
    UP(tiles, tiles_l.data(), tiles_l.size());
  }
  UP(chain_off, chain_off.data(), chain_off.size()); UP(pt_prev_edge, pt_prev_edge_new.data(), L);
  {
    std::vector<uint8_t> single(std::max(L, 1), 0);
    for (int c = 0; c < n_chains; ++c) if (chain_off[c + 1] - chain_off[c] == 1) single[chain_off[c]] = 1;
    UP(pt_single, single.data(), single.size());
  }
  eb_key.resize(std::max<size_t>(eb_key.size(), VDO_TILE_THREADS), -1);     // (>= one row: the tile kernels load a thread's edges unconditionally - entry `thread` of the first block for a tile without edges)
  UP(eb_key, eb_key.data(), eb_key.size());
  UP(et_key, et_key.data(), Et); UP(et_slot, et_slot.data(), Et);
  {
    // compact edge inputs where they are lossless (ba_dev.hpp): one information scalar per edge class, fp32 measurements
    const bool force_general = std::getenv("VDO_BA_GENERAL_EDGES") != nullptr;
    bool wb_uni = Eb > 0 && !force_general, wt_uni = Et > 0 && !force_general, zb_f32 = Eb > 0 && !force_general, zt_zero = Et > 0 && !force_general;
    for (int e = 1; e < Eb && wb_uni; ++e) wb_uni = g->eb_w[e] == g->eb_w[0];
    for (int e = 1; e < Et && wt_uni; ++e) wt_uni = et_w[e] == et_w[0];
    for (size_t i = 0; i < 3 * (size_t)Ebp && zb_f32; ++i) zb_f32 = eb_z[i] == (double)(float)eb_z[i];
    for (size_t i = 0; i < 3 * (size_t)Et && zt_zero; ++i) zt_zero = et_z[i] == 0.0;
    if (wb_uni) d.eb_w_uni = g->eb_w[0]; else UP(eb_w, eb_w.data(), Ebp);
    if (wt_uni) d.et_w_uni = et_w[0]; else UP(et_w, et_w.data(), Et);
    if (zb_f32) { eb_zf_host.assign(eb_z.begin(), eb_z.end()); UP(eb_zf, eb_zf_host.data(), eb_zf_host.size()); }
    else UP(eb_z, eb_z.data(), 3 * (size_t)Ebp);
    if (!zt_zero) UP(et_z, et_z.data(), 3 * (size_t)Et);
    ba->compact_edges = (wb_uni ? 1 : 0) | (zb_f32 ? 2 : 0) | (wt_uni ? 4 : 0) | (zt_zero ? 8 : 0);
  }
  UP(inc_key, inc_key.data(), inc_key.size());
  UP(ep_i, g->ep_i, Ep); UP(ep_j, g->ep_j, Ep); UP(ep_z, g->ep_z, 12 * (size_t)Ep); UP(ep_info, g->ep_info, 36 * (size_t)Ep);
  UP(pr_pose, g->pr_pose, Npr); UP(pr_z, g->pr_z, 12 * (size_t)Npr); UP(pr_info, g->pr_info, 36 * (size_t)Npr);
  UP(ps_off, ps_off.data(), P + 1); UP(ps_idx, ps_idx.data(), NPS);
  UP(slot_dst, slot_dst.data(), slot_dst.size()); UP(pose_kind, pose_kind.data(), std::max(P, 1));
  d.n_hubs = n_hubs; d.n_hub_edges = n_hub_edges;
  std::vector<int32_t> hub_row(std::max(n_hub_edges, 1), 0);
  std::vector<double> hub_z(3 * (size_t)std::max(n_hub_edges, 1), 0.0), hub_w(std::max(n_hub_edges, 1), 0.0);
  if (n_hubs) {
    for (int e = 0; e < n_hub_edges; ++e) {
      hub_row[e] = slot_dst[(size_t)NPS_tiles + e];
      const int o = hub_eb_old[e];
      for (int k = 0; k < 3; ++k) hub_z[(size_t)k * n_hub_edges + e] = g->eb_z[(size_t)k * Eb + o];
      hub_w[e] = g->eb_w[o];
    }
    UP(hub_off, hub_off.data(), hub_off.size()); UP(hub_point, hub_point.data(), hub_point.size()); UP(hub_pose, hub_pose.data(), hub_pose.size());
    UP(hub_row, hub_row.data(), (size_t)n_hub_edges); UP(hub_z, hub_z.data(), 3 * (size_t)n_hub_edges); UP(hub_w, hub_w.data(), (size_t)n_hub_edges);
    const double* Zh = nullptr;
    UP(hub_we, Zh, (size_t)n_hub_edges); UP(hub_chi, Zh, 2 * (size_t)n_hubs);
    ba->hub_eb_old = hub_eb_old;
  }
  d.ps_stride = ps_stride;
  UP(pe_off, pe_off.data(), P + 1); UP(pe_idx, pe_idx.data(), pe_idx.size());
  UP(pr_off, pr_off.data(), P + 1); UP(pr_idx, pr_idx.data(), pr_idx.size());
  d.n_pchains = n_pchains;
  {   // LDS strip of a pose chain's workgroup (ba_solve.hip pchain_solve_partitioned): [len][6] doubles, <= 144 KB
    int maxlen = 1;
    for (int c = 0; c < n_pchains; ++c) maxlen = std::max(maxlen, (int)(pc_off[c + 1] - pc_off[c]));
    d.pc_maxlen = maxlen;
    d.pc_lds = 48 * (size_t)maxlen <= (size_t)(144 * 1024) ? 1 : 0;     // (up to 144 of the 160 KB of a CU: launch_pcg_* raise the kernels' dynamic-LDS limit)
    if (std::getenv("VDO_BA_CHAIN_GLOBAL")) d.pc_lds = 0;
    d.pc_closed = std::getenv("VDO_BA_PCHAIN_CLOSED") ? 1 : 0;
    d.pc_nwave = d.pc_lds ? std::min(16, std::max(1, (maxlen + 7) / 8)) : 1;      // segments of >= 8 positions, one wave each
    if (const char* e = std::getenv("VDO_BA_CHAIN_WAVES")) d.pc_nwave = std::min(16, std::max(1, std::atoi(e)));
  }
  UP(pc_off, pc_off.data(), pc_off.size()); UP(pc_pose, pc_pose.data(), P); UP(pc_edge, pc_edge.data(), P);
  UP(pc_far_pos, pc_far_pos.data(), pc_far_pos.size()); UP(pc_far_edge, pc_far_edge.data(), pc_far_edge.size());
  const double* Z = nullptr;
  // ONE block: Hpp | bp | red_chi [4] | block-Jacobi sums msum [21 P] | failure flag | qs [6 P].  Sharded solves send Hpp .. red_chi[1] per linearisation, msum .. qs per trial -
  // and the whole block at once for the first trial of an LM iteration (launch_linearize(defer_exchange) + launch_factor_and_rhs(lin_pending): red_chi[2..3], the scale partial
  // of the last update, ride along; k_update rewrites them before they are read again)
  UP(Hpp, Z, 69 * (size_t)P + 6);
  ba->d.bp = ba->d.Hpp + 36 * (size_t)P; ba->d.red_chi = ba->d.bp + 6 * (size_t)P;
  ba->d.msum = ba->d.red_chi + 4;
  ba->d.qs = ba->d.msum + 21 * (size_t)P + 1;
  UP(Hll, Z, (size_t)L); UP(bl, Z, 3 * (size_t)L);
  UP(Finc, Z, std::max<size_t>((size_t)Ebp, VDO_TILE_THREADS) + (size_t)Et + 1); UP(Oll, Z, 9 * (size_t)Et); UP(Hpp_ep, Z, 36 * (size_t)Ep); UP(ep_blk, Z, 84 * (size_t)std::max(Ep + Npr, 1));
  UP(part_sums, Z, (size_t)ps_stride * (size_t)std::max(NPS, 1));
  UP(part_chi, Z, 2 * (size_t)n_tiles + 2 * (size_t)(Ep + Npr) + 2);
  UP(part_red, Z, 256);
  UP(Dinv, Z, 9 * (size_t)L); UP(Gl, Z, 9 * (size_t)L); UP(Gdiag, Z, 9 * (size_t)L); UP(Goff, Z, 9 * (size_t)L);
  UP(xl, Z, 3 * (size_t)L); UP(dscal, Z, (size_t)std::max(L, 1));
  UP(Minv, Z, 36 * (size_t)P); UP(Lc, Z, 36 * (size_t)P); UP(Lfar, Z, 36 * (size_t)std::max(n_pchains, 1)); UP(Pf, Z, 36 * (size_t)P); UP(Qb, Z, 36 * (size_t)P); UP(Adg, Z, 36 * (size_t)P);
  UP(xp, Z, 6 * (size_t)P); UP(rp, Z, 6 * (size_t)P); UP(zp, Z, 6 * (size_t)P); UP(pp, Z, 6 * (size_t)P); UP(pp2, Z, 6 * (size_t)P);
  UP(qp, Z, 6 * (size_t)P); UP(bs, Z, 6 * (size_t)P);
  UP(part_pq, Z, (size_t)(P + 3) / 4 + 1); UP(part_rz, Z, (size_t)n_pchains + 1);
  UP(part_q, Z, 8 * (size_t)NPS + 8); UP(part_m, Z, 16 * (size_t)NPS + 16); UP(part_m8, Z, 8 * (size_t)NPS + 8);
  UP(scal, Z, S_COUNT);
  const int32_t* ZI = nullptr;
  UP(flags, ZI, 4);
  static const bool one_stream = std::getenv("VDO_BA_ONE_STREAM") != nullptr;
  if (ba->pooled && ctx->ba_hscal) {                     // the pinned block, the side stream and the events of the last pooled handle
    ba->h_scal = ctx->ba_hscal; ba->d_hscal = ctx->ba_hscal_dev;
    ba->ev0 = ctx->ba_ev[0]; ba->ev1 = ctx->ba_ev[1]; ba->ev_fork = ctx->ba_ev[2]; ba->ev_join = ctx->ba_ev[3]; ba->side = ctx->ba_side;
  } else {
    void* dp = nullptr;
    if (hipHostMalloc((void**)&ba->h_scal, S_COUNT * sizeof(double) + 8 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(&dp, ba->h_scal, 0) != hipSuccess || !dp) {
      vdo_ba_destroy(ba);
      return set_error(VDO_ERR_OOM, "hipHostMalloc failed");
    }
    ba->d_hscal = (double*)dp;
    hipEventCreate(&ba->ev0); hipEventCreate(&ba->ev1);
    if (!one_stream && hipStreamCreateWithFlags(&ba->side, hipStreamNonBlocking) == hipSuccess) {
      hipEventCreateWithFlags(&ba->ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&ba->ev_join, hipEventDisableTiming);
    } else ba->side = nullptr;
    if (ba->pooled) {                                    // first pooled handle of this context: they stay with the context from here on
      ctx->ba_hscal = ba->h_scal; ctx->ba_hscal_dev = ba->d_hscal;
      ctx->ba_ev[0] = ba->ev0; ctx->ba_ev[1] = ba->ev1; ctx->ba_ev[2] = ba->ev_fork; ctx->ba_ev[3] = ba->ev_join; ctx->ba_side = ba->side;
    }
  }
  std::memset(ba->h_scal, 0, S_COUNT * sizeof(double) + 8 * sizeof(int32_t));      // (scalars, flags, the read-back ticket: ba_lm.hip fetch)
  ba->ticket = 0;
  ba->h_flags = (int32_t*)(ba->h_scal + S_COUNT);
  if (hipStreamSynchronize(s) != hipSuccess) { vdo_ba_destroy(ba); return set_error(VDO_ERR_NO_DEVICE, "upload failed: %s", hipGetErrorString(hipGetLastError())); }
  if (std::getenv("VDO_BATCH_TRACE"))
    std::fprintf(stderr, "[vdo_ba_create] validated %.2f, chains %.2f, tiles %.2f (of which closing tiles %.2f), hubs %.2f, permuted data %.2f ms (cumulative); tiles built in %.2f ms, uploaded in %.2f ms (%s)\n", t_mark[0], t_mark[1], t_mark[2], t_close_ms, t_mark[3], t_mark[4], std::chrono::duration<double, std::milli>(t_built - t_create0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_built).count(), ba->pooled ? "pooled" : "own allocations");
  *out = ba;
  return VDO_OK;
}

extern "C" int vdo_ba_destroy(vdo_ba* ba) {
  if (!ba) return VDO_OK;
  if (ba->ctx) ctx_bind(ba->ctx);
  if (ba->pooled) {
    // the slab, the pinned block, the side stream and the events go back to the context - once nothing in flight uses them (hipFree below waits by itself)
    if (ba->ctx->stream) hipStreamSynchronize(ba->ctx->stream);
    if (ba->side) hipStreamSynchronize(ba->side);
    ba->ctx->ba_slab_used = 0; ba->ctx->ba_pool_busy = false;
    const bool kept = ba->ctx->ba_hscal == ba->h_scal;       // (false: the handle failed before its pinned block was handed to the context)
    for (void* p : ba->allocs) hipFree(p);
    if (!kept) {
      if (ba->h_scal) hipHostFree(ba->h_scal);
      for (hipEvent_t e : {ba->ev0, ba->ev1, ba->ev_fork, ba->ev_join}) if (e) hipEventDestroy(e);
      if (ba->side) hipStreamDestroy(ba->side);
    }
    delete ba;
    return VDO_OK;
  }
  for (void* p : ba->allocs) hipFree(p);
  if (ba->h_scal) hipHostFree(ba->h_scal);      // (h_flags lives in the same block)
  if (ba->ev0) hipEventDestroy(ba->ev0);
  if (ba->ev1) hipEventDestroy(ba->ev1);
  if (ba->ev_fork) hipEventDestroy(ba->ev_fork);
  if (ba->ev_join) hipEventDestroy(ba->ev_join);
  if (ba->side) hipStreamDestroy(ba->side);
  delete ba;
  return VDO_OK;
}

extern "C" int vdo_ba_set_allreduce(vdo_ba* ba, vdo_allreduce_fn fn, void* user, int shard_rank) {
  if (!ba || shard_rank < 0) return set_error(VDO_ERR_INVALID, "vdo_ba_set_allreduce: bad argument");
  ba->red.fn = fn; ba->red.user = user; ba->red.err = 0;
  ba->d.sharded = fn ? 1 : 0;
  ba->d.shard_rank = fn ? shard_rank : 0;
  return VDO_OK;
}

extern "C" int vdo_ba_linearize(vdo_ba* ba, int repeat, float* ms_sweep) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (repeat < 1) repeat = 1;
  if (ms_sweep) {
    launch_sweep_only(ba->d, s);   // warm-up
    hipEventRecord(ba->ev0, s);
    for (int i = 0; i < repeat; ++i) launch_sweep_only(ba->d, s);
    hipEventRecord(ba->ev1, s);
    hipEventSynchronize(ba->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, ba->ev0, ba->ev1);
    *ms_sweep = ms / repeat;
  }
  for (int i = 0; i < (ms_sweep ? 1 : repeat); ++i) launch_linearize(ba->d, s, ba->red);
  ba->lin_current = true;
  return sync_check(ba, "vdo_ba_linearize");
}

extern "C" int vdo_ba_profile_linearize(vdo_ba* ba, int repeat, float ms[2], int64_t dims[8]) {
  if (!ba || !ms) return set_error(VDO_ERR_INVALID, "vdo_ba_profile_linearize: null argument");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (repeat < 1) repeat = 1;
  const BADev& d = ba->d;
  for (int which = 0; which < 2; ++which) {
    if (which == 0) launch_sweep_only(d, s); else launch_linearize(d, s, ba->red);          // warm-up
    hipEventRecord(ba->ev0, s);
    for (int i = 0; i < repeat; ++i) { if (which == 0) launch_sweep_only(d, s); else launch_linearize(d, s, ba->red); }
    hipEventRecord(ba->ev1, s);
    hipEventSynchronize(ba->ev1);
    float t = 0;
    hipEventElapsedTime(&t, ba->ev0, ba->ev1);
    ms[which] = t / repeat;
  }
  if (dims) {
    dims[0] = d.n_tiles; dims[1] = d.NPS; dims[2] = d.ps_stride; dims[3] = d.max_slots;
    dims[4] = 4 + (d.eb_zf ? 12 : 24) + (d.eb_w ? 8 : 0);
    dims[5] = 8 + (d.et_z ? 24 : 0) + (d.et_w ? 8 : 0);
    dims[6] = d.Eb; dims[7] = d.n_hubs;
  }
  ba->lin_current = true;
  return sync_check(ba, "vdo_ba_profile_linearize");
}

// Mean duration of the Schur mat-vec of one CG iteration (k_schur_tile<0>: part_q = B Hll^-1 B^T p over every tile) - the largest consumer of an LM
// iteration - timed alone with events on the context's stream.  Call after an optimisation (vdo_ba_optimize) of this handle: the landmark factors of its
// last trial and the direction of its last solve are what the launches read.  ms: mean milliseconds per launch.
extern "C" int vdo_ba_profile_schur(vdo_ba* ba, int repeat, float* ms) {
  if (!ba || !ms) return set_error(VDO_ERR_INVALID, "vdo_ba_profile_schur: null argument");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (repeat < 1) repeat = 1;
  const BADev& d = ba->d;
  if (!d.n_tiles) { *ms = 0.f; return VDO_OK; }
  hipMemsetAsync(d.flags, 0, 4 * sizeof(int32_t), s);          // (flags[1] = "PCG converged" turns the queued mat-vecs into no-ops)
  launch_schur_matvec_only(d, s);                              // warm-up
  hipEventRecord(ba->ev0, s);
  for (int i = 0; i < repeat; ++i) launch_schur_matvec_only(d, s);
  hipEventRecord(ba->ev1, s);
  hipEventSynchronize(ba->ev1);
  float t = 0;
  hipEventElapsedTime(&t, ba->ev0, ba->ev1);
  *ms = t / repeat;
  return sync_check(ba, "vdo_ba_profile_schur");
}

extern "C" int vdo_ba_download_system(vdo_ba* ba, vdo_ba_system* out) {
  if (!ba || !out) return set_error(VDO_ERR_INVALID, "null argument");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  const BADev& d = ba->d;
  hipStream_t s = ba->ctx->stream;
  auto D2H = [&](void* dst, const void* src, size_t bytes) { if (dst && bytes) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s); };
  // The pose-landmark blocks are kept factored (one scalar per incidence) and expanded with the CURRENT estimate; after an optimisation or
  // vdo_ba_set_estimates the stored Hpp / bp / Hll / Finc belong to an older estimate and the expansion would mix the two: re-linearise
  // first, so that what comes back is always one self-consistent system at the current estimate.
  if (!ba->lin_current) { launch_linearize(ba->d, s, ba->red); ba->lin_current = true; }
  std::vector<double> hll, bl, oll, binc, hub_binc;
  D2H(out->Hpp, d.Hpp, sizeof(double) * 36 * (size_t)d.P);
  D2H(out->bp, d.bp, sizeof(double) * 6 * (size_t)d.P);
  D2H(out->Hpp_ep, d.Hpp_ep, sizeof(double) * 36 * (size_t)d.Ep);
  if (out->Hll) { hll.resize((size_t)d.L); D2H(hll.data(), d.Hll, sizeof(double) * hll.size()); }      // device: one scalar per point (block = s * I3)
  if (out->bl) { bl.resize(3 * (size_t)d.L); D2H(bl.data(), d.bl, sizeof(double) * bl.size()); }
  if (out->Hll_et) { oll.resize(9 * (size_t)d.Et); D2H(oll.data(), d.Oll, sizeof(double) * oll.size()); }
  if (out->Hpl_eb || out->Hlp1_et || out->Hlp2_et) {
    // the solve path keeps the blocks factored (Finc); materialise the explicit 6x3 blocks on demand
    if (!ba->d.Binc && d.Ninc) {
      if (hipMalloc((void**)&ba->d.Binc, sizeof(double) * 18 * (size_t)d.Ninc) != hipSuccess) return set_error(VDO_ERR_OOM, "hipMalloc(Binc) failed");
      ba->allocs.push_back((void*)ba->d.Binc);
    }
    launch_expand_binc(ba->d, s);
    if (d.n_hub_edges && out->Hpl_eb) {                    // the hub landmarks' edges (ba_hub.hip)
      if (!ba->hub_binc) ba->hub_binc = (double*)ba_device_alloc(ba, sizeof(double) * 18 * (size_t)d.n_hub_edges);      // (kept with the handle, like Binc)
      double* hb = ba->hub_binc;
      if (!hb) return set_error(VDO_ERR_OOM, "hipMalloc(hub blocks) failed");
      launch_hub_expand_binc(ba->d, hb, s);
      hub_binc.resize(18 * (size_t)d.n_hub_edges);
      D2H(hub_binc.data(), hb, sizeof(double) * hub_binc.size());
    }
    binc.resize(18 * (size_t)d.Ninc);
    D2H(binc.data(), ba->d.Binc, sizeof(double) * binc.size());
  }
  D2H(ba->h_scal, d.scal, sizeof(double) * S_COUNT);
  rc = sync_check(ba, "vdo_ba_download_system");
  if (rc != VDO_OK) return rc;
  const size_t N = d.Ninc, Ebp = d.Eb, Eb = (size_t)ba->n_eb, Et = d.Et;      // (Ebp: padded edge entries of the device, Eb: edges of the graph)
  if (out->Hll) for (int l = 0; l < d.L; ++l) { double* o = out->Hll + 9 * (size_t)ba->pt_old_of_new[l]; for (int i = 0; i < 9; ++i) o[i] = (i % 4 == 0) ? hll[(size_t)l] : 0.0; }
  if (out->bl) for (int l = 0; l < d.L; ++l) std::memcpy(out->bl + 3 * (size_t)ba->pt_old_of_new[l], bl.data() + 3 * (size_t)l, 24);
  if (out->Hll_et)
    for (size_t e = 0; e < Et; ++e) for (int i = 0; i < 9; ++i) out->Hll_et[i * Et + ba->et_old_of_new[e]] = oll[9 * e + i];
  if (out->Hpl_eb) {
    for (size_t e = 0; e < Ebp; ++e) { if (ba->eb_old_of_new[e] < 0) continue; for (int i = 0; i < 18; ++i) out->Hpl_eb[i * Eb + ba->eb_old_of_new[e]] = binc[i * N + ba->inc_of_eb[e]]; }
    for (size_t e = 0; e < ba->hub_eb_old.size(); ++e) for (int i = 0; i < 18; ++i) out->Hpl_eb[i * Eb + ba->hub_eb_old[e]] = hub_binc[18 * e + i];      // (hub landmarks, ba_hub.hip)
  }
  for (int rep = 0; rep < 2; ++rep) {
    double* dst = rep == 0 ? out->Hlp1_et : out->Hlp2_et;
    if (!dst) continue;
    const std::vector<int32_t>& inc = rep == 0 ? ba->inc1_of_et : ba->inc2_of_et;
    for (size_t e = 0; e < Et; ++e)
      for (int r = 0; r < 3; ++r)          // dst: 3x6 (point x pose) = transpose of the stored 6x3
        for (int c = 0; c < 6; ++c) dst[(size_t)(r * 6 + c) * Et + ba->et_old_of_new[e]] = binc[(size_t)(c * 3 + r) * N + inc[e]];
  }
  out->chi2 = ba->h_scal[S_CHI2];
  out->robust_chi2 = ba->h_scal[S_RCHI2];
  return VDO_OK;
}

extern "C" int vdo_ba_get_estimates(vdo_ba* ba, double* pose_out, double* point_out) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (pose_out) hipMemcpyAsync(pose_out, ba->d.pose[0], sizeof(double) * 12 * (size_t)ba->d.P, hipMemcpyDeviceToHost, s);
  if (point_out && ba->d.L) {
    ba->h_tmp.resize(3 * (size_t)ba->d.L);
    hipMemcpyAsync(ba->h_tmp.data(), ba->d.point[0], sizeof(double) * 3 * (size_t)ba->d.L, hipMemcpyDeviceToHost, s);
  }
  rc = sync_check(ba, "vdo_ba_get_estimates");
  if (rc != VDO_OK) return rc;
  if (point_out)
    for (int l = 0; l < ba->d.L; ++l) std::memcpy(point_out + 3 * (size_t)ba->pt_old_of_new[l], ba->h_tmp.data() + 3 * (size_t)l, 24);
  return VDO_OK;
}

extern "C" int vdo_ba_set_estimates(vdo_ba* ba, const double* pose, const double* point) {
  if (!ba) return set_error(VDO_ERR_INVALID, "null handle");
  ba->lin_current = false;
  int rc = ctx_bind(ba->ctx);
  if (rc != VDO_OK) return rc;
  hipStream_t s = ba->ctx->stream;
  if (pose) hipMemcpyAsync(ba->d.pose[0], pose, sizeof(double) * 12 * (size_t)ba->d.P, hipMemcpyHostToDevice, s);
  if (point && ba->d.L) {
    ba->h_tmp.resize(3 * (size_t)ba->d.L);
    for (int l = 0; l < ba->d.L; ++l) std::memcpy(ba->h_tmp.data() + 3 * (size_t)l, point + 3 * (size_t)ba->pt_old_of_new[l], 24);
    hipMemcpyAsync(ba->d.point[0], ba->h_tmp.data(), sizeof(double) * 3 * (size_t)ba->d.L, hipMemcpyHostToDevice, s);
  }
  ba->oplus_calls = 0;
  return sync_check(ba, "vdo_ba_set_estimates");
}
