// Hybrid stream-K work plan of the GEMM kernel (gemm_h.hip: pre-split H-format operands).  W persistent workgroups; output tiles that fill whole rounds of W run data-parallel with the epilogue
// fused; the R = tiles mod W remaining tiles are cut along K, written as raw partial tiles and summed in a FIXED order.
#pragma once
#include "ofb_common.h"

namespace ofb_plan {

struct Plan { int bm, bn, mt, nt, ntiles, I, W, full_rounds, R, q, S, qs, stagger; };

__host__ __device__ inline Plan make_plan(int M, int N, int K, int W, int bm, int bn, int bk = 16) {
  Plan p;
  p.bm = bm; p.bn = bn; p.stagger = 0;
  p.mt = (M + bm - 1) / bm;
  p.nt = (N + bn - 1) / bn;
  p.ntiles = p.mt * p.nt;
  p.I = (K + bk - 1) / bk;
  p.W = W;
  p.full_rounds = p.ntiles / W;
  p.R = p.ntiles - p.full_rounds * W;
  // After >= 3 full rounds a remainder that fills at least half a round runs as one more (partly idle) data-parallel round:
  // the idle share (< 1/8 of the launch) costs less than the partial-tile traffic and the fix-up launch of a streamed tail.
  if (p.full_rounds >= 3 && 2 * p.R >= W) { p.full_rounds += 1; p.R = 0; }
  p.q = p.R ? (int)(((long long)p.R * p.I + W - 1) / W) : 0;      // K-iterations per workgroup in the streamed tail (q <= I)
  // Split-major tail: when the W workgroups divide (almost) evenly over the R tail tiles, cut every tile's K range into the
  // same S pieces and give workgroup v piece v / R of tile v % R.  Workgroups that run side by side on one XCD then walk the
  // SAME K rows of different tiles and share them in its L2 (weight gradients: every token row of dY / X is needed by all
  // tiles), where the flattened runs above place neighbours on different K ranges of one tile and nothing is shared.
  p.S = 0; p.qs = 0;
  if (p.R > 0) {
    const int S = W / p.R;
    if (S >= 2 && (W - S * p.R) * 10 <= W) { p.S = S; p.qs = (p.I + S - 1) / S; }
  }
  return p;
}

__host__ __device__ __forceinline__ void tile_coord(const Plan& p, int tile, int& m0, int& n0) {
  m0 = (tile / p.nt) * p.bm; n0 = (tile % p.nt) * p.bn;
}

// One unit of work: K-iterations [it0, it1) of output tile `tile`; slot < 0 -> full tile, fused epilogue to C;
// slot >= 0 -> raw partial tile to workspace[slot].
struct Seg { int m0, n0, it0, it1, slot; bool ok; };

// raw: the workgroup's own blockIdx.x (v is its XCD-contiguous remap).  Full rounds hand tile v + idx W to workgroup v: neighbours on
// an XCD share operand panels in its L2.  A PARTIAL last round of a multi-round launch (plan.stagger bit 3) goes by the raw index
// instead: its tiles then fall round-robin over ALL XCDs - the contiguous form piled them on the first XCDs, whose CUs ran one round
// more on both of their workgroups while the others idled - and on the first-dispatched workgroup of every CU, the one that runs
// its units ~1.4x faster than its later-dispatched neighbour
template <bool TAIL>
__device__ __forceinline__ Seg get_seg(const Plan p, int v, int idx, int raw = -1) {
  Seg s;
  s.m0 = s.n0 = s.it0 = s.it1 = 0; s.slot = -1; s.ok = false;
  if (!TAIL) {
    if (idx >= p.full_rounds) return s;
    const bool spread = (p.stagger & 8) != 0 && idx + 1 == p.full_rounds && raw >= 0;
    int tile = (spread ? raw : v) + idx * p.W;
    if ((p.stagger & 128) && raw >= 0) {
      // a launch of ONE partial round: the contiguous form fills the first XCDs with two workgroups per CU and leaves the last ones
      // empty (394 tiles on 512 slots: XCDs 0-5 full, XCD 7 idle); here every XCD gets an equal, still contiguous share, handed to
      // its workgroups in dispatch order - the first-dispatched workgroup of EVERY CU of the chip gets a tile before any second one
      if (raw >= p.ntiles) return s;
      tile = ofb_xcd_remap(raw, p.ntiles);
    }
    if (tile >= p.ntiles) return s;
    tile_coord(p, tile, s.m0, s.n0); s.it0 = 0; s.it1 = p.I; s.ok = true;
    return s;
  }
  const int part = idx;
  if (part > 1 || p.R == 0) return s;
  if (p.S > 0) {                          // split-major: one piece per workgroup, slot = v
    if (part != 0) return s;
    const int sp = v / p.R, tl = v - sp * p.R;
    const int i0 = sp * p.qs, i1 = min(i0 + p.qs, p.I);
    if (sp >= p.S || i0 >= i1) return s;
    const int tile = p.full_rounds * p.W + tl;
    tile_coord(p, tile, s.m0, s.n0); s.it0 = i0; s.it1 = i1; s.slot = v; s.ok = true;
    return s;
  }
  // R * I < W * I <= 768 * (K/16): fits 32 bits for every K the host accepts (checked in ofb_gemm_f32)
  const int beg = v * p.q, tot = p.R * p.I;
  const int end = min(beg + p.q, tot);
  if (beg >= end) return s;
  const int a = beg / p.I, it0 = beg - a * p.I;
  const int n0 = end - beg;
  const int first = min(p.I - it0, n0);
  int tl, i0, i1;
  if (part == 0) { tl = a; i0 = it0; i1 = it0 + first; }
  else {
    if (n0 - first <= 0) return s;
    tl = a + 1; i0 = 0; i1 = n0 - first;
  }
  const int tile = p.full_rounds * p.W + tl;
  tile_coord(p, tile, s.m0, s.n0); s.it0 = i0; s.it1 = i1; s.slot = 2 * v + part; s.ok = true;
  return s;
}

}  // namespace ofb_plan
