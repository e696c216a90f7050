// Work partition of the persistent ("stream-K") per-bin product kernels (conv_gemm.hip): the (bin, tile, k-tile) unit list
// is dealt in equal contiguous runs to a fixed number of workgroups, so that a launch whose tile count is not a multiple of the
// chip's workgroup slots (36 bins x 16 tiles = 576 tiles on 512 slots: three rounds for 2.25 rounds of work) costs every CU
// the same.  Plain C++ on purpose: the host builds the plan, the kernel walks it, and tests/host_cpp/streamk_map_check.cpp
// walks it exhaustively on the CPU (every unit covered exactly once, at most two pieces per tile, the hand-off pairs consistent).
//
// Units of one "virtual XCD" x (a label, = blockIdx.x % 8: the chip deals blocks to its 8 XCDs round-robin, so the label is the
// XCD in practice -- used for L2 locality only, never for correctness):
//   tiles [x * tiles_per_xcd, (x + 1) * tiles_per_xcd) of the global tile list (bin-major, then column panel, row tile fastest),
//   each `nk` k-tiles long; workgroup l of that XCD owns units [l * upw, (l + 1) * upw).
// A tile's pieces therefore belong to CONSECUTIVE workgroups l, l + 1, ...: the first (it holds k-tile 0, the HEAD) is the last
// piece of workgroup l's run, every other one is the FIRST piece of its workgroup's run (and, unless it reaches the tile's end,
// that workgroup's whole run).  Pieces that do not hold k-tile 0 publish their partial tile; the head's owner adds them in
// workgroup order (head + next + next ..., a fixed order) and writes the tile.
#pragma once

#ifdef __HIPCC__
#define ST_SK_HD __host__ __device__ __forceinline__
#else
#define ST_SK_HD inline
#endif

namespace st {

constexpr int SK_XCDS = 8;
constexpr int SK_MAX_SLOTS_PER_XCD = 96;             // up to three resident workgroups on each of an XCD's 32 CUs

struct SkPlan {
  int tiles_total;                                   // bins * tiles per bin
  int nk;                                            // k-tiles per tile
  int tiles_per_xcd;                                 // ceil(tiles_total / 8)
  int wgs_per_xcd;                                   // grid = 8 * wgs_per_xcd
  int upw;                                           // units per workgroup
};

struct SkCursor { int u, u_end, tile_lo; };
struct SkPiece { int tile, kt0, kt1; };

// slots_per_xcd: workgroups per XCD the launch should use (64 = two per CU, 96 = three; fewer only when there are fewer
// units).  false: nothing to launch this way (no tiles)
inline bool sk_make_plan(int tiles_total, int nk, int slots_per_xcd, SkPlan& p) {
  if (tiles_total <= 0 || nk <= 0 || slots_per_xcd <= 0 || slots_per_xcd > SK_MAX_SLOTS_PER_XCD) return false;
  p.tiles_total = tiles_total;
  p.nk = nk;
  p.tiles_per_xcd = (tiles_total + SK_XCDS - 1) / SK_XCDS;
  const long units = (long)p.tiles_per_xcd * nk;
  p.wgs_per_xcd = units < slots_per_xcd ? (int)units : slots_per_xcd;
  p.upw = (int)((units + p.wgs_per_xcd - 1) / p.wgs_per_xcd);
  return true;
}

// the run of workgroup (xcd, local); false: empty
ST_SK_HD bool sk_begin(const SkPlan& p, int xcd, int local, SkCursor& c) {
  c.tile_lo = xcd * p.tiles_per_xcd;
  int ntiles = p.tiles_total - c.tile_lo;
  if (ntiles > p.tiles_per_xcd) ntiles = p.tiles_per_xcd;
  c.u = local * p.upw;
  c.u_end = c.u + p.upw;
  const int units = ntiles > 0 ? ntiles * p.nk : 0;
  if (c.u_end > units) c.u_end = units;
  return c.u < c.u_end;
}

// the piece at the cursor: k-tiles [kt0, kt1) of global tile `tile`
ST_SK_HD SkPiece sk_piece(const SkPlan& p, const SkCursor& c) {
  SkPiece q;
  const int tl = c.u / p.nk;
  q.kt0 = c.u - tl * p.nk;
  q.kt1 = q.kt0 + (c.u_end - c.u);
  if (q.kt1 > p.nk) q.kt1 = p.nk;
  q.tile = c.tile_lo + tl;
  return q;
}

}  // namespace st
