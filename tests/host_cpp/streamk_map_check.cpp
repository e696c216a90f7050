// Exhaustive host walk of the stream-K work partition of speecht_amd/csrc/streamk_map.h, the way the kernel walks it:
// every (tile, k-tile) unit covered exactly once; a tile cut into at most two pieces; a tail piece [kt0, nk) is the FIRST piece
// of its workgroup and its head [0, kt0) the LAST piece of the workgroup before it on the same virtual XCD (the kernel's
// hand-off pair: slot (l + 1) publishes, slot l consumes); no workgroup is both waiting and waited on in a cycle.
// Built and run by tests/test_streamk_map_host.py with g++.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "streamk_map.h"

static int check(int tiles_total, int nk) {
  st::SkPlan p;
  if (!st::sk_make_plan(tiles_total, nk, p)) return 1;
  if (p.upw < p.nk || p.wgs_per_xcd > st::SK_SLOTS_PER_XCD || p.wgs_per_xcd < 1) { printf("bad plan\n"); return 1; }
  std::vector<int> cover((size_t)tiles_total * nk, 0), pieces(tiles_total, 0);
  for (int xcd = 0; xcd < st::SK_XCDS; ++xcd) {
    int pending_tail_tile = -1, pending_kt1 = -1;        // the head piece the previous workgroup ended with
    for (int l = 0; l < p.wgs_per_xcd; ++l) {
      st::SkCursor c;
      bool first = true;
      int head_tile = -1, head_kt1 = -1;
      if (st::sk_begin(p, xcd, l, c)) {
        while (c.u < c.u_end) {
          const st::SkPiece q = st::sk_piece(p, c);
          c.u += q.kt1 - q.kt0;
          if (q.tile < 0 || q.tile >= tiles_total || q.kt0 < 0 || q.kt1 > nk || q.kt0 >= q.kt1) { printf("bad piece\n"); return 1; }
          for (int k = q.kt0; k < q.kt1; ++k) cover[(size_t)q.tile * nk + k]++;
          pieces[q.tile]++;
          if (q.kt0 > 0) {                               // tail piece
            if (!first || q.kt1 != nk) { printf("tail piece not first / not reaching the end\n"); return 1; }
            if (pending_tail_tile != q.tile || pending_kt1 != q.kt0) { printf("tail without its head in the previous workgroup\n"); return 1; }
            pending_tail_tile = -1;
          } else if (q.kt1 < nk) {                        // head piece
            if (c.u < c.u_end) { printf("head piece not last\n"); return 1; }
            head_tile = q.tile; head_kt1 = q.kt1;
          }
          first = false;
        }
      }
      if (pending_tail_tile >= 0) { printf("head piece of tile %d never completed\n", pending_tail_tile); return 1; }
      pending_tail_tile = head_tile; pending_kt1 = head_kt1;
    }
    if (pending_tail_tile >= 0) { printf("last workgroup ends with a head piece\n"); return 1; }
  }
  for (size_t i = 0; i < cover.size(); ++i) if (cover[i] != 1) { printf("unit %zu covered %d times (tiles %d nk %d)\n", i, cover[i], tiles_total, nk); return 1; }
  for (int t = 0; t < tiles_total; ++t) if (pieces[t] < 1 || pieces[t] > 2) { printf("tile %d in %d pieces\n", t, pieces[t]); return 1; }
  return 0;
}

int main() {
  int n = 0;
  const int nks[] = {1, 2, 3, 8, 12, 16, 17, 64, 128};
  for (int nk : nks)
    for (int tiles = 1; tiles <= 1300; ++tiles) {
      if (check(tiles, nk)) { printf("FAILED tiles=%d nk=%d\n", tiles, nk); return 1; }
      ++n;
    }
  for (int tiles : {3072, 4096, 6144, 9999}) for (int nk : {16, 128}) { if (check(tiles, nk)) return 1; ++n; }
  st::SkPlan p;
  st::sk_make_plan(576, 16, p);
  printf("checked %d plans; 576 tiles x 16: wgs_per_xcd=%d upw=%d\n", n, p.wgs_per_xcd, p.upw);
  return 0;
}
