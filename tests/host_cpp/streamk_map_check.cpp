// Exhaustive host walk of the stream-K work partition of speecht_amd/csrc/streamk_map.h, the way the kernel walks it:
// every (tile, k-tile) unit covered exactly once; the pieces of a tile belong to consecutive workgroups of one virtual XCD; the
// piece holding k-tile 0 (the head) is the LAST piece of its workgroup, every other piece the FIRST piece of its workgroup and
// starts exactly where the previous piece ended (what the head's owner assumes when it adds the partial tiles of workgroups
// l + 1, l + 2, ... until the tile is whole); a workgroup publishes at most one partial tile.
// Built and run by tests/test_streamk_map_host.py with g++.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "streamk_map.h"

static int check(int tiles_total, int nk, int slots) {
  st::SkPlan p;
  if (!st::sk_make_plan(tiles_total, nk, slots, p)) return 1;
  if (p.upw < 1 || p.wgs_per_xcd > slots || p.wgs_per_xcd < 1) { printf("bad plan\n"); return 1; }
  std::vector<int> cover((size_t)tiles_total * nk, 0);
  for (int xcd = 0; xcd < st::SK_XCDS; ++xcd) {
    int open_tile = -1, open_at = -1;                  // a tile whose head was computed by an earlier workgroup, covered up to open_at
    for (int l = 0; l < p.wgs_per_xcd; ++l) {
      st::SkCursor c;
      bool first = true;
      int published = 0;
      if (st::sk_begin(p, xcd, l, c)) {
        while (c.u < c.u_end) {
          const st::SkPiece q = st::sk_piece(p, c);
          c.u += q.kt1 - q.kt0;
          if (q.tile < 0 || q.tile >= tiles_total || q.kt0 < 0 || q.kt1 > nk || q.kt0 >= q.kt1) { printf("bad piece\n"); return 1; }
          for (int k = q.kt0; k < q.kt1; ++k) cover[(size_t)q.tile * nk + k]++;
          if (q.kt0 > 0) {                               // a published piece
            if (!first) { printf("published piece is not the first of its workgroup\n"); return 1; }
            if (open_tile != q.tile || open_at != q.kt0) { printf("published piece does not continue the open tile\n"); return 1; }
            // the head's owner predicts this piece's length as min(upw, nk - covered): check
            const int predicted = (nk - open_at) < p.upw ? (nk - open_at) : p.upw;
            if (q.kt1 - q.kt0 != predicted) { printf("piece length %d != predicted %d\n", q.kt1 - q.kt0, predicted); return 1; }
            ++published;
            open_at = q.kt1;
            if (open_at == nk) open_tile = -1;
          } else {
            if (open_tile >= 0) { printf("a new tile starts while tile %d is open\n", open_tile); return 1; }
            if (q.kt1 < nk) {                            // head piece: must be the last piece of this workgroup
              if (c.u < c.u_end) { printf("head piece not last\n"); return 1; }
              open_tile = q.tile; open_at = q.kt1;
            }
          }
          first = false;
        }
      } else if (open_tile >= 0) { printf("empty workgroup while tile %d is open\n", open_tile); return 1; }
      if (published > 1) { printf("two published pieces in one workgroup\n"); return 1; }
    }
    if (open_tile >= 0) { printf("XCD ends with tile %d open\n", open_tile); return 1; }
  }
  for (size_t i = 0; i < cover.size(); ++i) if (cover[i] != 1) { printf("unit %zu covered %d times (tiles %d nk %d)\n", i, cover[i], tiles_total, nk); return 1; }
  return 0;
}

int main() {
  int n = 0;
  const int nks[] = {1, 2, 3, 8, 12, 16, 17, 64, 128};
  for (int slots : {64, 96})
    for (int nk : nks)
      for (int tiles = 1; tiles <= 1300; ++tiles) {
        if (check(tiles, nk, slots)) { printf("FAILED tiles=%d nk=%d slots=%d\n", tiles, nk, slots); return 1; }
        ++n;
      }
  for (int tiles : {3072, 4096, 6144, 9999}) for (int nk : {16, 128}) { if (check(tiles, nk, 96)) return 1; ++n; }
  st::SkPlan p, q;
  st::sk_make_plan(576, 16, 64, p);
  st::sk_make_plan(576, 16, 96, q);
  printf("checked %d plans; 576 tiles x 16: wgs_per_xcd=%d upw=%d / wgs_per_xcd=%d upw=%d\n", n, p.wgs_per_xcd, p.upw, q.wgs_per_xcd, q.upw);
  return 0;
}
