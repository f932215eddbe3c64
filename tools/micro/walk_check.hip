// Host-side check of the balanced walk's arithmetic (csrc/spd_ws.hpp): ColWalk::find_fast (closed form: fp32 square root, up to
// four steps either way) against ColWalk::find (binary search) on every block boundary +- 2 and a stride through the line — in
// the plain line and in the augmented one (cross = 1, 8, 16, 100, 1024; the shipped defaults are 16 / 8), for full launches
// and row shards up to n = 2^22, with the stepping required to need at most four steps —, and divmod_small / WalkShares
// against the integer division.  No GPU involved: tests/test_host_cpu.py builds and runs it.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../include/mm_manifolds.h"
#include "../../matrix-manifolds_amd/csrc/spd_ws.hpp"
using namespace mm;
int main() {
  long bad = 0, checked = 0;
  for (int n : {2, 3, 5, 63, 64, 65, 127, 129, 300, 1000, 2274, 4039, 5000, 16384, 100000, 1 << 22}) for (int bw : {64, 128, 192, 256, 512}) for (int shard = 0; shard < 4; ++shard) {
    int rb = 0, re = n;
    if (shard == 1) { rb = n / 3; re = (2 * n) / 3; } else if (shard == 2) { rb = n - n / 8; re = n; } else if (shard == 3) { rb = 0; re = n / 8 + 1; }
    ColWalk w(n, rb, re, bw);
    const int64_t tot = w.total();
    if (tot <= 0) continue;
    // every block boundary +-2, and a stride through the line
    for (int c = w.c0; c <= w.ncb; c += (w.ncb > 4000 ? 97 : 1)) for (int d = -2; d <= 2; ++d) {
      const int64_t p = w.prefix(c) + d;
      if (p < 0 || p >= tot) continue;
      ++checked;
      if (w.find_fast(p) != w.find(p) || !w.find_fast_converges(p)) { if (++bad < 10) printf("n=%d bw=%d rb=%d re=%d p=%lld fast=%d find=%d\n", n, bw, rb, re, (long long)p, w.find_fast(p), w.find(p)); }
    }
    // the augmented line (a share pays `cross` units per block it enters; the shipped defaults are 16 for fp32 and 8 for
    // fp64): find_fast(p, cross) against a bisection on aprefix() at every block boundary +-2 and a stride through the line —
    // and the stepping must need at most four steps either way (find_fast_converges: a drifting guess would cost every prologue)
    for (int cross : {1, 8, 16, 100, 1024}) {
      const int64_t atot = w.total_aug(cross);
      auto exact = [&](int64_t p) { int lo = w.c0, hi = w.ncb - 1; while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (w.aprefix(mid, cross) <= p) lo = mid; else hi = mid - 1; } return lo; };
      auto hold = [&](int64_t p) {
        if (p < 0 || p >= atot) return;
        ++checked;
        int64_t off = -1;
        const int f = w.find_fast_off(p, cross, &off), e = exact(p);
        if (f != e || off != p - w.aprefix(e, cross) || !w.find_fast_converges(p, cross)) { if (++bad < 10) printf("aug n=%d bw=%d rb=%d re=%d cross=%d p=%lld fast=%d exact=%d converged=%d\n", n, bw, rb, re, cross, (long long)p, f, e, int(w.find_fast_converges(p, cross))); }
      };
      for (int c = w.c0; c <= w.ncb; c += (w.ncb > 4000 ? 89 : 1)) for (int d = -2; d <= 2; ++d) { hold(w.aprefix(c, cross) + d); hold(w.aprefix(c, cross) + cross + d); }
      const int64_t astep = atot / 2503 + 1;
      for (int64_t p = 0; p < atot; p += astep) hold(p);
      hold(atot - 1);
    }
    const int64_t step = tot / 5003 + 1;
    for (int64_t p = 0; p < tot; p += step) { ++checked; if (w.find_fast(p) != w.find(p)) { if (++bad < 10) printf("n=%d bw=%d rb=%d re=%d p=%lld fast=%d find=%d\n", n, bw, rb, re, (long long)p, w.find_fast(p), w.find(p)); } }
    // share_begin without the division
    for (int g : {1, 7, 256, 1000, 1024, 1792}) {
      int64_t q, r;
      ColWalk::divmod_small(tot, g, q, r);
      if (q != tot / g || r != tot % g) { ++bad; printf("divmod n=%d g=%d\n", n, g); }
      const WalkShares sh(tot, g);   // the shares tile the line exactly (host arithmetic of the device's WalkShares::of)
      int64_t at = 0;
      for (int k = 0; k < g; ++k) {
        const int64_t begin = sh.q * k + (k < sh.r ? k : sh.r), count = sh.q + (k < sh.r ? 1 : 0);
        if (begin != at || begin != ColWalk::share_begin(tot, k, g)) { ++bad; break; }
        at += count;
      }
      if (at != tot) { ++bad; printf("shares n=%d g=%d\n", n, g); }
      // shares with a block-entry cost: walking every share as the kernels do (enter, one unit per row, `cross` per block
      // entered) visits every (block, row) of the triangle exactly once, in order
      for (int cross : {0, 3, 8, 40}) {
        if (tot > 3000000) continue;   // (the walk below is linear in the units)
        const int64_t atot = w.total_aug(cross);
        const WalkShares sc(atot, g, cross);
        int eb = w.c0, er = rb;      // the (block, row) the next share is expected to continue at
        int64_t seen = 0;
        bool ok = true;
        for (int k = 0; k < g && ok; ++k) {
          const int64_t a = sc.q * k + (k < sc.r ? k : sc.r);
          const int cnt = int(sc.q) + (k < sc.r ? 1 : 0);
          int cb, r, rem;
          w.enter(a, cnt, cross, &cb, &r, &rem);
          while (rem > 0) {
            const int hi = w.hi(cb);
            const int take = (hi - r) < rem ? (hi - r) : rem;
            if (take > 0) {
              // normalise the expectation past exhausted blocks
              while (er >= w.hi(eb) && eb < w.ncb - 1) { ++eb; er = rb; }
              if (cb != eb || r != er) { ok = false; break; }
              er += take; seen += take;
            }
            rem -= take; r += take;
            if (rem > 0) { ++cb; r = rb; rem -= cross; if (cb >= w.ncb) { ok = rem <= 0; break; } }
          }
        }
        if (!ok || seen != tot) { if (++bad < 10) printf("walk n=%d bw=%d rb=%d re=%d g=%d cross=%d seen=%lld tot=%lld\n", n, bw, rb, re, g, cross, (long long)seen, (long long)tot); }
      }
    }
  }
  printf("checked %ld, bad = %ld\n", checked, bad);
  return bad != 0;
}
