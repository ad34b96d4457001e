// ring_model.cc — the append protocol of the binned route's pass B (brainevent_amd/csrc/be_csr_binned.hip: StreamLds,
// stream_append, stream_flush_list, the drain of k_bin_stream) and the directory walk of pass C (k_bin_accumulate), lifted
// into host C++ so that it runs under ThreadSanitizer and under an exhaustive interleaving search in the CPU container.
// TEST INFRASTRUCTURE: nothing in the product links or runs this file (tests/test_ring_model_cpu.py builds and runs it).
//
// What is modelled, statement for statement:
//   * one WORKGROUP = the shared state {tick[], done[], gen[], ovf[], buf[][]} ("LDS") + its regions / directory ("HBM");
//   * one WAVE = one host thread; its LANES advance in lockstep through the stages of stream_append exactly as the
//     kernel's straight-line code does: all tickets -> { all generation reads -> all stores -> all commits -> flush list
//     -> flush } until no lane has a pending entry.  Lanes without an entry aim at dummy counters of their own;
//   * an LDS atomic = std::atomic RMW.  The kernel relies on "the LDS executes a wave's DS instructions in order"; the model
//     asks for LESS: a writer's store is ordered before its commit by release, a completer's block read after the
//     commits by acquire, `done = 0` before `gen = next` by release, a writer's slot test by acquire.  The block words are
//     PLAIN memory, so ThreadSanitizer reports any access the protocol does not order;
//   * a full region: the block goes through an "overflow image" with atomic adds (the kernel's float atomics);
//   * the drain (one partly filled block per bin at most, directory = min(tickets, room) | overflow flag);
//   * pass C: blocks per region from the directory, prefix sums, the last block's valid count.
// Checked at the end: every output equals the serial sum (each entry carries a unique weight, so a duplicate, a loss or a
// misplaced entry changes a sum), and the conservation counters of be_binned_workspace_status:
// entries == tickets == accumulated + overflowed.
//
// Modes:   ring_model stress <seeds>     threads + random yields, for ThreadSanitizer (build with -fsanitize=thread)
//          ring_model explore            exhaustive search over all interleavings of two waves' DS instructions, small case
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <thread>
#include <unordered_set>
#include <vector>

namespace {

struct Geo {
  int n_bins, cb, ring, cap_blocks, lanes, ne, list_cap;   // list_cap: flush-list length (kernel: 64)
};

struct Entry { uint32_t col; uint64_t w; };               // col = global column (bin * width + local), w = unique weight

// ------------------------------------------------------------------------------------------------ threaded model
struct Shared {
  Geo g;
  int width;
  std::vector<std::atomic<uint32_t>> tick, done, gen, ovf;
  std::vector<uint32_t> buf_col;                          // [ring * n_bins][cb]   plain memory: TSAN watches it
  std::vector<uint64_t> buf_w;
  std::vector<uint32_t> reg_col;                          // regions[bin][block][slot] of this workgroup
  std::vector<uint64_t> reg_w;
  std::vector<uint32_t> dir;                              // per bin
  std::vector<std::atomic<uint64_t>> ovf_img;             // [n_bins * width]
  std::atomic<uint64_t> n_tickets{0}, n_overflow{0}, n_expected{0};
  Shared(const Geo& g_, int width_, int n_waves)
      : g(g_), width(width_), tick(g.n_bins + n_waves * g.lanes), done(g.ring * g.n_bins + n_waves * g.lanes), gen(g.ring * g.n_bins),
        ovf(g.n_bins), buf_col((size_t)g.ring * g.n_bins * g.cb), buf_w(buf_col.size()),
        reg_col((size_t)g.n_bins * g.cap_blocks * g.cb, 0xdeadbeefu), reg_w(reg_col.size(), 0xdeadbeefdeadbeefull), dir(g.n_bins),
        ovf_img((size_t)g.n_bins * width) {
    for (auto& a : tick) a.store(0);
    for (auto& a : done) a.store(0);
    for (auto& a : gen) a.store(0);
    for (auto& a : ovf) a.store(0);
    for (auto& a : ovf_img) a.store(0);
  }
};

struct Yielder {
  std::mt19937 rng;
  int rate;
  void maybe() { if (rate && (int)(rng() % 100) < rate) std::this_thread::yield(); }
};

// stream_flush_list: the wave copies the listed completed blocks to the regions and frees their ring slots
void flush_list(Shared& S, const std::vector<uint32_t>& slot, const std::vector<uint32_t>& q, Yielder& y) {
  const Geo& g = S.g;
  const size_t n = slot.size();
  std::vector<std::vector<uint32_t>> vc(n);
  std::vector<std::vector<uint64_t>> vw(n);
  for (size_t j = 0; j < n; ++j) {                        // (the ds_read_b128 of the blocks)
    vc[j].assign(S.buf_col.begin() + (size_t)slot[j] * g.cb, S.buf_col.begin() + (size_t)(slot[j] + 1) * g.cb);
    vw[j].assign(S.buf_w.begin() + (size_t)slot[j] * g.cb, S.buf_w.begin() + (size_t)(slot[j] + 1) * g.cb);
  }
  y.maybe();
  for (size_t j = 0; j < n; ++j)                          // (rare) a full region: the block goes out through atomics
    if (q[j] >= (uint32_t)g.cap_blocks) {
      for (int s = 0; s < g.cb; ++s) S.ovf_img[vc[j][s]].fetch_add(vw[j][s], std::memory_order_relaxed);
      S.ovf[slot[j] / g.ring].store(1u, std::memory_order_relaxed);
    }
  for (size_t j = 0; j < n; ++j) S.done[slot[j]].store(0u, std::memory_order_relaxed);
  y.maybe();
  for (size_t j = 0; j < n; ++j) S.gen[slot[j]].store(q[j] / g.ring + 1u, std::memory_order_release);
  y.maybe();
  for (size_t j = 0; j < n; ++j)
    if (q[j] < (uint32_t)g.cap_blocks) {
      const size_t at = ((size_t)(slot[j] / g.ring) * g.cap_blocks + q[j]) * g.cb;
      for (int s = 0; s < g.cb; ++s) { S.reg_col[at + s] = vc[j][s]; S.reg_w[at + s] = vw[j][s]; }
    }
}

// stream_append for one wave: e[lane * ne + u]; col == 0xffffffff: no entry
void append(Shared& S, int wave, const std::vector<Entry>& e, Yielder& y) {
  const Geo& g = S.g;
  const int L = g.lanes, NE = g.ne, N = L * NE;
  std::vector<uint32_t> bin(N), t(N), lc(N);
  std::vector<char> pend(N);
  for (int i = 0; i < N; ++i) {
    const int lane = i / NE;
    const uint32_t dummy_bin = (uint32_t)g.n_bins + (uint32_t)(wave * L + lane);
    const uint32_t b = e[i].col == 0xffffffffu ? 0xffffffffu : e[i].col / (uint32_t)S.width;
    bin[i] = b < (uint32_t)g.n_bins ? b : dummy_bin;
    pend[i] = b < (uint32_t)g.n_bins;
    lc[i] = e[i].col;
  }
  for (int i = 0; i < N; ++i) { t[i] = S.tick[bin[i]].fetch_add(1u, std::memory_order_relaxed); if ((i & 3) == 3) y.maybe(); }
  for (;;) {
    std::vector<uint32_t> slotid(N), gv(N), d(N);
    std::vector<char> ok(N);
    for (int i = 0; i < N; ++i) {
      slotid[i] = pend[i] ? bin[i] * g.ring + ((t[i] / g.cb) % g.ring) : 0u;
      gv[i] = pend[i] ? S.gen[slotid[i]].load(std::memory_order_acquire) : 0u;
    }
    y.maybe();
    for (int i = 0; i < N; ++i) {
      ok[i] = pend[i] && gv[i] == t[i] / (uint32_t)(g.cb * g.ring);
      if (ok[i]) {
        const size_t at = (size_t)slotid[i] * g.cb + t[i] % g.cb;
        S.buf_col[at] = lc[i];
        S.buf_w[at] = e[i].w;
      }
    }
    y.maybe();
    std::vector<uint32_t> fl_slot, fl_q;
    for (int i = 0; i < N; ++i) {
      const int lane = i / NE;
      if (ok[i]) d[i] = S.done[slotid[i]].fetch_add(1u, std::memory_order_acq_rel);
      else d[i] = S.done[g.ring * g.n_bins + wave * L + lane].fetch_add(1u, std::memory_order_relaxed);
      if (ok[i] && d[i] == (uint32_t)g.cb - 1u) { fl_slot.push_back(slotid[i]); fl_q.push_back(t[i] / g.cb); }
      if (ok[i]) pend[i] = 0;
    }
    y.maybe();
    for (size_t j0 = 0; j0 < fl_slot.size(); j0 += g.list_cap) {     // the list holds list_cap blocks per pass
      const size_t j1 = std::min(fl_slot.size(), j0 + (size_t)g.list_cap);
      flush_list(S, std::vector<uint32_t>(fl_slot.begin() + j0, fl_slot.begin() + j1),
                 std::vector<uint32_t>(fl_q.begin() + j0, fl_q.begin() + j1), y);
    }
    bool any = false;
    for (int i = 0; i < N; ++i) any |= pend[i];
    if (!any) break;
    std::this_thread::yield();
  }
}

// the drain of k_bin_stream (after the workgroup barrier = after the threads joined)
void drain(Shared& S) {
  const Geo& g = S.g;
  for (int b = 0; b < g.n_bins; ++b) {
    const uint32_t T = S.tick[b].load(), blk = T / g.cb, dd = T % g.cb;
    const size_t slot = (size_t)b * g.ring + blk % g.ring;
    uint32_t o = S.ovf[b].load();
    if (dd > 0 && blk < (uint32_t)g.cap_blocks) {
      const size_t at = ((size_t)b * g.cap_blocks + blk) * g.cb;
      for (int s = 0; s < g.cb; ++s) { S.reg_col[at + s] = S.buf_col[slot * g.cb + s]; S.reg_w[at + s] = S.buf_w[slot * g.cb + s]; }
    }
    if (dd > 0 && blk >= (uint32_t)g.cap_blocks) {
      for (uint32_t s = 0; s < dd; ++s) S.ovf_img[S.buf_col[slot * g.cb + s]].fetch_add(S.buf_w[slot * g.cb + s]);
      o = 1u;
    }
    const uint64_t room = (uint64_t)g.cap_blocks * g.cb;
    S.n_tickets += T;
    S.n_overflow += T > room ? T - room : 0;
    S.dir[b] = (uint32_t)(T < room ? T : room) | (o ? 0x80000000u : 0u);
  }
}

// pass C for one workgroup's regions
bool accumulate_and_check(Shared& S, const std::vector<uint64_t>& expect, uint64_t n_entries, bool verbose) {
  const Geo& g = S.g;
  std::vector<uint64_t> out((size_t)g.n_bins * S.width, 0);
  uint64_t n_added = 0;
  for (int b = 0; b < g.n_bins; ++b) {
    const uint32_t raw = S.dir[b], cnt = raw & 0x7fffffffu, nb = (cnt + g.cb - 1) / g.cb;
    for (uint32_t blk = 0; blk < nb; ++blk)
      for (int s = 0; s < g.cb; ++s) {
        const uint32_t first = blk * g.cb + s;
        if (first >= cnt) continue;
        const size_t at = ((size_t)b * g.cap_blocks + blk) * g.cb + s;
        out[S.reg_col[at]] += S.reg_w[at];
        ++n_added;
      }
    if (raw >> 31)
      for (int i = 0; i < S.width; ++i) { out[(size_t)b * S.width + i] += S.ovf_img[(size_t)b * S.width + i].load(); }
  }
  bool good = out == expect && S.n_tickets == n_entries && n_added + S.n_overflow == n_entries;
  if (!good && verbose)
    fprintf(stderr, "MISMATCH: entries %llu tickets %llu accumulated %llu overflow %llu\n", (unsigned long long)n_entries,
            (unsigned long long)S.n_tickets.load(), (unsigned long long)n_added, (unsigned long long)S.n_overflow.load());
  return good;
}

int stress(int seeds) {
  int bad = 0;
  for (int seed = 0; seed < seeds; ++seed) {
    std::mt19937 rng(seed * 7919 + 13);
    Geo g;
    g.n_bins = 1 + rng() % 5;
    g.cb = 1 << (1 + rng() % 3);                     // 2, 4, 8
    g.ring = 2;
    g.lanes = 4 + rng() % 5;
    g.ne = 2 + 2 * (rng() % 2);
    g.list_cap = 1 + rng() % 4;
    const int n_waves = 2 + rng() % 5, appends = 6 + rng() % 20, width = 7;
    const uint64_t per_bin = (uint64_t)n_waves * appends * g.lanes * g.ne / g.n_bins;
    g.cap_blocks = (rng() % 3 == 0) ? (int)std::max<uint64_t>(1, per_bin / g.cb / 2) : (int)(per_bin * 3 / g.cb + 4);   // sometimes too small: overflow path
    Shared S(g, width, n_waves);
    std::vector<std::vector<std::vector<Entry>>> work(n_waves);
    std::vector<uint64_t> expect((size_t)g.n_bins * width, 0);
    uint64_t uid = 1, n_entries = 0;
    for (int w = 0; w < n_waves; ++w)
      for (int a = 0; a < appends; ++a) {
        std::vector<Entry> e(g.lanes * g.ne);
        const bool skew = rng() % 4 == 0;                     // every lane on one bin: the same-counter case
        for (auto& x : e) {
          if (rng() % 7 == 0) { x.col = 0xffffffffu; x.w = 0; continue; }
          const uint32_t b = skew ? 0u : rng() % g.n_bins;
          x.col = b * width + rng() % width;
          x.w = (uid++) * 1000003ull;
          expect[x.col] += x.w;
          ++n_entries;
        }
        work[w].push_back(e);
      }
    std::vector<std::thread> th;
    for (int w = 0; w < n_waves; ++w)
      th.emplace_back([&, w] {
        Yielder y{std::mt19937((unsigned)(seed * 131 + w)), 30};
        for (auto& e : work[w]) append(S, w, e, y);
      });
    for (auto& t : th) t.join();
    drain(S);
    if (!accumulate_and_check(S, expect, n_entries, true)) { fprintf(stderr, "seed %d FAILED\n", seed); ++bad; }
  }
  printf("stress: %d seeds, %d bad\n", seeds, bad);
  return bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ exhaustive model
// Two waves, every DS instruction of a wave (tickets / generation reads / stores / commits / block reads / done stores / gen
// stores / region stores) one atomic step, all interleavings of the two waves' steps by depth-first search with a visited set.
// (Finer than the hardware: the kernel's lanes issue each of these as ONE instruction per wave, which the LDS executes whole.)
struct XGeo { static constexpr int BINS = 2, CB = 2, RING = 2, LANES = 2, NE = 2, N = LANES * NE, APPENDS = 4, WAVES = 2, CAPB = 16; };
struct XWave {
  uint8_t pc = 0, app = 0;                  // stage within the append, append index
  uint8_t pend = 0;                         // bit per entry
  uint8_t t[XGeo::N] = {0}, gv[XGeo::N] = {0}, ok = 0;
  uint8_t nfl = 0, fl_slot[XGeo::N] = {0}, fl_q[XGeo::N] = {0};
  uint16_t vblk[XGeo::N][XGeo::CB] = {{0}};
  bool done = false;
};
struct XState {
  uint8_t tick[XGeo::BINS] = {0}, dn[XGeo::RING * XGeo::BINS] = {0}, gen[XGeo::RING * XGeo::BINS] = {0};
  uint16_t buf[XGeo::RING * XGeo::BINS][XGeo::CB] = {{0}};
  uint16_t reg[XGeo::BINS][XGeo::CAPB][XGeo::CB] = {{{0}}};
  XWave w[XGeo::WAVES];
};
static uint16_t x_entry_id(int wave, int app, int i) { return (uint16_t)(1 + wave * 64 + app * 8 + i); }
static int x_entry_bin(int wave, int app, int i) { return (wave + app + i / 2) % XGeo::BINS; }    // fixed pattern: both waves hit both bins

struct XExplorer {
  std::unordered_set<std::string> seen;
  uint64_t finals = 0, states = 0;
  bool failed = false;

  static bool step(XState& s, int wi) {                      // one DS instruction of wave wi; false: the wave has finished
    using G = XGeo;
    XWave& w = s.w[wi];
    if (w.done) return false;
    switch (w.pc) {
      case 0:                                                // tickets
        for (int i = 0; i < G::N; ++i) w.t[i] = s.tick[x_entry_bin(wi, w.app, i)]++;
        w.pend = (1 << G::N) - 1;
        w.pc = 1;
        return true;
      case 1:                                                // generation reads
        for (int i = 0; i < G::N; ++i) {
          const int slot = x_entry_bin(wi, w.app, i) * G::RING + (w.t[i] / G::CB) % G::RING;
          w.gv[i] = s.gen[slot];
        }
        w.pc = 2;
        return true;
      case 2:                                                // stores of the entries whose slot is free
        w.ok = 0;
        for (int i = 0; i < G::N; ++i) {
          const int slot = x_entry_bin(wi, w.app, i) * G::RING + (w.t[i] / G::CB) % G::RING;
          if (((w.pend >> i) & 1) && w.gv[i] == w.t[i] / (G::CB * G::RING)) {
            w.ok |= 1 << i;
            s.buf[slot][w.t[i] % G::CB] = x_entry_id(wi, w.app, i);
          }
        }
        w.pc = 3;
        return true;
      case 3:                                                // commits; completed blocks go on the list
        w.nfl = 0;
        for (int i = 0; i < G::N; ++i)
          if ((w.ok >> i) & 1) {
            const int slot = x_entry_bin(wi, w.app, i) * G::RING + (w.t[i] / G::CB) % G::RING;
            if (s.dn[slot]++ == G::CB - 1) { w.fl_slot[w.nfl] = (uint8_t)slot; w.fl_q[w.nfl] = w.t[i] / G::CB; ++w.nfl; }
          }
        w.pend &= ~w.ok;
        w.pc = w.nfl ? 4 : 8;
        return true;
      case 4:                                                // block reads
        for (int j = 0; j < w.nfl; ++j) memcpy(w.vblk[j], s.buf[w.fl_slot[j]], sizeof(w.vblk[j]));
        w.pc = 5;
        return true;
      case 5:
        for (int j = 0; j < w.nfl; ++j) s.dn[w.fl_slot[j]] = 0;
        w.pc = 6;
        return true;
      case 6:
        for (int j = 0; j < w.nfl; ++j) s.gen[w.fl_slot[j]] = w.fl_q[j] / G::RING + 1;
        w.pc = 7;
        return true;
      case 7:                                                // region stores
        for (int j = 0; j < w.nfl; ++j) memcpy(s.reg[w.fl_slot[j] / G::RING][w.fl_q[j]], w.vblk[j], sizeof(w.vblk[j]));
        w.pc = 8;
        return true;
      case 8:
        memset(w.gv, 0, sizeof(w.gv)); w.ok = 0; w.nfl = 0;          // (dead registers: keep equal states equal)
        memset(w.fl_slot, 0, sizeof(w.fl_slot)); memset(w.fl_q, 0, sizeof(w.fl_q)); memset(w.vblk, 0, sizeof(w.vblk));
        if (w.pend) { w.pc = 1; return true; }
        memset(w.t, 0, sizeof(w.t));
        if (++w.app == G::APPENDS) { w.done = true; return true; }
        w.pc = 0;
        return true;
    }
    return false;
  }

  bool check_final(XState s) {
    using G = XGeo;
    std::vector<int> seen_id(256, 0);
    for (int b = 0; b < G::BINS; ++b) {                      // drain + pass C
      const int T = s.tick[b], blk = T / G::CB, dd = T % G::CB;
      if (dd > 0) memcpy(s.reg[b][blk], s.buf[b * G::RING + blk % G::RING], sizeof(s.reg[b][blk]));
      for (int e = 0; e < T; ++e) {
        const uint16_t id = s.reg[b][e / G::CB][e % G::CB];
        if (id == 0 || id >= 256) return false;
        ++seen_id[id];
      }
    }
    for (int wv = 0; wv < G::WAVES; ++wv)
      for (int a = 0; a < G::APPENDS; ++a)
        for (int i = 0; i < G::N; ++i)
          if (seen_id[x_entry_id(wv, a, i)] != 1) return false;       // every entry delivered exactly once
    return true;
  }

  void dfs(const XState& s) {
    if (failed) return;
    std::string key(reinterpret_cast<const char*>(&s), sizeof(s));
    if (!seen.insert(key).second) return;
    ++states;
    bool any = false;
    for (int wi = 0; wi < XGeo::WAVES; ++wi) {
      if (s.w[wi].done) continue;
      // a wave that only spins (stage 1-3 with nothing writable and nothing changed) is still a step: the visited set ends the loop
      XState n;
      memcpy(&n, &s, sizeof(n));
      if (step(n, wi)) { any = true; dfs(n); }
    }
    if (!any) {
      ++finals;
      if (!check_final(s)) { failed = true; fprintf(stderr, "explore: a final state delivers an entry zero or several times\n"); }
    }
  }
};

int explore() {
  XExplorer x;
  XState s0;
  memset(&s0, 0, sizeof(s0));
  x.dfs(s0);
  printf("explore: %llu states, %llu final states, %s\n", (unsigned long long)x.states, (unsigned long long)x.finals,
         x.failed ? "FAILED" : "every entry delivered exactly once in all of them");
  return x.failed || x.finals == 0 ? 1 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "stress")) return stress(argc >= 3 ? atoi(argv[2]) : 50);
  if (argc >= 2 && !strcmp(argv[1], "explore")) return explore();
  fprintf(stderr, "usage: ring_model stress <seeds> | explore\n");
  return 2;
}
