// The synchronisation protocol of the kinematic optimisation's workgroup clusters (chd_kinopt_kernels.hpp, kc_sync) replayed on host threads: one thread per workgroup,
// relaxed 64-bit atomics for the tagged granules, random delays.  It checks the LOGIC the device code relies on -- not the device's memory system:
//   * every value a workgroup reads at synchronisation e is the one its owner published FOR e (tags), although slots are only two deep;
//   * that holds because every synchronisation gathers a value from every rank (nobody gets more than one synchronisation ahead) -- with `--neighbours-only`
//     the all-to-all gather is replaced by the neighbour's halo alone, and the check must FAIL (the test of the test);
//   * a missing member ends in the bounded wait, not in a hang.
// Build: g++ -O1 -g -std=c++17 -pthread [-fsanitize=thread] cluster_protocol.cpp -o cluster_protocol
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

enum { PARTS = 4, HALO = 24 };
struct Slot { std::atomic<uint64_t> gran[2][2 * (PARTS + HALO)]; };

static void put(std::atomic<uint64_t>* g2, double x, uint32_t tag) {
  uint64_t b; memcpy(&b, &x, 8);
  g2[0].store((b & 0xffffffff00000000ull) | tag, std::memory_order_relaxed);
  g2[1].store((b << 32) | tag, std::memory_order_relaxed);
}
static bool poll(const std::atomic<uint64_t>* g, uint32_t tag, uint64_t& v, std::atomic<int>& abort_flag, int patience_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    v = g->load(std::memory_order_relaxed);
    if ((uint32_t)v == tag) return true;
    if ((spins & 1023u) == 1023u) {
      if (abort_flag.load() || std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(patience_ms)) { abort_flag.store(1); return false; }
      std::this_thread::yield();
    }
  }
}
static bool get(const std::atomic<uint64_t>* g2, uint32_t tag, double& out, std::atomic<int>& abort_flag, int patience_ms) {
  uint64_t hi, lo;
  if (!poll(g2, tag, hi, abort_flag, patience_ms) || !poll(g2 + 1, tag, lo, abort_flag, patience_ms)) return false;
  const uint64_t b = (hi & 0xffffffff00000000ull) | (lo >> 32);
  memcpy(&out, &b, 8);
  return true;
}
static double expected(int rank, uint32_t e, int i) { return 1e6 * rank + 10.0 * e + i + 0.5; }

int main(int argc, char** argv) {
  int G = 8, rounds = 20000, absent = -1; bool neighbours_only = false;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--neighbours-only")) neighbours_only = true;
    else if (!strncmp(argv[a], "--absent=", 9)) absent = atoi(argv[a] + 9);
    else if (!strncmp(argv[a], "--rounds=", 9)) rounds = atoi(argv[a] + 9);
    else if (!strncmp(argv[a], "--G=", 4)) G = atoi(argv[a] + 4);
  }
  std::vector<Slot> slots(G);
  for (auto& s : slots) for (auto& p : s.gran) for (auto& g : p) g.store(0);
  std::atomic<int> abort_flag{0};
  std::atomic<long long> wrong{0}, done{0};
  auto body = [&](int me) {
    std::mt19937 rng(1234 + me);
    for (uint32_t e = 1; e <= (uint32_t)rounds; ++e) {
      const int par = e & 1, dir = (e & 2) ? +1 : -1;               // alternate halo directions as LSMR's two synchronisations do
      if (rng() % 7 == 0) std::this_thread::yield();
      if (rng() % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 50));
      for (int i = 0; i < HALO; ++i) put(&slots[me].gran[par][2 * (PARTS + i)], expected(me, e, PARTS + i), e);
      for (int i = 0; i < PARTS; ++i) put(&slots[me].gran[par][2 * i], expected(me, e, i), e);
      double v;
      if (!neighbours_only)
        for (int g = 0; g < G; ++g)
          for (int i = 0; i < PARTS; ++i) {
            if (!get(&slots[g].gran[par][2 * i], e, v, abort_flag, 300)) return;
            if (v != expected(g, e, i)) wrong.fetch_add(1);
          }
      const int nb = me + dir;
      if (nb >= 0 && nb < G)
        for (int i = 0; i < HALO; ++i) {
          if (!get(&slots[nb].gran[par][2 * (PARTS + i)], e, v, abort_flag, 300)) return;
          if (v != expected(nb, e, PARTS + i)) wrong.fetch_add(1);
        }
    }
    done.fetch_add(1);
  };
  std::vector<std::thread> th;
  for (int g = 0; g < G; ++g) if (g != absent) th.emplace_back(body, g);
  for (auto& t : th) t.join();
  printf("{\"G\": %d, \"rounds\": %d, \"finished\": %lld, \"wrong_values\": %lld, \"gave_up\": %d}\n", G, rounds, (long long)done.load(), (long long)wrong.load(), abort_flag.load());
  return 0;
}
