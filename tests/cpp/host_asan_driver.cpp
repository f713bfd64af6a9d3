// Drives the C ABI from several threads -- lone calls, batches of every layout class, verify, the pipe in its three
// delivery modes, shutdown and re-init -- and checks every digest against the CPU oracle.  Built twice by
// tests/cpp/Makefile: plain, and with the HOST code of the library and of this file under AddressSanitizer +
// UndefinedBehaviorSanitizer (the GPU code is not instrumented), which is what finds out-of-bounds staging offsets,
// stale pointers into the gather pool's jobs and the like while the real GPU path runs.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ozone_checksum.h"
#include "../../oracle/ozcs_oracle.h"

static std::atomic<int> g_fail{0};
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                       \
      fprintf(stderr, " [%s]\n", ozcs_last_error());      \
      g_fail++;                                           \
    }                                                     \
  } while (0)

constexpr uint64_t MIB = 1ull << 20;

static std::vector<uint8_t> bytes(uint64_t seed, uint64_t n) {
  std::vector<uint8_t> v(n ? n : 1);
  oracle_fill_splitmix64(seed, v.data(), n);
  v.resize(n);
  return v;
}

static std::vector<uint8_t> want(int type, uint32_t bpc, const std::vector<uint8_t>& d) {
  const uint64_t n = ozcs_num_checksums(d.size(), bpc);
  std::vector<uint8_t> out(n * ozcs_digest_size(type) + 1);
  const int64_t got = oracle_compute(type, bpc, d.data(), d.size(), out.data(), out.size());
  out.resize(got > 0 ? (size_t)got * ozcs_digest_size(type) : 0);
  return out;
}

static void lone_calls() {
  const struct { int type; uint32_t bpc; uint64_t len; } cases[] = {
      {OZCS_CRC32, 1u << 20, 4 * MIB},      {OZCS_CRC32C, 16384, 4 * MIB + 77}, {OZCS_CRC32, 512, 100001},
      {OZCS_MD5, 65536, MIB + 3},           {OZCS_SHA256, 16384, 300000},       {OZCS_CRC32, 1u << 20, 40 * MIB + 5},
      {OZCS_CRC32C, 100000, 33 * MIB},      {OZCS_CRC32, 1, 700},               {OZCS_CRC32C, 1u << 20, 1}};
  for (const auto& c : cases) {
    const auto d = bytes(c.len + c.bpc, c.len);
    const auto exp = want(c.type, c.bpc, d);
    std::vector<uint8_t> got(exp.size());   // exactly as large as needed: a write past it is caught
    uint64_t n = 0;
    CHECK(ozcs_compute(c.type, c.bpc, d.data(), d.size(), got.data(), got.size(), &n) == OZCS_OK, "compute %d %u %llu", c.type, c.bpc, (unsigned long long)c.len);
    CHECK(n * ozcs_digest_size(c.type) == exp.size() && got == exp, "digests %d %u %llu", c.type, c.bpc, (unsigned long long)c.len);
    int64_t fm = 7;
    CHECK(ozcs_verify(c.type, c.bpc, d.data(), d.size(), exp.data(), n, 0, &fm) == OZCS_OK && fm == -1, "verify");
    if (exp.size() > 8) {
      auto bad = exp;
      bad[bad.size() - 1] ^= 0x80;
      CHECK(ozcs_verify(c.type, c.bpc, d.data(), d.size(), bad.data(), n, 0, &fm) == OZCS_OK && fm == (int64_t)n - 1, "verify mismatch index");
    }
  }
}

static void batches(uint64_t seed) {
  struct Spec { int type; uint32_t bpc; uint64_t len; };
  std::vector<Spec> plan;
  for (int i = 0; i < 9; i++) plan.push_back({OZCS_CRC32, 16384, 65536});           // a run of small chunks
  for (int i = 0; i < 3; i++) plan.push_back({OZCS_CRC32, 16384, 65537});           // stride != length
  plan.push_back({OZCS_MD5, 16384, 65536});
  for (int i = 0; i < 6; i++) plan.push_back({OZCS_CRC32C, 1u << 20, 4 * MIB});     // a run of large chunks
  plan.push_back({OZCS_SHA256, 65536, MIB});
  plan.push_back({OZCS_CRC32, 512, 999});                                           // lone small chunk
  for (int i = 0; i < 40; i++) plan.push_back({OZCS_CRC32, 512, 1000});
  plan.push_back({OZCS_CRC32, 1u << 20, 130 * MIB + 5});                            // cut into pieces
  plan.push_back({OZCS_CRC32C, 12u << 20, 30 * MIB});
  plan.push_back({OZCS_CRC32C, 1u << 20, 1});
  plan.push_back({OZCS_MD5, 1u << 20, 9 * MIB});                                    // job list
  for (int i = 0; i < 200; i++) plan.push_back({(i % 3) ? OZCS_CRC32 : OZCS_CRC32C, 4096, (uint64_t)(1 + (i * 7919) % 70000)});
  std::vector<std::vector<uint8_t>> data, got, exp;
  std::vector<ozcs_chunk> chunks(plan.size());
  for (size_t i = 0; i < plan.size(); i++) {
    data.push_back(bytes(seed + i, plan[i].len));
    exp.push_back(want(plan[i].type, plan[i].bpc, data.back()));
    got.emplace_back(exp.back().size());
  }
  for (size_t i = 0; i < plan.size(); i++) {
    chunks[i] = ozcs_chunk{};
    chunks[i].data = data[i].data();
    chunks[i].len = data[i].size();
    chunks[i].sums = got[i].data();
    chunks[i].sums_cap = got[i].size();
    chunks[i].bpc = plan[i].bpc;
    chunks[i].type = plan[i].type;
  }
  CHECK(ozcs_compute_batch(chunks.data(), (uint32_t)chunks.size()) == OZCS_OK, "compute_batch");
  for (size_t i = 0; i < plan.size(); i++)
    CHECK(chunks[i].status == OZCS_OK && got[i] == exp[i], "batch chunk %zu (%d, %u, %llu)", i, plan[i].type, plan[i].bpc, (unsigned long long)plan[i].len);
  std::vector<ozcs_vchunk> v(plan.size());
  exp[2][0] ^= 1;
  for (size_t i = 0; i < plan.size(); i++) {
    v[i] = ozcs_vchunk{};
    v[i].data = data[i].data();
    v[i].len = data[i].size();
    v[i].expected = exp[i].data();
    v[i].n_expected = ozcs_num_checksums(plan[i].len, plan[i].bpc);
    v[i].bpc = plan[i].bpc;
    v[i].type = plan[i].type;
  }
  CHECK(ozcs_verify_batch(v.data(), (uint32_t)v.size()) == OZCS_OK, "verify_batch");
  for (size_t i = 0; i < plan.size(); i++) CHECK(v[i].first_mismatch == (i == 2 ? 0 : -1), "verify_batch chunk %zu", i);
}

static void pipe_per_ticket(ozcs_pipe* pipe, uint64_t seed, int rounds) {
  for (int i = 0; i < rounds; i++) {
    const int type = (i % 5 == 4) ? OZCS_MD5 : (i % 2 ? OZCS_CRC32 : OZCS_CRC32C);
    const uint32_t bpc = (i % 3) ? 65536 : (1u << 20);
    const auto d = bytes(seed * 1000 + i, (i % 7 == 0) ? 2 * MIB : 2 * MIB - 4099 * (uint64_t)i);
    const auto exp = want(type, bpc, d);
    std::vector<uint8_t> got(exp.size());
    int ticket = -1, rc;
    while ((rc = ozcs_pipe_submit(pipe, type, bpc, d.data(), d.size(), got.data(), got.size(), &ticket)) == OZCS_EBUSY)
      std::this_thread::yield();
    CHECK(rc == OZCS_OK, "pipe submit");
    uint64_t n = 0;
    CHECK(ozcs_pipe_wait(pipe, ticket, &n) == OZCS_OK, "pipe wait");
    CHECK(n * ozcs_digest_size(type) == exp.size() && got == exp, "pipe digests round %d", i);
  }
}

static void pipe_in_order() {
  ozcs_pipe* pipe = nullptr;
  CHECK(ozcs_pipe_create(0, 6, 2 * MIB, &pipe) == OZCS_OK, "pipe create");
  if (!pipe) return;
  CHECK(ozcs_pipe_set_in_order(pipe, 1) == OZCS_OK, "in order");
  struct Job { std::vector<uint8_t> d, exp, got; int ticket; int type; };
  std::vector<Job> jobs(60);
  size_t delivered = 0;
  auto deliver = [&](int block) {
    int t = -1, st = 0;
    uint64_t n = 0;
    const int rc = ozcs_pipe_next_in_order(pipe, block, &t, &st, &n);
    if (rc != OZCS_OK) return false;
    Job& j = jobs[delivered++];
    CHECK(t == j.ticket && st == OZCS_OK && j.got == j.exp, "in-order delivery %zu", delivered - 1);
    return true;
  };
  for (size_t i = 0; i < jobs.size(); i++) {
    Job& j = jobs[i];
    j.type = (i % 4 == 3) ? OZCS_SHA256 : OZCS_CRC32;
    j.d = bytes(5000 + i, 2 * MIB - 31 * i);
    j.exp = want(j.type, 65536, j.d);
    j.got.assign(j.exp.size(), 0);
    void* buf = nullptr;
    int rc;
    while ((rc = ozcs_pipe_acquire_wait(pipe, &buf, &j.ticket)) == OZCS_EBUSY) deliver(1);
    CHECK(rc == OZCS_OK && buf, "acquire_wait");
    if (rc != OZCS_OK) break;
    memcpy(buf, j.d.data(), j.d.size());
    CHECK(ozcs_pipe_submit_slot(pipe, j.ticket, j.type, 65536, j.d.size(), j.got.data(), j.got.size()) == OZCS_OK, "submit_slot");
    while (deliver(0)) {}
  }
  while (delivered < jobs.size() && deliver(1)) {}
  CHECK(delivered == jobs.size(), "all delivered");
  ozcs_pipe_destroy(pipe);
}

int main() {
  if (!ozcs_available()) {
    fprintf(stderr, "no GPU: %s\n", ozcs_last_error());
    return 2;
  }
  lone_calls();
  batches(100);
  {  // concurrent callers: lone calls and batches beside each other share the gather pool
    std::vector<std::thread> th;
    for (int t = 0; t < 3; t++) th.emplace_back(lone_calls);
    th.emplace_back(batches, 900);
    for (auto& t : th) t.join();
  }
  {
    ozcs_pipe* pipe = nullptr;
    CHECK(ozcs_pipe_create(0, 5, 2 * MIB, &pipe) == OZCS_OK, "pipe create");
    if (pipe) {
      std::vector<std::thread> th;
      for (int t = 0; t < 4; t++) th.emplace_back(pipe_per_ticket, pipe, (uint64_t)t + 1, 25);
      std::atomic<bool> stop{false};
      std::thread drainer([&] { while (!stop) (void)ozcs_pipe_drain(pipe); });
      for (auto& t : th) t.join();
      stop = true;
      drainer.join();
      ozcs_pipe_destroy(pipe);
    }
  }
  pipe_in_order();
  ozcs_shutdown();               // joins the gather pool's helpers
  CHECK(ozcs_init(0) == OZCS_OK, "re-init");
  lone_calls();
  batches(300);
  ozcs_shutdown();
  if (g_fail) {
    fprintf(stderr, "%d check(s) failed\n", g_fail.load());
    return 1;
  }
  printf("host driver ok\n");
  return 0;
}
