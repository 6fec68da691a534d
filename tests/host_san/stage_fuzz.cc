// Host-only harness for the staging half of the real-data path (amcpy_amd/csrc/amcx_upload.h: the fork-join
// Pool, stage_runs over memory and file sources, classify_layout), built by tests/test_host_cpu.py three ways --
//   g++ -O1 -g -fsanitize=address,undefined     g++ -O1 -g -fsanitize=thread     g++ -O2
// -- and run in the CPU suite.  No HIP, no GPU.  The reference's own threading defect is of exactly this class
// (worker threads sharing one queue and one output array, exceptions swallowed: feature_extraction.py:22-39,74),
// and this pool is hand-rolled: a generation counter polled lock-free, a condition variable, a pointer to a
// std::function on the caller's stack, pread into thread_local scratch, non-temporal stores.
//
//   stage_fuzz [seed]     exit code 0 and "STAGE_FUZZ_OK ..." on success; any mismatch aborts with a message
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <string>

#include "../../amcpy_amd/csrc/amcx_upload.h"

namespace {

struct Rng {
  unsigned long long s;
  explicit Rng(unsigned long long seed) : s(seed * 2654435761ULL + 88172645463325252ULL) {}
  unsigned long long next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)(next() % (unsigned long long)(hi - lo)); }   // [lo, hi)
  double real() { return (double)(int64_t)(next() >> 11) * (1.0 / 9007199254740992.0) * 200.0 - 100.0; }
};

[[noreturn]] void die(const char* what, int64_t a = 0, int64_t b = 0, int64_t c = 0) {
  fprintf(stderr, "stage_fuzz: %s (%lld, %lld, %lld)\n", what, (long long)a, (long long)b, (long long)c);
  abort();
}

// one random container: (S, K, N) frames inside a padded box laid out with a random axis order
struct Box {
  int64_t S, K, N;
  int64_t ss, sk, sn;          // element strides of the (snr, frame, sample) axes
  int64_t elems;               // elements of the padded box
  int kind;
  bool has_im;
  std::vector<double> re64, im64;      // kinds C128 (interleaved in re64) and F64_SPLIT
  std::vector<float> re32, im32;       // kinds C64 (interleaved in re32) and F32_SPLIT

  // what staging must deliver for (s, k, n): the complex64 value
  void expect(int64_t s, int64_t k, int64_t n, float* out) const {
    const int64_t e = s * ss + k * sk + n * sn;
    switch (kind) {
      case amcx::kSrcC64: out[0] = re32[2 * e]; out[1] = re32[2 * e + 1]; break;
      case amcx::kSrcC128: out[0] = (float)re64[2 * e]; out[1] = (float)re64[2 * e + 1]; break;
      case amcx::kSrcF32Split: out[0] = re32[e]; out[1] = has_im ? im32[e] : 0.f; break;
      default: out[0] = (float)re64[e]; out[1] = has_im ? (float)im64[e] : 0.f; break;
    }
  }
  amcx::Source memory() const {
    amcx::Source src;
    src.kind = kind;
    const bool f32 = kind == amcx::kSrcC64 || kind == amcx::kSrcF32Split;
    src.re = f32 ? reinterpret_cast<const char*>(re32.data()) : reinterpret_cast<const char*>(re64.data());
    if (has_im && kind >= amcx::kSrcF32Split)
      src.im = f32 ? reinterpret_cast<const char*>(im32.data()) : reinterpret_cast<const char*>(im64.data());
    return src;
  }
  size_t re_bytes() const {
    const bool f32 = kind == amcx::kSrcC64 || kind == amcx::kSrcF32Split;
    return f32 ? re32.size() * 4 : re64.size() * 8;
  }
  size_t im_bytes() const {
    if (!(has_im && kind >= amcx::kSrcF32Split)) return 0;
    return kind == amcx::kSrcF32Split ? im32.size() * 4 : im64.size() * 8;
  }
};

Box make_box(Rng& rng, bool want_unit_axis, bool big = false) {
  Box b;
  for (;;) {
    b.S = rng.range(1, 6); b.K = rng.range(1, 30); b.N = rng.range(2, 90);
    if (big) { b.K = rng.range(200, 400); b.N = rng.range(300, 600); }   // > 256 KiB staged: stage_runs cuts it into parts for the pool
    const int64_t ext[3] = {b.S + rng.range(0, 3), b.K + rng.range(0, 3), b.N + rng.range(0, 3)};
    int order[3] = {0, 1, 2};                                   // order[0] is the slowest axis
    for (int i = 2; i > 0; --i) { const int j = (int)rng.range(0, i + 1); const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int64_t st[3], acc = 1;
    for (int i = 2; i >= 0; --i) { st[order[i]] = acc; acc *= ext[order[i]]; }
    b.ss = st[0]; b.sk = st[1]; b.sn = st[2]; b.elems = acc;
    bool rows, inner; amcx::RunMap m;
    if (!want_unit_axis || amcx::classify_layout(b.S, b.K, (int32_t)b.N, b.ss, b.sk, b.sn, &rows, &inner, &m)) break;
  }
  b.kind = (int)rng.range(0, 4);
  b.has_im = b.kind < amcx::kSrcF32Split || rng.range(0, 4) != 0;     // a quarter of the split containers are real signals
  const bool interleaved = b.kind < amcx::kSrcF32Split;
  const size_t n = (size_t)b.elems * (interleaved ? 2 : 1);
  if (b.kind == amcx::kSrcC64 || b.kind == amcx::kSrcF32Split) {
    b.re32.resize(n);
    for (auto& v : b.re32) v = (float)rng.real();
    if (!interleaved && b.has_im) { b.im32.resize(n); for (auto& v : b.im32) v = (float)rng.real(); }
  } else {
    b.re64.resize(n);
    for (auto& v : b.re64) v = rng.real() * 1.0000001;             // not representable in float32: the rounding is exercised
    if (!interleaved && b.has_im) { b.im64.resize(n); for (auto& v : b.im64) v = rng.real() * 1.0000001; }
  }
  return b;
}

// stage units [first, first + count) of the box the way amcx.hip's stage_any / ctx_run_strided do, and check every value
void stage_and_check(const Box& b, const amcx::Source& src, amcx::Pool& pool, int64_t first, int64_t count,
                     bool rows, bool inner, const amcx::RunMap& map) {
  const int64_t F = b.S * b.K, unit = rows ? b.N : F, per_unit = rows ? 1 : map.cnt_b;
  // exactly the bytes the call may write, at an address that is 8 but not 16 bytes aligned half of the time:
  // AddressSanitizer sees one byte too many, and the unaligned head of the streaming-store loops runs
  std::vector<float> guard((size_t)(count * unit) * 2 + 4, -7.0f);
  float* dst = guard.data() + ((first & 1) ? 2 : 0);
  amcx::stage_runs(pool, reinterpret_cast<char*>(dst), src, map, first * per_unit, (first + count) * per_unit, false);
  for (int64_t u = 0; u < count; ++u)
    for (int64_t j = 0; j < unit; ++j) {
      int64_t s, k, n;
      if (rows) { const int64_t g = first + u; s = g / b.K; k = g % b.K; n = j; }
      else if (inner) { n = first + u; k = j / b.S; s = j % b.S; }
      else { n = first + u; s = j / b.K; k = j % b.K; }
      float want[2];
      b.expect(s, k, n, want);
      const float* got = dst + 2 * (u * unit + j);
      if (memcmp(got, want, 8) != 0) die("staged value differs", u, j, b.kind);
    }
  // nothing beyond the range was touched
  for (float* p = guard.data(); p < dst; ++p) if (*p != -7.0f) die("write before the destination");
  for (float* p = dst + 2 * count * unit; p < guard.data() + guard.size(); ++p) if (*p != -7.0f) die("write past the destination");
}

int fuzz_layouts(Rng& rng, const std::string& dir, int cases) {
  int seen = 0, files = 0;
  amcx::Pool pool;
  for (int c = 0; c < cases; ++c) {
    const Box b = make_box(rng, true, c % 6 == 5);
    bool rows = false, inner = false;
    amcx::RunMap map;
    if (!amcx::classify_layout(b.S, b.K, (int32_t)b.N, b.ss, b.sk, b.sn, &rows, &inner, &map)) die("layout refused");
    seen |= rows ? 1 : inner ? 2 : 4;
    const int64_t n_units = rows ? b.S * b.K : b.N;
    const int64_t first = rng.range(0, n_units), count = rng.range(1, n_units - first + 1);
    pool.resize((int)rng.range(1, 6));                              // the pool changes size between runs
    stage_and_check(b, b.memory(), pool, first, count, rows, inner, map);
    if (c % 2 == 0) {
      // the same container from a FILE: real array at offset 24, imaginary array behind it
      const std::string path = dir + "/box" + std::to_string(c) + ".bin";
      FILE* fh = fopen(path.c_str(), "wb");
      if (fh == nullptr) die("cannot create the scratch file");
      const char pad[24] = {0};
      const amcx::Source mem = b.memory();
      fwrite(pad, 1, sizeof pad, fh);
      fwrite(mem.re, 1, b.re_bytes(), fh);
      if (b.im_bytes()) fwrite(mem.im, 1, b.im_bytes(), fh);
      fclose(fh);
      amcx::Source f;
      f.kind = b.kind;
      f.fd = open(path.c_str(), O_RDONLY);
      if (f.fd < 0) die("cannot open the scratch file");
      f.re_off = 24;
      f.im_off = b.im_bytes() ? (int64_t)(24 + b.re_bytes()) : -1;
      std::atomic<int> io_error{0};
      f.io_error = &io_error;
      stage_and_check(b, f, pool, first, count, rows, inner, map);
      if (io_error.load() != 0) die("read error on a complete file", io_error.load());
      // ... and from a file that ends inside the variable: an error code, nothing written out of bounds, no hang
      // (cut at half the offset of the last element the configuration uses: the box is padded, its tail may be unused)
      const int64_t last = (b.S - 1) * b.ss + (b.K - 1) * b.sk + (b.N - 1) * b.sn;
      const int64_t elem_bytes = (int64_t)(b.re_bytes() / (size_t)b.elems);
      if (truncate(path.c_str(), (off_t)(24 + last * elem_bytes / 2)) != 0) die("truncate");
      std::vector<float> out((size_t)(n_units * (rows ? b.N : b.S * b.K)) * 2);
      amcx::stage_runs(pool, reinterpret_cast<char*>(out.data()), f, map, 0, n_units * (rows ? 1 : map.cnt_b), false);
      if (io_error.load() == 0) die("a truncated file went unnoticed");
      close(f.fd);
      unlink(path.c_str());
      ++files;
    }
  }
  if (seen != 7) die("the random layouts did not cover rows / snr-inner planes / frame-inner planes", seen);
  // a container with no unit-stride axis is refused, not staged
  {
    bool rows, inner; amcx::RunMap m;
    if (amcx::classify_layout(3, 4, 8, 2, 6, 24, &rows, &inner, &m)) die("a layout without a contiguous axis was accepted");
  }
  return files;
}

// 1000 back-to-back runs with resizes in between: every part runs exactly once, whatever the pool's size was a
// moment ago, whether its workers poll or sleep
void hammer_pool(Rng& rng) {
  amcx::Pool pool;
  std::vector<std::atomic<int>> hits(64);
  long long total = 0;
  for (int it = 0; it < 1000; ++it) {
    if (it % 7 == 0) pool.resize((int)rng.range(1, 9));
    if (it % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(400));     // workers go to sleep
    const int parts = (int)rng.range(0, 40);
    for (auto& h : hits) h.store(0, std::memory_order_relaxed);
    std::atomic<long long> sum{0};
    const std::function<void(int)> job = [&](int p) {            // lives on this stack frame: the pool holds a pointer to it
      hits[(size_t)p].fetch_add(1, std::memory_order_relaxed);
      sum.fetch_add(p + 1, std::memory_order_relaxed);
    };
    pool.run(parts, job);
    for (int p = 0; p < 64; ++p)
      if (hits[(size_t)p].load() != (p < parts ? 1 : 0)) die("a part ran the wrong number of times", it, p, parts);
    if (sum.load() != (long long)parts * (parts + 1) / 2) die("parts lost", it, parts);
    total += parts;
  }
  pool.resize(1);
  pool.resize(4);                                                   // destroyed with workers alive
  if (total == 0) die("nothing ran");
}

// several pools at once (one context per device, each with its own staging threads: DeviceFanOut)
void concurrent_pools(Rng& rng) {
  const Box b = make_box(rng, true, true);
  bool rows = false, inner = false;
  amcx::RunMap map;
  amcx::classify_layout(b.S, b.K, (int32_t)b.N, b.ss, b.sk, b.sn, &rows, &inner, &map);
  const int64_t n_units = rows ? b.S * b.K : b.N;
  std::vector<std::thread> callers;
  for (int t = 0; t < 3; ++t)
    callers.emplace_back([&, t] {
      amcx::Pool pool;
      pool.resize(2 + t);
      for (int it = 0; it < 12; ++it) stage_and_check(b, b.memory(), pool, 0, n_units, rows, inner, map);
    });
  for (auto& th : callers) th.join();
}

// host placement (round 5): the cpulist parser on good and malformed text, the PCI-tree reader on a fake tree, and a
// pool bound to a subset of the CPUs this process may use -- every part must run on one of them, the caller's own mask
// must come back after the guard, and re-binding while workers exist must replace them cleanly
void placement(const char* dir) {
  auto eq = [](const std::vector<int>& a, std::initializer_list<int> b) { return a == std::vector<int>(b); };
  if (!eq(amcx::parse_cpulist("0-3,8,10-11\n"), {0, 1, 2, 3, 8, 10, 11})) die("cpulist ranges");
  if (!eq(amcx::parse_cpulist("5"), {5}) || !eq(amcx::parse_cpulist(""), {}) || !eq(amcx::parse_cpulist("\n"), {})) die("cpulist singles");
  if (!eq(amcx::parse_cpulist("2-1"), {}) || !eq(amcx::parse_cpulist("1,x"), {1}) || !eq(amcx::parse_cpulist("3-"), {})) die("cpulist malformed");
  if (!eq(amcx::parse_cpulist("0-1,99999999"), {0, 1})) die("cpulist out of range");
  const std::string root = std::string(dir) + "/sys";
  auto put = [&](const char* bdf, const char* node, const char* cpus) {
    const std::string d = root + "/bus/pci/devices/" + bdf;
    const std::string cmd = "mkdir -p '" + d + "'";
    if (system(cmd.c_str()) != 0) die("mkdir");
    if (node) { FILE* f = fopen((d + "/numa_node").c_str(), "w"); fputs(node, f); fclose(f); }
    if (cpus) { FILE* f = fopen((d + "/local_cpulist").c_str(), "w"); fputs(cpus, f); fclose(f); }
  };
  put("0000:05:00.0", "0\n", "0-3,8-11\n");
  put("0000:85:00.0", "1\n", "4-7,12-15\n");
  put("0000:a5:00.0", "-1\n", "0-15\n");
  put("0000:b5:00.0", "1\n", nullptr);
  amcx::NumaPlace a = amcx::numa_place_of(root, "0000:05:00.0"), b = amcx::numa_place_of(root, "0000:85:00.0");
  if (a.node != 0 || !eq(a.cpus, {0, 1, 2, 3, 8, 9, 10, 11}) || b.node != 1 || b.cpus.size() != 8 || b.cpus[0] != 4) die("place");
  if (!amcx::numa_place_of(root, "0000:A5:00.0").empty()) die("node -1 must not bind");
  if (!amcx::numa_place_of(root, "0000:b5:00.0").empty() || !amcx::numa_place_of(root, "0000:ff:00.0").empty() ||
      !amcx::numa_place_of(root, "../x").empty() || !amcx::numa_place_of(root, "").empty()) die("missing files must not bind");
  const std::string rm = "rm -rf '" + root + "'";
  if (system(rm.c_str()) != 0) die("rm");

  cpu_set_t mine;
  CPU_ZERO(&mine);
  if (pthread_getaffinity_np(pthread_self(), sizeof mine, &mine) != 0) die("getaffinity");
  std::vector<int> usable;
  for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &mine)) usable.push_back(c);
  if (usable.empty()) die("no cpus");
  std::vector<int> want(usable.begin(), usable.begin() + (usable.size() > 2 ? 2 : 1));
  want.push_back(CPU_SETSIZE - 1);                                  // one this process (almost certainly) may not use: dropped
  const std::vector<int> ok = amcx::allowed_subset(want);
  if (ok.empty() || ok.size() > want.size()) die("allowed_subset");
  amcx::Pool pool;
  pool.resize(4);
  pool.set_cpus(want);                                              // workers exist: they are replaced
  if (pool.size() != 4) die("set_cpus changed the size");
  {
    amcx::AffinityGuard g(want);
    if (!g.bound()) die("guard did not bind");
    for (int it = 0; it < 50; ++it) {
      std::atomic<int> off{0};
      const std::function<void(int)> job = [&](int) {
        const int cpu = sched_getcpu();
        bool in = false;
        for (int c : ok) in = in || c == cpu;
        if (!in) off.fetch_add(1);
      };
      pool.run(16, job);
      if (off.load() != 0) die("a part ran outside the bound CPUs", it, off.load());
    }
  }
  cpu_set_t after;
  CPU_ZERO(&after);
  pthread_getaffinity_np(pthread_self(), sizeof after, &after);
  if (!CPU_EQUAL(&mine, &after)) die("the caller's mask did not come back");
  pool.set_cpus({});                                                // unbound again
  std::atomic<int> ran{0};
  const std::function<void(int)> job2 = [&](int) { ran.fetch_add(1); };
  pool.run(8, job2);
  if (ran.load() != 8) die("unbound pool lost parts");
  amcx::AffinityGuard none((std::vector<int>()));
  if (none.bound()) die("empty guard bound");
}

}  // namespace

int main(int argc, char** argv) {
  const unsigned long long seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2026;
  char tmpl[] = "/tmp/amcx_stage_fuzz_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if (dir == nullptr) die("mkdtemp");
  Rng rng(seed);
  const int files = fuzz_layouts(rng, dir, 60);
  hammer_pool(rng);
  concurrent_pools(rng);
  placement(dir);
  rmdir(dir);
  printf("STAGE_FUZZ_OK seed %llu: 60 layouts (%d also from a file, each truncated once), 1000 pool runs, 3 concurrent pools, placement\n",
         seed, files);
  return 0;
}
