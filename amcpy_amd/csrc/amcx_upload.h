// Host side of the real-data path: pageable container memory -> pinned staging slots, by a small
// persistent thread pool, in whatever order the container keeps its samples.
//
// The reference hands every frame to a worker thread as a numpy slice of what scipy.io.loadmat
// returned (feature_extraction.py:46-48,64-72).  That array is pageable, complex128 and
// Fortran-ordered; the copy engine wants pinned memory, the kernels want complex64.  One pass of
// the staging threads does both conversions that have to happen on the host anyway: it copies
// contiguous RUNS of the source (whole sample planes of a Fortran-ordered container, whole rows of a
// C-ordered one) into a pinned slot and rounds doubles to float32 on the way (cvtpd2ps: round to
// nearest even, bit-identical to numpy's astype and to the GPU's conversion), so that PCIe carries
// 8 bytes per sample instead of 16.  MATLAB v5 files keep real and imaginary parts as two separate
// arrays; the split kinds interleave them in the same pass, so a memory-mapped .mat goes up without
// a complex array ever being built on the host (amcpy_amd/matfile.py).  Measured on the MI355X box
// (profiles/r3_h2d_probe.txt): 8 threads copy pageable -> pinned at 138 GB/s, the link does 57 GB/s.
#pragma once

#include <ctype.h>
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#define AMCX_STAGE_SSE2 1
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace amcx {

// ---- where a device's host work should run ------------------------------------------------------
// One process can drive all eight GPUs of a node (DeviceFanOut, feature_extraction.py), each with its own staging
// threads and pinned slots.  The node has two sockets; a GPU hangs off one of them, and a staging thread on the other
// socket writes its pinned slot across the socket link before the copy engine reads it back across the same link.
// The kernel's PCI sysfs tree says which CPUs are local to a device:
//     <sysfs>/bus/pci/devices/<domain:bus:dev.fn>/numa_node        (-1: the platform does not say -- one node, a VM)
//     <sysfs>/bus/pci/devices/<domain:bus:dev.fn>/local_cpulist    ("0-63,128-191")
// numa_place_of reads the two; node -1, a missing file or an empty list give an empty place, and an empty place binds
// nothing.  Host-only logic: tests feed it a fake tree (tests/test_host_cpu.py), stage_fuzz.cc runs it under the sanitizers.
struct NumaPlace {
  int node = -1;
  std::vector<int> cpus;
  bool empty() const { return node < 0 || cpus.empty(); }
};

// "0-3,8,10-11\n" -> {0,1,2,3,8,10,11}; anything malformed ends the list where it stops making sense
inline std::vector<int> parse_cpulist(const char* text) {
  std::vector<int> out;
  const char* p = text;
  while (p != nullptr && *p != '\0') {
    while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') ++p;
    if (!isdigit((unsigned char)*p)) break;
    char* end = nullptr;
    long a = strtol(p, &end, 10), b = a;
    p = end;
    if (*p == '-') {
      ++p;
      if (!isdigit((unsigned char)*p)) break;
      b = strtol(p, &end, 10);
      p = end;
    }
    if (a < 0 || b < a || b >= 1 << 16) break;
    for (long c = a; c <= b; ++c) out.push_back((int)c);
  }
  return out;
}

inline bool read_small_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "r");
  if (f == nullptr) return false;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[n] = '\0';
  *out = buf;
  return true;
}

inline NumaPlace numa_place_of(const std::string& sysfs_root, const std::string& pci_bus_id) {
  NumaPlace place;
  std::string bdf = pci_bus_id;
  for (auto& ch : bdf) ch = (char)tolower((unsigned char)ch);
  if (bdf.empty() || bdf.find('/') != std::string::npos) return place;
  const std::string dir = sysfs_root + "/bus/pci/devices/" + bdf + "/";
  std::string text;
  if (!read_small_file(dir + "numa_node", &text)) return place;
  char* end = nullptr;
  const long node = strtol(text.c_str(), &end, 10);
  if (end == text.c_str() || node < 0) return place;
  if (!read_small_file(dir + "local_cpulist", &text)) return place;
  place.cpus = parse_cpulist(text.c_str());
  if (!place.cpus.empty()) place.node = (int)node;
  return place;
}

// the CPUs of `want` that the calling thread may run on at all (a container's cpuset, taskset): binding never
// widens what the process was given.  Empty: nothing to bind to.
inline std::vector<int> allowed_subset(const std::vector<int>& want) {
  std::vector<int> out;
  cpu_set_t cur;
  CPU_ZERO(&cur);
  if (pthread_getaffinity_np(pthread_self(), sizeof cur, &cur) != 0) return out;
  for (int c : want)
    if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &cur)) out.push_back(c);
  return out;
}

inline bool bind_this_thread(const std::vector<int>& cpus) {
  if (cpus.empty()) return false;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus)
    if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
  return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
}

// the CALLING thread on `cpus` for the lifetime of the guard (it stages too, and what it allocates and first touches
// in that time -- the pinned slots -- lands on that node); its own mask comes back afterwards
class AffinityGuard {
 public:
  explicit AffinityGuard(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    CPU_ZERO(&old_);
    if (pthread_getaffinity_np(pthread_self(), sizeof old_, &old_) != 0) return;
    const std::vector<int> ok = allowed_subset(cpus);
    bound_ = bind_this_thread(ok);
  }
  ~AffinityGuard() { if (bound_) (void)pthread_setaffinity_np(pthread_self(), sizeof old_, &old_); }
  bool bound() const { return bound_; }
  AffinityGuard(const AffinityGuard&) = delete;
  AffinityGuard& operator=(const AffinityGuard&) = delete;
 private:
  cpu_set_t old_;
  bool bound_ = false;
};

// ---- a fork-join pool whose caller works too ----------------------------------------------------
// run(parts, f) calls f(0..parts-1) on the workers AND the calling thread and returns when all are
// done.  Parts are handed out one at a time, so a worker that wakes late (a condition-variable wake
// costs tens of microseconds, a 2 MB part takes as long) simply finds less left to do; and a worker
// that has just finished keeps polling for ~200 us before it goes back to sleep, because the chunks of
// one upload follow each other within that time (a 16 MB modulation is five chunks in half a
// millisecond: with sleeping workers the caller staged most of it alone).
class Pool {
 public:
  ~Pool() { resize(1); }
  int size() const { return (int)workers_.size() + 1; }
  // the CPUs the workers run on (empty: wherever the scheduler puts them); existing workers are replaced
  void set_cpus(const std::vector<int>& cpus) {
    if (cpus == cpus_) return;
    const int n = size();
    resize(1);
    cpus_ = cpus;
    resize(n);
  }
  const std::vector<int>& cpus() const { return cpus_; }
  void resize(int n) {             // n threads in all (the caller is one of them)
    if (n < 1) n = 1;
    if (n == size()) return;
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      gen_.store(gen_.load(std::memory_order_relaxed) + 1, std::memory_order_release);   // pollers look, see stop_
    }
    wake_.notify_all();
    for (auto& t : workers_) t.join();
    workers_.clear();
    stop_ = false;
    for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
  }
  void run(int parts, const std::function<void(int)>& f) {
    if (parts <= 0) return;
    if (workers_.empty() || parts == 1) {
      for (int p = 0; p < parts; ++p) f(p);
      return;
    }
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; parts_ = parts; next_ = 0; left_ = parts;
      gen_.store(gen_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) wake_.notify_all();
    work();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return left_ == 0; });
    job_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      int p;
      const std::function<void(int)>* f;
      {
        std::lock_guard<std::mutex> g(m_);
        if (job_ == nullptr || next_ >= parts_) return;
        p = next_++;
        f = job_;
      }
      (*f)(p);
      std::lock_guard<std::mutex> g(m_);
      if (--left_ == 0) done_.notify_all();
    }
  }
  void loop() {
    if (!cpus_.empty()) (void)bind_this_thread(allowed_subset(cpus_));
    unsigned long long seen = 0;
    for (;;) {
      // poll for the next run for a while (lock-free), then sleep on the condition variable
      const auto t0 = std::chrono::steady_clock::now();
      bool fresh = false;
      for (int spin = 0;; ++spin) {
        if (gen_.load(std::memory_order_acquire) != seen) { fresh = true; break; }
        if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#endif
      }
      if (!fresh) {
        std::unique_lock<std::mutex> g(m_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        wake_.wait(g, [&] { return stop_ || gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        if (stop_) return;
      }
      {
        std::lock_guard<std::mutex> g(m_);
        if (stop_) return;
        seen = gen_.load(std::memory_order_acquire);
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::vector<int> cpus_;
  std::mutex m_;
  std::condition_variable wake_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int parts_ = 0, next_ = 0, left_ = 0;
  std::atomic<unsigned long long> gen_{0};
  std::atomic<int> sleepers_{0};
  bool stop_ = false;
};

// ---- source description ---------------------------------------------------------------------------
// kinds as in include/amcx.h (AMCX_SRC_*): interleaved complex64 / complex128, or split real / imaginary
// arrays of float32 / float64 (im == nullptr: a real signal, imaginary part zero)
//
// A source is memory (re / im pointers) or a FILE (fd >= 0; re_off / im_off are byte offsets of the same arrays,
// im_off < 0: no imaginary part).  A file is read with pread by the staging threads themselves, 16 K elements at a
// time into a per-thread scratch that stays in the core's L2, and converted from there into the pinned slot: the
// container never exists in host memory outside the page cache.  First-touching the pages of a fresh mapping of a
// 436 MB variable costs 30 ms however many threads fault, reading it into a fresh buffer as much again for the
// buffer's own pages (profiles/r3_read_probe.txt); pread into 256 KB that every thread reuses has neither cost.
struct Source {
  const char* re = nullptr;
  const char* im = nullptr;
  int kind = 0;
  int fd = -1;
  int64_t re_off = 0, im_off = -1;
  std::atomic<int>* io_error = nullptr;     // set to an errno (EIO for a short file) by the first read that fails
  bool has_im() const { return fd >= 0 ? im_off >= 0 : im != nullptr; }
};
enum { kSrcC64 = 0, kSrcC128 = 1, kSrcF32Split = 2, kSrcF64Split = 3 };

// run i of the source starts at element (i / cnt_b) * stride_a + (i % cnt_b) * stride_b and is
// run_len contiguous elements long; staged runs are packed back to back
struct RunMap {
  int64_t cnt_b = 1, stride_a = 0, stride_b = 0, run_len = 0;
  int64_t offset(int64_t i) const { return (i / cnt_b) * stride_a + (i % cnt_b) * stride_b; }
};

// which axis is contiguous decides how a container of (S, K) frames of N samples with element strides (ss, sk, sn)
// goes up: rows (a frame's samples contiguous) or sample planes (the snr or the frame axis contiguous); false: no axis
// has unit stride.  Host-only logic (tests/host_san/stage_fuzz.cc runs it under the sanitizers).
inline bool classify_layout(int64_t S, int64_t K, int32_t N, int64_t ss, int64_t sk, int64_t sn, bool* rows,
                            bool* inner_snr, RunMap* map) {
  const int64_t F = S * K;
  *rows = sn == 1;
  *inner_snr = false;
  if (*rows) {                        // run = one frame's N samples; run index = g = s * K + k
    map->cnt_b = K; map->stride_a = ss; map->stride_b = sk; map->run_len = N;
  } else if (sk == 1 && K > 1) {      // plane position j = s * K + k = g; runs of K frames
    map->cnt_b = S; map->stride_a = sn; map->stride_b = ss; map->run_len = K;
    if (ss == K) { map->cnt_b = 1; map->run_len = F; }          // the plane is one run
  } else if (ss == 1 || S == 1) {     // plane position j = k * S + s; runs of S snr values
    *inner_snr = true;
    map->cnt_b = K; map->stride_a = sn; map->stride_b = sk; map->run_len = S;
    if (sk == S) { map->cnt_b = 1; map->run_len = F; }
  } else {
    return false;
  }
  return true;
}

// The two conversions that carry the real-data path, written out for SSE2 (baseline x86-64): doubles are
// rounded with cvtpd2ps (round to nearest even under the default MXCSR, as numpy's astype and the GPU's
// conversion do) and the result leaves with non-temporal stores -- a pinned slot is written once and read
// by the copy engine, so keeping it out of the caches saves the read-for-ownership of every line (a third of
// the traffic of a plain store loop).
#ifdef AMCX_STAGE_SSE2
// dst[i] = (float) src[i], n doubles (n even: whole complex elements)
inline void round_doubles(float* dst, const double* src, int64_t n) {
  int64_t i = 0;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 8 && n >= 2) {      // one complex element up to 16-byte alignment
    dst[0] = (float)src[0]; dst[1] = (float)src[1];
    i = 2;
  }
  if ((reinterpret_cast<uintptr_t>(dst + i) & 15) == 0) {
    for (; i + 8 <= n; i += 8) {
      const __m128 a = _mm_movelh_ps(_mm_cvtpd_ps(_mm_loadu_pd(src + i)), _mm_cvtpd_ps(_mm_loadu_pd(src + i + 2)));
      const __m128 b = _mm_movelh_ps(_mm_cvtpd_ps(_mm_loadu_pd(src + i + 4)), _mm_cvtpd_ps(_mm_loadu_pd(src + i + 6)));
      _mm_stream_ps(dst + i, a);
      _mm_stream_ps(dst + i + 4, b);
    }
  }
  for (; i < n; ++i) dst[i] = (float)src[i];
}
// dst[i] = ((float) re[i], (float) im[i]), n complex elements
inline void round_interleave_doubles(float* dst, const double* re, const double* im, int64_t n) {
  int64_t i = 0;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 8 && n >= 1) {
    dst[0] = (float)re[0]; dst[1] = (float)im[0];
    i = 1;
  }
  if ((reinterpret_cast<uintptr_t>(dst + 2 * i) & 15) == 0) {
    for (; i + 4 <= n; i += 4) {
      const __m128 r = _mm_movelh_ps(_mm_cvtpd_ps(_mm_loadu_pd(re + i)), _mm_cvtpd_ps(_mm_loadu_pd(re + i + 2)));
      const __m128 q = _mm_movelh_ps(_mm_cvtpd_ps(_mm_loadu_pd(im + i)), _mm_cvtpd_ps(_mm_loadu_pd(im + i + 2)));
      _mm_stream_ps(dst + 2 * i, _mm_unpacklo_ps(r, q));
      _mm_stream_ps(dst + 2 * i + 4, _mm_unpackhi_ps(r, q));
    }
  }
  for (; i < n; ++i) { dst[2 * i] = (float)re[i]; dst[2 * i + 1] = (float)im[i]; }
}
inline void stage_fence() { _mm_sfence(); }       // non-temporal stores are globally visible before the copy is queued
#else
inline void round_doubles(float* dst, const double* src, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}
inline void round_interleave_doubles(float* dst, const double* re, const double* im, int64_t n) {
  for (int64_t i = 0; i < n; ++i) { dst[2 * i] = (float)re[i]; dst[2 * i + 1] = (float)im[i]; }
}
inline void stage_fence() {}
#endif

inline bool read_exact(int fd, char* buf, size_t n, int64_t off) {
  while (n > 0) {
    const ssize_t got = pread(fd, buf, n, (off_t)off);
    if (got < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    if (got == 0) { errno = EIO; return false; }       // the file ends inside the variable
    buf += got; n -= (size_t)got; off += got;
  }
  return true;
}

constexpr int64_t kFileBlockElems = 16384;

inline void stage_elems(char* dst, const Source& s, int64_t off, int64_t count, bool as_c128);

// the file form of stage_elems: blocks of the source through the calling thread's scratch
inline void stage_elems_file(char* dst, const Source& s, int64_t off, int64_t count, bool as_c128) {
  if (s.io_error != nullptr && s.io_error->load(std::memory_order_relaxed) != 0) return;   // the call has failed already
  bool ok = true;
  if (s.kind == kSrcC64) {
    ok = read_exact(s.fd, dst, (size_t)count * 8, s.re_off + off * 8);
  } else if (s.kind == kSrcC128 && as_c128) {
    ok = read_exact(s.fd, dst, (size_t)count * 16, s.re_off + off * 16);
  } else {
    thread_local std::vector<char> scratch;
    if (scratch.size() < (size_t)kFileBlockElems * 16) scratch.resize((size_t)kFileBlockElems * 16);
    char* const a = scratch.data();
    char* const b = a + kFileBlockElems * 8;
    const int64_t part = s.kind == kSrcF32Split ? 4 : 8;       // bytes per element of one split array
    Source mem;
    mem.kind = s.kind;
    mem.re = a;
    mem.im = (s.kind != kSrcC128 && s.im_off >= 0) ? b : nullptr;
    for (int64_t done = 0; done < count && ok; done += kFileBlockElems) {
      const int64_t n = count - done < kFileBlockElems ? count - done : kFileBlockElems;
      if (s.kind == kSrcC128) {
        ok = read_exact(s.fd, a, (size_t)n * 16, s.re_off + (off + done) * 16);
      } else {
        ok = read_exact(s.fd, a, (size_t)(n * part), s.re_off + (off + done) * part);
        if (ok && mem.im != nullptr) ok = read_exact(s.fd, b, (size_t)(n * part), s.im_off + (off + done) * part);
      }
      if (ok) stage_elems(dst + (size_t)done * 8, mem, 0, n, false);
    }
  }
  if (!ok && s.io_error != nullptr) {
    int expected = 0;
    s.io_error->compare_exchange_strong(expected, errno != 0 ? errno : EIO);
  }
}

// count elements starting at source element `off` -> dst.  as_c128: complex128 copied as it is
// (16 bytes per element, rounded later on the device); otherwise dst is complex64.
inline void stage_elems(char* dst, const Source& s, int64_t off, int64_t count, bool as_c128) {
  if (s.fd >= 0) {
    stage_elems_file(dst, s, off, count, as_c128);
    return;
  }
  float* const q = reinterpret_cast<float*>(dst);
  switch (s.kind) {
    case kSrcC64:
      memcpy(dst, s.re + off * 8, (size_t)count * 8);
      return;
    case kSrcC128:
      if (as_c128) memcpy(dst, s.re + off * 16, (size_t)count * 16);
      else round_doubles(q, reinterpret_cast<const double*>(s.re) + 2 * off, 2 * count);
      return;
    case kSrcF32Split: {
      const float* __restrict__ a = reinterpret_cast<const float*>(s.re) + off;
      if (s.im != nullptr) {
        const float* __restrict__ b = reinterpret_cast<const float*>(s.im) + off;
        for (int64_t i = 0; i < count; ++i) { q[2 * i] = a[i]; q[2 * i + 1] = b[i]; }
      } else {
        for (int64_t i = 0; i < count; ++i) { q[2 * i] = a[i]; q[2 * i + 1] = 0.f; }
      }
      return;
    }
    default: {
      const double* a = reinterpret_cast<const double*>(s.re) + off;
      if (s.im != nullptr) {
        round_interleave_doubles(q, a, reinterpret_cast<const double*>(s.im) + off, count);
      } else {
        for (int64_t i = 0; i < count; ++i) { q[2 * i] = (float)a[i]; q[2 * i + 1] = 0.f; }
      }
      return;
    }
  }
}

// runs [run0, run1) of the source, packed into dst, split over the pool at element granularity
inline void stage_runs(Pool& pool, char* dst, const Source& s, const RunMap& m, int64_t run0, int64_t run1,
                       bool as_c128) {
  const int64_t total = (run1 - run0) * m.run_len;
  if (total <= 0) return;
  const size_t esz = as_c128 ? 16 : 8;
  // parts of >= 256 KiB staged, about four per thread, so late wakers and slow cores even out
  int64_t parts = pool.size() > 1 ? (int64_t)pool.size() * 4 : 1;
  const int64_t min_elems = (int64_t)(256 * 1024 / esz);
  if (parts > (total + min_elems - 1) / min_elems) parts = (total + min_elems - 1) / min_elems;
  if (parts < 1) parts = 1;
  const int64_t per = ((total + parts - 1) / parts + 63) & ~int64_t(63);
  const std::function<void(int)> job = [&](int p) {
    int64_t e = (int64_t)p * per;
    const int64_t e1 = (e + per < total) ? e + per : total;
    while (e < e1) {
      const int64_t r = e / m.run_len, within = e - r * m.run_len;
      int64_t n = m.run_len - within;
      if (n > e1 - e) n = e1 - e;
      const int64_t from = m.offset(run0 + r) + within;
      // runs that follow each other in the source (the rows of a raw stream, the planes of an unpadded
      // container) are one copy -- for a file, one pread
      for (int64_t rr = r; e + n < e1 && m.offset(run0 + rr + 1) == m.offset(run0 + rr) + m.run_len; ++rr)
        n += (m.run_len < e1 - e - n) ? m.run_len : e1 - e - n;
      stage_elems(dst + (size_t)e * esz, s, from, n, as_c128);
      e += n;
    }
    stage_fence();
  };
  pool.run((int)((total + per - 1) / per), job);
}

}  // namespace amcx
