// Host -> HBM upload probe: what the real-data path (amcx_ctx_features18_planes_host) can be built on.
//   hipcc --offload-arch=gfx950 -O2 -o tools/h2d_probe tools/h2d_probe.hip -lpthread
// Prints GB/s for: pinned -> device by chunk size; pageable -> device (runtime staging);
// hipHostRegister cost and the copy rate from registered pages; T-thread memcpy pageable -> pinned;
// the staged pipeline (T threads filling two pinned slots while the copy engine drains the other).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t n, int T) {
  if (T <= 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = ((n + T - 1) / T + 4095) & ~size_t(4095);
  for (int t = 0; t < T; ++t) {
    const size_t a = (size_t)t * per;
    if (a >= n) break;
    const size_t len = n - a < per ? n - a : per;
    th.emplace_back([=] { memcpy(dst + a, src + a, len); });
  }
  for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
  const size_t total = (argc > 1 ? (size_t)atof(argv[1]) : 2048) << 20;   // MiB of pageable source
  char *d = nullptr, *pin = nullptr;
  CK(hipMalloc(&d, total));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t pin_bytes = 512u << 20;
  CK(hipHostMalloc(&pin, pin_bytes, hipHostMallocDefault));
  memset(pin, 1, pin_bytes);
  char* page = (char*)aligned_alloc(4096, total);
  for (size_t i = 0; i < total; i += 4096) page[i] = (char)i;             // touched: pages exist
  printf("cores seen %u; source %.0f MiB\n", std::thread::hardware_concurrency(), total / 1048576.0);

  for (size_t chunk : {1u << 20, 4u << 20, 16u << 20, 64u << 20, 256u << 20}) {
    CK(hipMemcpyAsync(d, pin, chunk, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    const int reps = (int)((1024u << 20) / chunk);
    double t0 = now();
    for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(d + (size_t)(r % 2) * chunk, pin, chunk, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    double dt = now() - t0;
    printf("pinned->dev   chunk %4zu MiB : %6.1f GB/s\n", chunk >> 20, reps * (double)chunk / dt / 1e9);
  }
  { double t0 = now(); CK(hipMemcpy(d, page, total, hipMemcpyHostToDevice)); double dt = now() - t0;
    printf("pageable->dev hipMemcpy %zu MiB : %6.1f GB/s\n", total >> 20, total / dt / 1e9);
    t0 = now(); CK(hipMemcpy(d, page, total, hipMemcpyHostToDevice)); dt = now() - t0;
    printf("pageable->dev again           : %6.1f GB/s\n", total / dt / 1e9); }
  for (size_t reg : {(size_t)32 << 20, (size_t)256 << 20, total}) {
    double t0 = now();
    hipError_t e = hipHostRegister(page, reg, hipHostRegisterDefault);
    double t_reg = now() - t0;
    if (e != hipSuccess) { printf("hipHostRegister(%zu MiB): %s\n", reg >> 20, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    t0 = now(); CK(hipMemcpyAsync(d, page, reg, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    double t_cp = now() - t0;
    t0 = now(); CK(hipHostUnregister(page)); double t_un = now() - t0;
    printf("register %5zu MiB: %.1f ms (%.1f GB/s), copy %.1f GB/s, unregister %.1f ms; register+copy+unregister %.1f GB/s\n",
           reg >> 20, t_reg * 1e3, reg / t_reg / 1e9, reg / t_cp / 1e9, t_un * 1e3, reg / (t_reg + t_cp + t_un) / 1e9);
  }
  for (int T : {1, 2, 4, 8, 12, 16}) {
    const size_t n = pin_bytes;
    par_copy(pin, page, n, T);
    double t0 = now();
    for (int r = 0; r < 4; ++r) par_copy(pin, page + (size_t)(r % 2) * n, n, T);
    double dt = now() - t0;
    printf("memcpy pageable->pinned %2d threads: %6.1f GB/s\n", T, 4.0 * n / dt / 1e9);
  }
  // staged pipeline: two pinned slots; threads fill slot k+1 while the copy engine drains slot k
  for (size_t slot : {(size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20})
    for (int T : {2, 4, 8, 12}) {
      hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
      double t0 = now();
      int k = 0;
      for (size_t off = 0; off < total; off += slot, ++k) {
        const size_t len = total - off < slot ? total - off : slot;
        if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
        par_copy(pin + (size_t)(k & 1) * slot, page + off, len, T);
        CK(hipMemcpyAsync(d + off, pin + (size_t)(k & 1) * slot, len, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[k & 1], s));
      }
      CK(hipStreamSynchronize(s));
      double dt = now() - t0;
      printf("staged pipeline slot %3zu MiB, %2d threads (spawned per slot): %6.1f GB/s\n", slot >> 20, T, total / dt / 1e9);
      CK(hipEventDestroy(ev[0])); CK(hipEventDestroy(ev[1]));
    }
  { // D2H of a 46 MB result
    const size_t n = 46u << 20;
    CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    double t0 = now();
    for (int r = 0; r < 10; ++r) CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    printf("dev->pinned 46 MiB: %6.1f GB/s\n", 10.0 * n / (now() - t0) / 1e9);
  }
  return 0;
}
