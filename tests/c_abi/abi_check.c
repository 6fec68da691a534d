/* Plain-C client of include/amcx.h: what a non-Python caller of the boundary looks like.
 *   gcc -std=c99 -Iinclude tests/c_abi/abi_check.c -Lamcpy_amd/lib -lamcx -Wl,-rpath,$PWD/amcpy_amd/lib -lm -o abi_check
 *   ./abi_check            argument / error-code checks only: needs no GPU
 *   ./abi_check compute N FILE   18 features of the N-sample complex64 frame in FILE (raw interleaved float32)
 *                          through the host-buffer entry and through a reusable context (the two must agree
 *                          bit for bit); prints them, one per line
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amcx.h"

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { fprintf(stderr, "abi_check: %s failed (line %d)\n", #cond, __LINE__); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  CHECK(amcx_abi_version() == AMCX_ABI_VERSION);
  CHECK(strcmp(amcx_strerror(AMCX_OK), "ok") == 0);
  CHECK(amcx_strerror(-99) != NULL && amcx_last_hip_error() != NULL);
  /* validation happens before any device is touched */
  CHECK(amcx_features18_c64(NULL, -1, 2048, 2048, NULL, 18, NULL) == AMCX_EINVAL);
  CHECK(amcx_features18_c64(NULL, 4, 2048, 100, NULL, 18, NULL) == AMCX_EINVAL);
  CHECK(amcx_features18_c64(NULL, 4, 2048, 2048, NULL, 17, NULL) == AMCX_EINVAL);
  CHECK(amcx_features18_c64(NULL, 0, 2048, 2048, NULL, 18, NULL) == AMCX_OK);
  CHECK(amcx_features18_c64_ex(NULL, 0, 1000, 1000, NULL, 18, NULL, AMCX_VARIANT_WAVE) == AMCX_ENOTSUP);
  CHECK(amcx_features18_c64_host(NULL, 3, 2048, 2048, NULL, 18, 0, AMCX_VARIANT_AUTO) == AMCX_EINVAL);
  CHECK(amcx_ctx_create(0, NULL) == AMCX_EINVAL);
  CHECK(amcx_ctx_destroy(NULL) == AMCX_OK);
  char name[96];
  CHECK(amcx_kernel_name(2048, AMCX_VARIANT_AUTO, name, (int32_t)sizeof name) == AMCX_OK);
  CHECK(strstr(name, "wave_kernel<2048>") != NULL);
  CHECK(amcx_kernel_name(1000, AMCX_VARIANT_AUTO, name, (int32_t)sizeof name) == AMCX_OK);
  CHECK(strstr(name, "block_kernel") != NULL);
  /* ABI 6: every frame size up to 32768; the workspace the any-size path's FFT form wants */
  CHECK(amcx_kernel_name(12345, AMCX_VARIANT_AUTO, name, (int32_t)sizeof name) == AMCX_OK);
  CHECK(strstr(name, "stream_kernel") != NULL);
  CHECK(amcx_features18_workspace_bytes(2048, 1000, AMCX_VARIANT_AUTO) == 0);
  CHECK(amcx_features18_workspace_bytes(12345, 2, AMCX_VARIANT_AUTO) == 3 * 32768 * 8);
  CHECK(amcx_features18_workspace_bytes(32769, 2, AMCX_VARIANT_AUTO) == -1);
  CHECK(amcx_features18_c64_ws(NULL, 0, 12345, 12345, NULL, 18, NULL, AMCX_VARIANT_AUTO, NULL, 0) == AMCX_OK);
  CHECK(amcx_features18_c64_ws(NULL, 2, 12345, 12345, NULL, 18, NULL, AMCX_VARIANT_AUTO, NULL, 0) == AMCX_EINVAL);
  /* version-2 additions: the strided-container entries validate before touching a device, and the staging half
   * works without one */
  CHECK(amcx_ctx_features18_strided_host(NULL, NULL, NULL, AMCX_SRC_C64, 1, 1, 64, 64, 64, 1, NULL, 18, AMCX_VARIANT_AUTO) == AMCX_EINVAL);
  CHECK(amcx_ctx_configure(NULL, 1, 0, -1) == AMCX_EINVAL);
  CHECK(amcx_pack_planes_c64(NULL, AMCX_SRC_C64, -1, 0, 1, 1, 1, NULL, 1, 0, NULL) == AMCX_EINVAL);
  CHECK(amcx_pack_planes_c64(NULL, AMCX_SRC_C64, 0, 4, 2, 2, 1, NULL, 8, 0, NULL) == AMCX_OK);
  {
    /* a 2 x 3 x 4 container in Fortran order (snr fastest), doubles: planes come out position-major, rounded */
    double src[2 * 2 * 3 * 4];
    float staged[2 * 6 * 2];
    int32_t plane_major = -1, inner_snr = -1;
    for (int i = 0; i < 2 * 2 * 3 * 4; ++i) src[i] = 0.1 * i + 1e-9;
    CHECK(amcx_stage_host(src, NULL, AMCX_SRC_C128, 2, 3, 4, 1, 2, 6, 1, 2, staged, sizeof staged, 2,
                          &plane_major, &inner_snr) == AMCX_OK);
    CHECK(plane_major == 1 && inner_snr == 1);
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 6; ++j) {
        CHECK(staged[2 * (r * 6 + j)] == (float)src[2 * ((1 + r) * 6 + j)]);
        CHECK(staged[2 * (r * 6 + j) + 1] == (float)src[2 * ((1 + r) * 6 + j) + 1]);
      }
    CHECK(amcx_stage_host(src, NULL, AMCX_SRC_C128, 2, 3, 4, 2, 4, 12, 0, 1, staged, sizeof staged, 1, NULL, NULL) == AMCX_ENOTSUP);
    CHECK(amcx_stage_host(src, NULL, AMCX_SRC_C128, 2, 3, 4, 1, 2, 6, 3, 2, staged, sizeof staged, 1, NULL, NULL) == AMCX_EINVAL);
  }
  /* version-3 additions: workspace sizes are pure functions of their arguments */
  CHECK(amcx_group_stats_workspace_bytes(156, 4096, 18) > 0);
  CHECK(amcx_group_stats_workspace_bytes(1, 0, 18) == -1 && amcx_group_stats_workspace_bytes(1, 4, 33) == -1);
  CHECK(amcx_group_stats_workspace_bytes(0, 4, 18) == 0);
  CHECK(amcx_standardize_workspace_bytes(638976, 18) > amcx_group_stats_workspace_bytes(1, 638976, 18));
  CHECK(amcx_group_stats_ws_f32(NULL, 1, 4, 18, 18, NULL, NULL, NULL, 0, NULL) == AMCX_EINVAL);
  {
    const int32_t cols[2] = {2, 18};
    CHECK(amcx_standardize_fit_transform_f32(NULL, 4, 18, 18, cols, 2, NULL, 2, NULL, NULL, NULL, 0, NULL) == AMCX_EINVAL);
    CHECK(amcx_standardize_fit_transform_f32(NULL, 0, 18, 18, cols, 1, NULL, 1, NULL, NULL, NULL, 0, NULL) == AMCX_OK);
  }
  /* version-4 additions: the placement mapper is host-only (a device that is not in the tree binds nothing), the
   * others validate before touching a device */
  {
    int32_t node = 7, n = 7, cpus[4];
    CHECK(amcx_numa_place("/nonexistent-sysfs", "0000:05:00.0", &node, cpus, 4, &n) == AMCX_OK && node == -1 && n == 0);
    CHECK(amcx_numa_place(NULL, NULL, &node, cpus, 4, &n) == AMCX_EINVAL);
    CHECK(amcx_numa_place(NULL, "0000:05:00.0", &node, NULL, 4, &n) == AMCX_EINVAL);
    CHECK(amcx_ctx_bind_cpus(NULL, NULL, 0) == AMCX_EINVAL && amcx_ctx_placement(NULL, NULL) == AMCX_EINVAL);
    double rate = -1.0;
    CHECK(amcx_probe_fma_rate(0.0, NULL, &rate, NULL) == AMCX_EINVAL && amcx_probe_fma_rate(1.0, NULL, NULL, NULL) == AMCX_EINVAL);
    char bus[8];
    CHECK(amcx_device_pci_bus_id(0, bus, (int32_t)sizeof bus) == AMCX_EINVAL);
  }
  if (argc < 4 || strcmp(argv[1], "compute") != 0) {
    /* without a GPU the host-buffer entries must refuse, not compute */
    if (amcx_device_count() <= 0) {
      float x[2 * 64] = {1.0f}, out[18];
      amcx_ctx* c = NULL;
      CHECK(amcx_features18_c64_host(x, 1, 64, 64, out, 18, 0, AMCX_VARIANT_AUTO) == AMCX_ENODEV);
      CHECK(amcx_ctx_create(0, &c) == AMCX_ENODEV && c == NULL);
    }
    printf("abi_check ok\n");
    return 0;
  }
  const int n = atoi(argv[2]);
  CHECK(n >= AMCX_MIN_FRAME_SIZE && n <= AMCX_MAX_FRAME_SIZE);
  float* x = (float*)malloc(sizeof(float) * 2 * (size_t)n);
  CHECK(x != NULL);
  FILE* fh = fopen(argv[3], "rb");
  CHECK(fh != NULL);
  CHECK(fread(x, sizeof(float) * 2, (size_t)n, fh) == (size_t)n);
  fclose(fh);
  float once[18], looped[18];
  int rc = amcx_features18_c64_host(x, 1, n, n, once, 18, 0, AMCX_VARIANT_AUTO);
  if (rc != AMCX_OK) { fprintf(stderr, "amcx_features18_c64_host: %s %s\n", amcx_strerror(rc), amcx_last_hip_error()); return 2; }
  amcx_ctx* ctx = NULL;
  CHECK(amcx_ctx_create(0, &ctx) == AMCX_OK && ctx != NULL);
  for (int rep = 0; rep < 3; ++rep)
    CHECK(amcx_ctx_features18_c64_host(ctx, x, 1, n, n, looped, 18, AMCX_VARIANT_AUTO) == AMCX_OK);
  /* the same frame as a 1 x 1 x n container with a unit sample stride: the row path of the strided entry */
  float strided[18];
  amcx_upload_stats st;
  CHECK(amcx_ctx_configure(ctx, 2, 1 << 20, -1) == AMCX_OK);
  CHECK(amcx_ctx_features18_strided_host(ctx, x, NULL, AMCX_SRC_C64, 1, 1, n, 0, n, 1, strided, 18, AMCX_VARIANT_AUTO) == AMCX_OK);
  CHECK(amcx_ctx_upload_stats(ctx, &st) == AMCX_OK && st.frames == 1 && st.pcie_bytes == 8 * (int64_t)n && st.plane_major == 0);
  CHECK(memcmp(once, strided, sizeof once) == 0);
  CHECK(amcx_ctx_destroy(ctx) == AMCX_OK);
  CHECK(memcmp(once, looped, sizeof once) == 0);
  for (int j = 0; j < 18; ++j) printf("%.9g\n", once[j]);
  free(x);
  return 0;
}
