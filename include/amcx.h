/*
 * amcx.h -- C ABI of the MI355X IQ feature-extraction engine (libamcx.so).
 *
 * The reference (ronnymilleo/amcpy) is pure Python and has no FFI layer; the
 * seams this ABI replaces are plain Python call sites:
 *
 *   per frame : amcpy.features.calculate_features(feature_ids, signal)
 *               src/amcpy/features.py:214-232   (18 feature fns :66-185)
 *   per batch : the `_Worker` loop body
 *               feature_matrix[snr, frame, :] = calculate_features(1..18, signal)
 *               src/amcpy/feature_extraction.py:30-39, fed by :64-72
 *
 * A binding is a ctypes stub of ~15 lines; see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary.
 *   - IQ layout: row-major frames of interleaved (re, im) float32 == numpy
 *     complex64 == the reference's (n_snr, n_frames, >=frame_size) container
 *     flattened to [n_frames][row_stride_elems] (README.md:60-73;
 *     feature_extraction.py:68 takes the first frame_size samples of a row).
 *   - output: out[f * out_row_stride + j], j = 0..17 is feature id j+1 of
 *     frame f, float32 (feature_extraction.py:35,56).  Caller allocates both
 *     buffers and keeps them alive until the stream has drained; the library
 *     holds no persistent state and never frees caller memory.
 *   - every function returns 0 or a negative AMCX_E* code and never throws.
 *     NaN/Inf in the data are not errors: a frame holding a non-finite sample
 *     yields 18 NaNs, as numpy's arithmetic does for the reference.
 *   - amplitude range: any normal float32 sample, at every frame size (the squares of samples
 *     below ~3e-19 underflow; a frame of nothing but non-negative reals that small is taken
 *     for zeros).  The block kernel stages each frame times an exact power of two and
 *     un-scales in fp64.  The throughput kernel's fp32 sums hold inside 1e-5 <~ rms|x| <~ 1e5;
 *     a frame outside that (or any of whose sums overflows: a single 1e7 sample among unit
 *     ones) is re-run on a copy multiplied by an exact power of two by the throughput kernel itself
 *     -- at the frame sizes 1024 ... 4096 right behind the batch it was found in (128, 256 and 512 take EVERY frame times a
 *     power of two, as the block kernel does: nothing to re-run), at 8192 in a pass of
 *     the quad at the end of the launch, at 16384 / 32768 at the end of every epoch of 2048 frames of a
 *     workgroup (a data set that is out of range throughout, e.g. raw 24-bit ADC counts, runs at half
 *     the normal rate) -- so
 *     results match the reference, which evaluates in complex128 (features.py:46-58), including
 *     the inf / 0 / denormals its float32 store produces (feature_extraction.py:35,56).
 *   - AMCX_VARIANT_WAVE / AUTO at a power-of-two frame size 128 ... 32768 is ONE launch on the stream; when
 *     it has completed every row is final (ABI 3 as built in rounds 2-3 took two launches and left out-of-range
 *     frames marked in band in between, and so did N = 8192 until late in round 4).  Frames with a phase step
 *     within an fp32 ulp of +-pi are finished exactly inside every throughput kernel.
 *   - PARITY CONTRACT with the reference (the executable form is tests/test_gpu_parity.py): features
 *     1-9 and 11 within 1e-5 relative of the reference's complex128 evaluation stored as float32;
 *     features 10, 12-18 (cumulants whose terms cancel) within 1e-5 of max(|value|, S), S the sum of
 *     the magnitudes of the terms of that cumulant's formula (the reference's own complex64 path is
 *     up to 8.5e-3 off in plain relative terms there).  EVERY frame, whatever the size of the sample: a
 *     frame one of whose cumulants cancels below what fp32 sums resolve (S < kappa x the first-order
 *     error scale of that cumulant, kappa = 4e-3 (2048 / N)^(1/3); ~0.7 % of the BASELINE configs' frames)
 *     is found by the kernel's finaliser and gets its moment sums from an fp64 sweep in the same launch
 *     (amcx_math.h: cancellation_suspect; until ABI 6's round-6 build samples of thousands of frames were
 *     held to a floored S with 0.1 % of the frames excepted).
 *   - DEVICE OWNERSHIP: the entry points that take device pointers launch on the calling thread's
 *     CURRENT HIP device.  The buffers and the stream must belong to it: on a multi-GPU node call
 *     hipSetDevice(d) (torch.cuda.set_device / `with torch.cuda.device(d)`) first.  A pointer
 *     that lives on another device is refused with AMCX_EINVAL (checked with
 *     hipPointerGetAttributes; a stream cannot be checked: passing another device's stream is a
 *     HIP launch error, AMCX_EHIP).  The host-buffer entries take a device index and switch to it
 *     themselves for the duration of the call.
 *   - re-entrant and thread-safe; launches are asynchronous on `hip_stream`
 *     (a hipStream_t, NULL = default stream); completion = caller's stream sync.
 *     The device entry points allocate nothing and never synchronise, so after one
 *     warm call per kernel and device they can be captured in a HIP graph.
 */
#ifndef AMCX_H_
#define AMCX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version policy: the number goes up by one whenever entry points or constants are ADDED;
 * nothing that exists is ever removed or changes meaning under the same library name, so a binding
 * written against version v works with every library whose amcx_abi_version() >= v (and must
 * refuse an older one).  History: 1 = round 1-2 (features18, host / context entries, post kernels);
 * 2 = strided containers (amcx_ctx_features18_strided_host, amcx_stage_host, amcx_pack_planes_c64,
 * amcx_ctx_configure, amcx_ctx_upload_stats), device-ownership rule below made explicit;
 * 3 = single-read statistics with a caller workspace (amcx_group_stats_ws_f32,
 * amcx_group_stats_workspace_bytes), amcx_standardize_fit_transform_f32 / _workspace_bytes, and containers
 * read straight from their file (amcx_ctx_features18_strided_file, amcx_stage_file, AMCX_EIO);
 * 4 = host placement (amcx_numa_place, amcx_device_pci_bus_id, amcx_ctx_bind_cpus, amcx_ctx_placement: a context's
 * staging threads and pinned slots on the CPUs local to its device) and the instruction-issue ceiling probe
 * (amcx_probe_fma_rate);
 * 5 = frame sizes 16384 and 32768 (AMCX_MAX_FRAME_SIZE 32768, AMCX_MAX_BLOCK_FRAME_SIZE);
 * 6 = EVERY frame size 2 ... 32768: AMCX_MAX_BLOCK_FRAME_SIZE 32768 (8193 ... 32767 were AMCX_ENOTSUP but for 16384);
 * amcx_features18_c64_ws / amcx_features18_workspace_bytes. */
#define AMCX_ABI_VERSION 6
#define AMCX_NUM_FEATURES 18

/* error codes */
#define AMCX_OK 0
#define AMCX_EINVAL (-1)   /* null pointer, negative count, bad stride, frame_size out of range */
#define AMCX_ENOTSUP (-2)  /* requested kernel variant cannot handle this frame_size */
#define AMCX_EHIP (-3)     /* HIP runtime / launch failure (see amcx_last_hip_error) */
#define AMCX_ENODEV (-4)   /* no usable gfx950 device */
#define AMCX_ENOMEM (-5)   /* device allocation failed (host-buffer entry point only) */
#define AMCX_EIO (-6)      /* a read from the container's file failed or the file ends inside the variable
                              (the *_file entries; amcx_last_hip_error() holds the errno text) */

/* kernel variants (amcx_features18_c64_ex) */
#define AMCX_VARIANT_AUTO 0     /* fastest kernel that supports frame_size */
#define AMCX_VARIANT_BLOCK 1    /* one 256-thread workgroup per frame, frame staged in LDS,
                                   radix-2 LDS FFT (power of two), Bluestein chirp-z FFT (every
                                   other N >= 65: both spectra in LDS up to 4096; 4097..8191 a
                                   16384-point convolution with the chirp's spectrum held in
                                   registers) or direct O(N^2) fp64 DFT (N <= 64); fp64
                                   accumulation.  Above 8192 samples (ABI 6) one 1024-thread
                                   workgroup per frame, the frame read where it lies, the phase in
                                   LDS, the spectral peak by a chirp-z / plain FFT run in place on a
                                   global workspace (amcx_features18_c64_ws below) or, without one,
                                   as the DFT by its definition with exact twiddle indices, O(N^2);
                                   2 <= frame_size <= AMCX_MAX_BLOCK_FRAME_SIZE */
#define AMCX_VARIANT_WAVE 2     /* the throughput kernels: the frame held in registers, register
                                   FFT with LDS exchanges, fp32 sums with an fp64 finaliser, frames
                                   outside the fp32 range re-run inside the launch; frame_size a
                                   power of two, 128 ... 32768 (1024 ... 4096: one wavefront per
                                   frame; 128 ... 512: four frames per wavefront; 8192: four waves
                                   per frame, 16384 / 32768: eight / sixteen) */

#define AMCX_MIN_FRAME_SIZE 2
#define AMCX_MAX_FRAME_SIZE 32768       /* the reference takes any frame_size (config.py:96; np.fft.fft, features.py:68); this
                                           library: every size 2 ... 32768 (ABI 6; ABI 5: 2 ... 8192 and the powers of two
                                           16384, 32768).  Larger: AMCX_EINVAL. */
#define AMCX_MAX_BLOCK_FRAME_SIZE 32768 /* AMCX_VARIANT_BLOCK (and with it every size that is not a power of two) ends here */

int amcx_abi_version(void);
const char* amcx_strerror(int code);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char* amcx_last_hip_error(void);
/* number of visible devices whose architecture is gfx950; <0 on error */
int amcx_device_count(void);

/*
 * All 18 features of n_frames frames, device buffers.
 * Replaces the `_Worker.run` loop body, feature_extraction.py:30-39, for a
 * whole (n_snr * n_frames) block at once.
 *   iq_dev            device pointer, complex64 [n_frames][row_stride_elems]
 *   row_stride_elems  distance between frame starts in complex elements, >= frame_size
 *   out_dev           device pointer, float32 [n_frames][out_row_stride]
 *   out_row_stride    in floats, >= 18
 *   hip_stream        hipStream_t or NULL
 * n_frames == 0 is a valid no-op.
 */
int amcx_features18_c64(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                        int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                        void* hip_stream);

/* Same, with an explicit kernel variant (tests and benchmarks). */
int amcx_features18_c64_ex(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                           int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                           void* hip_stream, int32_t variant);

/*
 * ABI 6.  The any-size path above 8192 samples (AMCX_VARIANT_BLOCK there; AUTO where frame_size is not a power of
 * two) has two forms of its spectral term: an FFT through a device workspace -- Bluestein's chirp-z as a 32768- /
 * 65536-point circular convolution run in place on one private buffer per workgroup, the plain transform at 16384 /
 * 32768 -- and, without one, the DFT by its definition (O(N^2), ~100x slower at 32767; same results within the parity
 * contract).  amcx_features18_c64 / _ex take the workspace from the stream-ordered allocator (hipMallocAsync /
 * hipFreeAsync on hip_stream), except while the stream is being captured into a graph or when the allocator fails:
 * then they run the form that needs none (AMCX_VERBOSE=1 in the environment: one line on stderr, once per process,
 * when that happens -- the kernel's name and the results do not show which form ran; a caller whose device memory is
 * owned by another allocator, e.g. torch's caching one, should hand in the workspace itself).
 * amcx_features18_c64_ws takes the caller's: at least
 * amcx_features18_workspace_bytes(frame_size, n_frames, variant) bytes of device memory (contents undefined before,
 * garbage after; 8-byte aligned; on the current device) for the full number of frames in flight, fewer bytes mean fewer
 * workgroups, less than one workgroup's share (or NULL) the workspace-free form.  _workspace_bytes is 0 for every
 * frame size and variant that needs none (all of 2 ... 8192, and the WAVE kernels), -1 for arguments the call itself
 * would refuse.  Every other frame size: _ws behaves exactly as _ex and ignores the workspace.
 */
int64_t amcx_features18_workspace_bytes(int32_t frame_size, int64_t n_frames, int32_t variant);
int amcx_features18_c64_ws(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                           int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                           void* hip_stream, int32_t variant, void* workspace_dev, int64_t workspace_bytes);

/*
 * Same computation for HOST buffers (numpy arrays): allocates device scratch,
 * copies in, runs the kernel on `device`, copies the (n_frames x 18) result
 * back and returns when it is in `out_host`.  Replaces a direct
 * calculate_features(...) call, features.py:214-232, without torch.
 * This is NOT a CPU implementation: it fails with AMCX_ENODEV without a GPU.
 */
int amcx_features18_c64_host(const void* iq_host, int64_t n_frames, int32_t frame_size,
                             int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                             int32_t device, int32_t variant);

/*
 * Same for a HOST container of complex128 (MATLAB doubles, what scipy.io.loadmat
 * returns for the reference's all_modulations.mat, feature_extraction.py:46-48):
 * rows are rounded to complex64 (round-to-nearest-even, identical to numpy's astype and
 * to the GPU's conversion) by the staging copy into pinned memory that has to happen
 * anyway, so PCIe carries 8 B/sample (amcx_ctx_configure(..., round_on_device = 1) sends
 * the doubles and rounds on the GPU instead).  row_stride_elems is in complex128 elements.
 */
int amcx_features18_c128_host(const void* iq_host, int64_t n_frames, int32_t frame_size,
                              int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                              int32_t device, int32_t variant);

/*
 * The same two host-buffer entry points over a reusable context: the context owns a
 * HIP stream and device scratch that only grows, so calling per frame in a loop -- the
 * reference's own usage pattern, calculate_features once per queue item
 * (feature_extraction.py:30-39) -- costs two small copies and the launches instead of
 * hipMalloc / hipFree / stream creation per call.  Small row-major calls (one chunk, under
 * 1 MiB) are replayed as a HIP GRAPH: copy in, conversion, kernels and copy out are captured
 * once per shape (frames, frame size, variant, element type; four shapes cached) and a call
 * is one hipGraphLaunch and one synchronisation (39 us per 2048-sample frame against 47 us as
 * separate runtime calls); AMCX_NO_GRAPH=1 in the environment turns that off.  A context belongs to one device and
 * must not be used by two threads at once (one context per thread).  amcx_ctx_destroy
 * accepts NULL.  The one-shot entry points above are these with a context that lives
 * for the duration of the call.
 */
typedef struct amcx_ctx amcx_ctx;
int amcx_ctx_create(int32_t device, amcx_ctx** ctx_out);
int amcx_ctx_destroy(amcx_ctx* ctx);
int amcx_ctx_features18_c64_host(amcx_ctx* ctx, const void* iq_host, int64_t n_frames,
                                 int32_t frame_size, int64_t row_stride_elems, float* out_host,
                                 int64_t out_row_stride, int32_t variant);
int amcx_ctx_features18_c128_host(amcx_ctx* ctx, const void* iq_host, int64_t n_frames,
                                  int32_t frame_size, int64_t row_stride_elems, float* out_host,
                                  int64_t out_row_stride, int32_t variant);

/*
 * Name of the kernel `variant` resolves to for this frame_size (as it shows in
 * rocprofv3 kernel traces), written NUL-terminated into buf.  Returns 0, or
 * AMCX_ENOTSUP / AMCX_EINVAL.
 */
int amcx_kernel_name(int32_t frame_size, int32_t variant, char* buf, int32_t buf_len);

/*
 * The real-data path: all 18 features of every frame of a HOST container indexed
 * [snr][frame][sample] with ARBITRARY element strides -- what the reference slices frame by frame
 * out of the array scipy.io.loadmat returned (feature_extraction.py:46-48,64-72: Fortran-ordered,
 * complex128, rows longer than frame_size).  Frame g = snr * n_frames + frame, the order the
 * reference enqueues them in; out_host[g * out_row_stride + j] = feature j + 1.
 *
 *   kind      AMCX_SRC_C64 / AMCX_SRC_C128: `re` points at interleaved (re, im) float32 / float64,
 *             `im` is ignored.  AMCX_SRC_F32_SPLIT / AMCX_SRC_F64_SPLIT: `re` and `im` are two
 *             separate real arrays with the same strides (how a MATLAB v5 file stores a complex
 *             variable; im == NULL: a real signal).  Doubles are rounded to float32 to nearest even
 *             (== numpy astype), by the staging threads on their way to pinned memory.
 *   strides   in ELEMENTS of the source (complex elements for the interleaved kinds), all >= 0.
 *             One of them must be 1:
 *               stride_sample == 1 (C order)        rows go up in chunks of whole frames and the
 *                                                   kernel runs on chunk k while chunk k+1 uploads;
 *               stride_snr == 1 or stride_frame == 1 (Fortran order, as loadmat returns it)
 *                                                   contiguous sample planes go up as they lie and
 *                                                   a device kernel transposes them to frame-major
 *                                                   (amcx_pack_planes_c64); the feature kernel runs
 *                                                   once, after the last plane.
 *             Otherwise AMCX_ENOTSUP (make a contiguous copy first).
 *
 * Pageable memory is fine (that is the point): a pool of host threads copies runs into three pinned
 * slots while the copy engine drains them -- ~55 GB/s of PCIe, i.e. ~110 GB/s of complex128 source.
 * Blocks until out_host is complete.  One context per thread.
 */
#define AMCX_SRC_C64 0
#define AMCX_SRC_C128 1
#define AMCX_SRC_F32_SPLIT 2
#define AMCX_SRC_F64_SPLIT 3
int amcx_ctx_features18_strided_host(amcx_ctx* ctx, const void* re, const void* im, int32_t kind,
                                     int64_t n_snr, int64_t n_frames, int32_t frame_size,
                                     int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                                     float* out_host, int64_t out_row_stride, int32_t variant);

/*
 * The host half on its own (no device involved; works without a GPU): stage `n_units` units of the
 * container, starting at `first_unit`, into `dst` as packed complex64, with `threads` host threads.
 * A unit is a frame (row-major containers: dst[unit][sample]) or a sample plane (plane-major:
 * dst[unit][position], position order as amcx_pack_planes_c64's inner_snr says); *plane_major and
 * *inner_snr report which (either may be NULL).  For callers that run their own copy engine -- and
 * how the staging and rounding logic is tested where there is no GPU.
 */
int amcx_stage_host(const void* re, const void* im, int32_t kind, int64_t n_snr, int64_t n_frames,
                    int32_t frame_size, int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                    int64_t first_unit, int64_t n_units, void* dst, int64_t dst_bytes, int32_t threads,
                    int32_t* plane_major, int32_t* inner_snr);

/*
 * The same two entries for a container that is still in its FILE: fd is an open descriptor (the library
 * neither closes nor seeks it: every read is a pread), re_offset / im_offset the byte offsets of the arrays
 * that `re` / `im` would point to (im_offset < 0: no imaginary part; ignored for the interleaved kinds);
 * strides stay in elements.  A MATLAB level-5 variable that is not compressed is exactly this: two column-major
 * arrays at fixed offsets (amcpy_amd/matfile.py locates them), so all_modulations.mat goes from the page cache
 * to the pinned slots through 256 KB of per-thread scratch -- without the page faults of a mapping (30 ms per
 * 436 MB variable however many threads fault) and without a host copy of the variable.  Replaces
 * scipy.io.loadmat + the slicing of feature_extraction.py:46-48,64-72.  AMCX_EIO if a read fails or the file
 * is shorter than the strides say (nothing is left in flight; the output is not written).
 */
int amcx_ctx_features18_strided_file(amcx_ctx* ctx, int32_t fd, int64_t re_offset, int64_t im_offset,
                                     int32_t kind, int64_t n_snr, int64_t n_frames, int32_t frame_size,
                                     int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                                     float* out_host, int64_t out_row_stride, int32_t variant);
int amcx_stage_file(int32_t fd, int64_t re_offset, int64_t im_offset, int32_t kind, int64_t n_snr,
                    int64_t n_frames, int32_t frame_size, int64_t stride_snr, int64_t stride_frame,
                    int64_t stride_sample, int64_t first_unit, int64_t n_units, void* dst, int64_t dst_bytes,
                    int32_t threads, int32_t* plane_major, int32_t* inner_snr);

/*
 * Tuning of a context's upload path: threads = staging threads including the caller's (0 = keep;
 * default min(8, hardware threads); the reference's SignalConfig.num_threads maps here),
 * slot_bytes = size of one pinned staging slot (0 = keep; default 32 MiB; three are allocated),
 * round_on_device != 0: complex128 goes over PCIe as it is and is rounded by the device kernel
 * (twice the link bytes, no host arithmetic; -1 = keep).
 */
int amcx_ctx_configure(amcx_ctx* ctx, int32_t threads, int64_t slot_bytes, int32_t round_on_device);

/* what the last strided call of this context moved, and where its host time went */
typedef struct amcx_upload_stats {
  int64_t frames;
  int64_t source_bytes;     /* bytes of the container that were read */
  int64_t pcie_bytes;       /* bytes that crossed the link host -> device */
  int32_t chunks;
  int32_t threads;
  int32_t plane_major;      /* 1: planes + device transposition, 0: rows */
  int32_t from_file;       /* 1: the source was a file read by the staging threads (version 3; `reserved`, always 0, before) */
  double seconds;           /* whole call */
  double seconds_staging;   /* caller thread inside the staging copies */
  double seconds_waiting;   /* caller thread blocked on a pinned slot still being uploaded */
  double seconds_prepare;   /* before the first chunk: (re)allocation of slots and scratch */
  double seconds_tail;      /* after the last chunk was queued: feature kernel, result copy, sync */
} amcx_upload_stats;
int amcx_ctx_upload_stats(const amcx_ctx* ctx, amcx_upload_stats* out);

/*
 * The device half of the plane-major path, for callers that do their own uploads: slab_dev holds
 * n_planes sample planes, plane r = sample n0 + r of every position j < n_snr * n_frames at
 * slab_dev[r * plane_stride + j] (complex64 if src_kind == AMCX_SRC_C64, complex128 rounded here if
 * AMCX_SRC_C128).  inner_snr != 0: j = frame * n_snr + snr (Fortran order); 0: j = snr * n_frames +
 * frame.  Writes frames_dev[g * row_stride_elems + n0 + r], g = snr * n_frames + frame.
 */
int amcx_pack_planes_c64(const void* slab_dev, int32_t src_kind, int32_t n_planes, int64_t plane_stride,
                         int64_t n_snr, int64_t n_frames, int32_t inner_snr, void* frames_dev,
                         int64_t row_stride_elems, int32_t n0, void* hip_stream);

/*
 * Streaming-read ceiling probe: sums n_bytes of device memory with 16-byte
 * loads (one float per workgroup written to partial_dev, >= 4096 floats).
 * Used by bench.py to report the measured HBM read ceiling next to the
 * nominal 8 TB/s.  n_bytes must be a multiple of 16.
 */
int amcx_probe_read_bw(const void* src_dev, int64_t n_bytes, float* partial_dev,
                       void* hip_stream);

/*
 * Instruction-issue ceiling probe -- the bound that binds this path on an MI355X (the board's power cap, then VALU
 * issue: DESIGN.md section 4): one 16-wavefront workgroup per CU (four waves per SIMD, the N = 2048 kernel's
 * occupancy) runs independent v_fma_f32 on registers, no memory traffic, back to back for `seconds` (0 < seconds <=
 * 60) on hip_stream; the first half lets the clock settle, the second half is timed with HIP events.
 * *wave_instr_per_s: wavefront-instructions per second of the whole device; *clock_ghz (may be NULL): the shader
 * clock during the last launch (s_memtime cycles / s_memrealtime ticks).  Unlike the other device entries this one
 * allocates scratch and SYNCHRONISES the stream.  bench.py reports the feature kernel's own instruction rate against
 * it (roofline.secondary.measured).
 */
int amcx_probe_fma_rate(double seconds, void* hip_stream, double* wave_instr_per_s, double* clock_ghz);

/*
 * Host placement.  One process may drive every GPU of a node (the reference starts one process per modulation,
 * feature_extraction.py:89-97; here amcpy_amd.feature_extraction.DeviceFanOut runs one context per device), and a
 * context's staging threads and pinned slots belong on the socket its device hangs off.
 *
 * amcx_device_pci_bus_id: "dddd:bb:dd.f" (lower case) of visible device `device`; buf_len >= 16.
 * amcx_numa_place: host-only (no GPU needed).  Reads <sysfs_root>/bus/pci/devices/<pci_bus_id>/numa_node and
 *   .../local_cpulist (sysfs_root NULL or "" = "/sys").  *node_out = the node or -1, *n_cpus_out = how many CPUs
 *   are local (0 with node -1), the first min(n, cpus_cap) of them in cpus_out (ascending).  A platform that does
 *   not say (numa_node -1, a missing file, a malformed list) is not an error: node -1, no CPUs, nothing gets bound.
 * amcx_ctx_create does this for its device by itself (environment: AMCX_NUMA=0 turns it off, AMCX_SYSFS_ROOT names
 *   another tree); amcx_ctx_bind_cpus replaces the choice (n_cpus = 0: bind nothing; AMCX_EINVAL while a call is
 *   running on the context: bind between calls).  Bound are: the context's
 *   staging threads, and the CALLING thread for the duration of an upload large enough to use them (its own
 *   affinity mask is restored on return) -- so the pinned slots, allocated on first use inside such a call, are placed
 *   on that node too.  A binding never widens the mask the process was given (cpusets, taskset): CPUs outside it
 *   are dropped, and if none is left nothing is bound.  A thread that the CALLER's thread creates while such an upload
 *   runs (from another thread of the same Python process, say) does not inherit the narrowed mask -- only threads created
 *   BY the calling thread inside that window would, and the library creates none there: its staging threads exist before,
 *   and amcx_ctx_create warms the HIP runtime (first pinned allocation, first stream copy) so that the runtime's own
 *   helper threads are started under the creator's mask, not inside the window.
 * amcx_ctx_placement: what a context did.
 */
typedef struct amcx_placement {
  int32_t device;
  int32_t numa_node;        /* -1: unknown / unbound */
  int32_t n_cpus;           /* CPUs local to the device (or given to amcx_ctx_bind_cpus) */
  int32_t n_cpus_allowed;   /* ... of which the calling thread may use */
  int32_t first_cpu, last_cpu;
  char pci_bus_id[32];
} amcx_placement;
int amcx_device_pci_bus_id(int32_t device, char* buf, int32_t buf_len);
int amcx_numa_place(const char* sysfs_root, const char* pci_bus_id, int32_t* node_out, int32_t* cpus_out,
                    int32_t cpus_cap, int32_t* n_cpus_out);
int amcx_ctx_bind_cpus(amcx_ctx* ctx, const int32_t* cpus, int32_t n_cpus);
int amcx_ctx_placement(const amcx_ctx* ctx, amcx_placement* out);

/*
 * Consumers directly behind the path, on the device-resident
 * (rows x >=18) float32 feature matrix.
 *
 * amcx_group_stats_f32: for each of n_groups consecutive blocks of rows_per_group
 * rows, column mean and POPULATION standard deviation (ddof = 0) of the first n_cols
 * (<= 32) columns, fp64: mean_dev / std_dev are [n_groups][n_cols] doubles.
 * Replaces the per-SNR np.mean / np.std triple loop, graphics.py:50-62, and the fit
 * of StandardScaler, preprocessing.py:59-61.  The matrix is read once: every group is
 * cut into chunks (one workgroup each, ~2048 in all), a chunk's (n, sum, M2) is formed
 * from register-held tiles about each tile's own mean, and a second small launch merges
 * the chunks pairwise (Chan, Golub & LeVeque); an inf in a column gives mean +-inf and
 * std nan, as numpy's two passes do.  The chunk triples go through a workspace:
 * amcx_group_stats_f32 takes it from the stream-ordered allocator (hipMallocAsync /
 * hipFreeAsync on hip_stream: no synchronisation, but not capturable in a graph on
 * every runtime); amcx_group_stats_ws_f32 takes the caller's (at least
 * amcx_group_stats_workspace_bytes(...) bytes of device memory, contents undefined
 * afterwards; the size depends on the three arguments only, -1 for invalid ones).
 */
int amcx_group_stats_f32(const float* x_dev, int64_t n_groups, int64_t rows_per_group,
                         int64_t row_stride, int32_t n_cols, double* mean_dev, double* std_dev,
                         void* hip_stream);
int64_t amcx_group_stats_workspace_bytes(int64_t n_groups, int64_t rows_per_group, int32_t n_cols);
int amcx_group_stats_ws_f32(const float* x_dev, int64_t n_groups, int64_t rows_per_group,
                            int64_t row_stride, int32_t n_cols, double* mean_dev, double* std_dev,
                            void* workspace_dev, int64_t workspace_bytes, void* hip_stream);

/*
 * amcx_select_scale_f32: out[r][j] = (x[r][cols[j]] - mean[j]) / scale[j], rounded to
 * float32 after each of the two operations as numpy's in-place float32 arithmetic does
 * (column pick preprocessing.py:55, StandardScaler.transform :62).  cols_dev: n_sel
 * int32 column indices (0-based, as the reference passes list(FeatureConfig.used)).
 */
int amcx_select_scale_f32(const float* x_dev, int64_t n_rows, int64_t row_stride,
                          const int32_t* cols_dev, int32_t n_sel, const double* mean_dev,
                          const double* scale_dev, float* out_dev, int64_t out_stride,
                          void* hip_stream);

/*
 * amcx_standardize_fit_transform_f32: StandardScaler().fit_transform(X[:, cols]) of
 * preprocessing.py:52-62 as three launches on hip_stream and nothing else -- no host
 * round trip, no synchronisation: statistics of all n_cols (<= 32) columns over the
 * n_rows rows (the two launches above, one group), then one launch that picks the
 * n_sel (<= 32) columns cols_host[] (HOST array of 0-based indices, passed by value),
 * turns their std into sklearn's scale_ (a column that the two-pass error bound cannot
 * tell from constant gets 1: var <= n eps var + (n mean eps)^2, sklearn >= 0.24
 * _is_constant_feature), writes mean_dev[n_sel] / scale_dev[n_sel] (doubles) and
 * out_dev[r][j] with the two float32 roundings of amcx_select_scale_f32.
 * workspace_dev: at least amcx_standardize_workspace_bytes(n_rows, n_cols) bytes.
 */
int64_t amcx_standardize_workspace_bytes(int64_t n_rows, int32_t n_cols);
int amcx_standardize_fit_transform_f32(const float* x_dev, int64_t n_rows, int64_t row_stride,
                                       int32_t n_cols, const int32_t* cols_host, int32_t n_sel,
                                       float* out_dev, int64_t out_stride, double* mean_dev,
                                       double* scale_dev, void* workspace_dev, int64_t workspace_bytes,
                                       void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* AMCX_H_ */
