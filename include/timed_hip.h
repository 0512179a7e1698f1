/*
 * timed_hip.h — C ABI of libtimedhip.so, the MI355X (gfx950) engine behind the reference's two
 * numeric seams.  Plain pointers and sizes only; no C++/torch types; never throws across the ABI.
 *
 * The reference (wells-wood-research/timed-design) has no FFI/plugin layer: the seams are Python
 * call sites.  Each entry point below names the reference interface it replaces (file:line are
 * relative to the reference repo).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every function returns 0 (TH_OK) or a negative TH_E* code; th_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - the caller owns every host buffer; the library owns device buffers it allocates.
 *   - threading: a th_model handle is bound to one device.  Submissions on one handle (th_predict,
 *     th_predict_async, th_predict_device) are serialised by a lock inside the handle, and
 *     th_predict_wait may be called from a DIFFERENT thread than the one that submitted the ticket
 *     (predict.py waits on its writer thread while the main thread keeps submitting): a ticket slot
 *     is only handed out again after its waiter has returned, whether it succeeded or failed.  One
 *     waiter per ticket (a second concurrent wait on the same ticket gets TH_EBUSY).  th_model_free,
 *     th_model_set_chunk and th_model_profile must not run concurrently with anything else on the
 *     handle.  Distinct handles may be driven from distinct threads (ctypes releases the GIL).
 *   - frames are channels-last [n, D, H, W, C], C fastest — the layout
 *     design_utils/utils.py:519-527 (load_batch) builds and Keras consumes.
 */
#ifndef TIMED_HIP_H
#define TIMED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TH_OK 0
#define TH_EINVAL (-1)   /* bad argument / shape / dtype                     */
#define TH_EIO (-2)      /* pack file unreadable or malformed                */
#define TH_EHIP (-3)     /* a HIP runtime call failed (message has details)  */
#define TH_EUNSUP (-4)   /* layer/option outside the supported op set        */
#define TH_ENOMEM (-5)
#define TH_ECOMM (-6)    /* RCCL failure / librccl.so not loadable           */
#define TH_EBUSY (-7)    /* every async ticket of the model is in flight     */

/* element types accepted for frames (what load_batch can hand to Model.predict,
 * design_utils/utils.py:518-521: float64 when voxels_as_gaussian else bool) */
#define TH_F32 0
#define TH_F64 1
#define TH_U8 2
#define TH_BOOL 3
#define TH_F16 4

/* th_model_load flags */
#define TH_LOAD_DEFAULT 0u
#define TH_LOAD_NO_FUSE 1u      /* one kernel per Keras layer (debug / parity bisecting)          */
#define TH_LOAD_NO_MFMA 2u      /* direct (VALU) convolution kernels only                         */
#define TH_LOAD_KEEP_ALL 4u     /* keep every layer output materialised (th_model_fetch)          */

/* th_predict* flags */
#define TH_PREDICT_DEFAULT 0u
#define TH_PREDICT_LOGITS 1u    /* skip the final Softmax: return its input (parity on the logits) */
#define TH_PREDICT_OUT_DEVICE 2u /* th_predict/_async: probs_out is memory of the model's DEVICE (frames still come from the
                                  * host); the rows stay in HBM, e.g. for th_comm_gather_rows */
#define TH_PREDICT_IN_DEVICE 4u  /* th_predict/_async: `frames` is memory of the model's DEVICE already (e.g. frames that
                                  * th_h5_decode_device inflated there): no host->device copy; it must stay valid until the wait */

typedef struct th_model th_model;
typedef struct th_comm th_comm;

/* ---- library ---------------------------------------------------------------------------- */
int th_version(void);                 /* ABI version, currently 1 */
const char* th_last_error(void);
int th_device_count(int* n_out);
/* CPUs the host-side thread pools of this library will use: min(hardware threads, affinity, cgroup CPU quota) */
int th_host_cpus(void);
/* name/arch of a device, e.g. "AMD Instinct MI355X" / "gfx950" */
int th_device_info(int device, char* name, size_t name_len, char* arch, size_t arch_len, int* cus);

/* ---- model: replaces tf.keras.models.load_model(path) — predict.py:121 ------------------- */
/* `pack_path` is a THPK0001 pack written by timed_hip/pack.py from the .h5's model_config+weights */
int th_model_load(const char* pack_path, int device, unsigned flags, th_model** out);
int th_model_load_mem(const void* pack, size_t nbytes, int device, unsigned flags, th_model** out);
void th_model_free(th_model* m);
/* input dims {D,H,W,C} (= dataset attr frame_dims, utils.py:515) and width of the output row */
int th_model_info(const th_model* m, int dims[4], int* n_classes);
/* algorithmic work per frame: FLOPs (2*V*k^3*Cin*Cout per conv + 2*F*out per Dense) and the
 * executed FLOPs including tile padding; number of device launches per chunk */
int th_model_cost(const th_model* m, double* algo_flops, double* exec_flops, int* n_steps);
/* frames processed per internal pass (Keras' own predict() minibatch is 32 — no numeric effect) */
int th_model_set_chunk(th_model* m, int frames_per_chunk);

/* ---- forward: replaces frame_model.predict(X_batch) — predict.py:142 --------------------- */
/* host frames [n,D,H,W,C] of `dtype` -> host probs_out [n,n_classes] fp32 (rows sum to 1) */
int th_predict(th_model* m, const void* frames, int dtype, int64_t n, float* probs_out, unsigned flags);
/* The same call split in two so that the caller's loop (the per-batch loop of predict.py:125-155, whose load_batch
 * — design_utils/utils.py:487-530 — re-opens the dataset and gathers the next batch on the host) overlaps with the GPU:
 * th_predict_async queues host->device copy, kernels and the device->host copy of the probabilities and returns a
 * ticket; th_predict_wait blocks until that batch is complete and only then writes probs_out.  Up to 4 tickets per
 * model may be in flight (TH_EBUSY beyond that); they complete in submission order.  `frames` must stay valid and
 * unmodified until the matching th_predict_wait returns.  Frames in page-locked memory (th_host_alloc /
 * th_host_register) are copied by the DMA engine without blocking the caller; with pageable memory the copy of a piece
 * blocks the caller (the kernels of earlier pieces and tickets still run underneath it).  th_predict ==
 * th_predict_async + th_predict_wait. */
int th_predict_async(th_model* m, const void* frames, int dtype, int64_t n, float* probs_out, unsigned flags, int* ticket);
int th_predict_wait(th_model* m, int ticket);
/* Lossless sparse transport of float32 frames (row f-1: what design_utils/utils.py:487-530 hands to Model.predict, predict.py:142).
 * Gaussian aposteriori frames are ~8 % non-zero; over PCIe a frame then costs ~25 KB instead of 222 KB, and is rebuilt bit for bit
 * on the device in front of the first layer.  `blob` (host memory, little-endian, 16-byte aligned) is
 *     0   char     magic[8] = "THSPF001"
 *     8   uint32   n_frames, elems_per_frame E, bitmap words per frame W (>= ceil(E / 32), a multiple of 4), element bytes (4)
 *     24  uint64   n_values
 *     32  uint64   rank[n_frames + 1]      stored elements in front of frame i (rank[n] - rank[0] = n_values)
 *     ..  (to the next multiple of 16)
 *         uint32   bitmap[n_frames][W]     bit k of word w set <=> element 32 w + k is stored (every element whose bit pattern is
 *                                          not +0.0: -0.0, NaN payloads and denormals are stored)
 *         float    values[n_values]        the stored elements, frame by frame, in element order
 * Same ticket / wait / ownership rules as th_predict_async (TH_PREDICT_LOGITS and TH_PREDICT_OUT_DEVICE apply). */
#define TH_SPARSE_MAGIC "THSPF001"
int th_predict_sparse_async(th_model* m, const void* blob, size_t blob_bytes, float* probs_out, unsigned flags, int* ticket);
/* page-locked host memory for frame batches (what load_batch fills): allocate, or pin an existing range in place */
int th_host_alloc(size_t bytes, void** out);
int th_host_free(void* p);
int th_host_register(void* p, size_t bytes);
int th_host_unregister(void* p);
/* same with frames and probabilities already resident in this model's device memory */
int th_predict_device(th_model* m, const void* d_frames, int dtype, int64_t n, float* d_probs, unsigned flags);
/* copy a layer's output for the first n frames of the LAST chunk run (TH_LOAD_KEEP_ALL / unfused
 * layers only): out is [n, D,H,W,C] or [n,F] fp32.  Debug/parity aid. */
int th_model_fetch(th_model* m, const char* layer_name, int64_t n, float* out, int64_t out_floats);

/* per-launch timing with HIP events on the model's stream, accumulated over th_predict* calls until the
 * next th_model_profile call: enable = 0 off, 1 every step of the plan, 2 only the step with the most
 * algorithmic FLOPs (2 events per chunk: negligible cost inside a timed region) */
int th_model_profile(th_model* m, int enable);
/* step i of the plan: kernel label, accumulated ms and launch count since profiling was enabled,
 * algorithmic FLOPs and bytes per frame attributed to the step */
int th_model_step_info(const th_model* m, int i, char* label, size_t label_len, double* ms, int64_t* launches,
                       double* flops_per_frame, double* exec_flops_per_frame, double* bytes_per_frame);
/* the SURVEY §8(d) direct-form FLOPs per frame of step i when they differ from what the step's kernel computes (Cook-Toom /
 * Winograd steps: flops_per_frame of th_model_step_info is their own, smaller count); -1 for every other step */
int th_model_step_direct_flops(const th_model* m, int i, double* direct_flops_per_frame);

/* Load-time guard.  th_model_load checks the plan it built — Winograd layers, the bf16x3-split GEMMs — against a direct fp32-MFMA
 * plan of the same pack on four internally generated frames: the logits must agree to 1e-5 x max(1, max |logit|).  If they do
 * not, fast features are dropped (split GEMM, 5^3 Winograd, fused 10^3 Winograd, first-layer F(2,3), in that order) until they
 * do.  state: 0 not run (TH_GUARD=0 or nothing to check), 1 passed, 2 tripped — `note` then says what was measured and dropped.
 * max_dlogit: the kept plan's distance from the direct plan; logit_scale: max |logit| of the direct plan on the guard frames. */
int th_model_guard_info(const th_model* m, int* state, double* max_dlogit, double* logit_scale, char* note, size_t note_len);

/* the A/B and test knobs (TH_* environment variables) this handle was loaded under, as "NAME=value ..." — empty when every
 * knob had its default.  They are read once, by th_model_load, and never again (a later setenv does not reach a loaded handle). */
int th_model_knobs(const th_model* m, char* buf, size_t buf_len);

/* ---- device memory helpers (so host code needs no torch/hip-python for resident buffers) -- */
int th_dev_alloc(int device, size_t bytes, void** d_out);
int th_dev_free(int device, void* d);
int th_dev_upload(int device, void* d_dst, const void* h_src, size_t bytes);
int th_dev_download(int device, void* h_dst, const void* d_src, size_t bytes);
int th_dev_sync(int device);
/* fill d_frames [n,side,side,side,channels] fp32 with synthetic Gaussian-splat frames generated on
 * the device (Philox; statistically like timed_hip.synth.synthetic_frames, not bit-identical) */
int th_dev_synth_frames(int device, float* d_frames, int64_t n, int side, int channels, int atoms, uint64_t seed);

/* ---- sampler: replaces design_utils/sampling_utils.py ------------------------------------- */
/* apply_temp_to_probs (sampling_utils.py:139-161): q = p**(1/t), rows renormalised; fp64 */
int th_apply_temp(const double* probs, int64_t n_res, int n_cls, double t, double* out);           /* device 0 */
int th_apply_temp_on(int device, const double* probs, int64_t n_res, int n_cls, double t, double* out);
/* random_choice_prob_index (sampling_utils.py:53-90, kernel lines 81-82), n_samples draws fused:
 *   idx[s,i] = first j with cumsum_j(q[i,:]) > r[s,i], 0 if none   (q = tempered probs, t==1: q=p)
 * rng_mode 0: r = uniforms[s*n_res+i] supplied by the caller (e.g. np.random.rand — bit-exact
 *             replay of the reference's MT19937 stream);
 * rng_mode 1: r drawn on the device: rocRAND Philox4x32-10, seed, subsequence s*n_res+i;
 * rng_mode 2: r drawn on the device from MT19937 seeded like np.random.seed(seed), consumed in the
 *             reference's single-process order (for s: r = rand(n_res)). */
#define TH_RNG_HOST 0
#define TH_RNG_PHILOX 1
#define TH_RNG_MT19937 2
#define TH_RNG_MT_WORDS 3 /* th_sampler_run only: `uniforms` holds RAW MT19937 state words (uint32, two per draw, as th_mt19937_words
                           * returns them); the draw kernel tempers them and forms genrand_res53 itself — the doubles are the ones
                           * np.random.rand returns, the host only advances the recurrence */
int th_sample(const double* probs, int64_t n_res, int n_cls, int64_t n_samples, double temperature,
              int rng_mode, uint64_t seed, const double* uniforms, int32_t* idx_out);               /* device 0 */
/* optionally also return the uniforms the device drew (r_out [n_samples,n_res], may be NULL) and
 * residue letters (letters_out [n_samples, n_res] bytes, via cat_letters[n_cls], may be NULL) */
/* rng_offset = uniforms already consumed from the device stream (Philox: added to the subsequence
 * index; MT19937: doubles skipped) so successive calls continue one stream, as the reference's
 * per-PDB loop does.  q_out (may be NULL) receives the tempered probabilities [n_res,n_cls]. */
int th_sample_ex(const double* probs, int64_t n_res, int n_cls, int64_t n_samples, double temperature,
                 int rng_mode, uint64_t seed, uint64_t rng_offset, const double* uniforms, int32_t* idx_out,
                 double* r_out, const char* cat_letters, char* letters_out, double* q_out, int device);

/* A resident sampler: the probability rows of a whole run (every PDB key of sample.py's prediction matrix) stay on
 * one device, tempered and normalised, with their running sums; any number of sequences for any contiguous range of
 * keys is then drawn by ONE launch sequence — instead of one upload + launch + synchronise per key as the per-PDB
 * loop of sample_with_multiprocessing (sampling_utils.py:164-197) would do.  A handle serialises its own calls. */
typedef struct th_sampler th_sampler;
int th_sampler_create(int device, th_sampler** out);
void th_sampler_free(th_sampler* s);
#define TH_TEMPER_NONE 0        /* rows used as they are (sample.py:40 skips apply_temp_to_probs at T == 1)        */
#define TH_TEMPER_POW 1         /* q = p**(1/t), rows renormalised (sampling_utils.py:159-161)                      */
#define TH_TEMPER_PREPOWERED 2  /* rows already hold p**(1/t) (powered by the caller's NumPy); rows renormalised    */
/* probs [n_rows, n_cls] fp64 (host).  TH_TEMPER_POW: exponents 1, 2 and 0.5 are exact IEEE operations on the device
 * (NumPy takes the same fast paths); any other exponent is raised on the HOST with libm pow() before the upload —
 * device pow() is not correctly rounded and one ulp in q can move a residue index.  NumPy's own float64 `**` is libm
 * pow() in its scalar loop but an SVML routine under AVX-512, which differs in the last bit for ~5 % of inputs: a
 * caller that must reproduce one particular NumPy build bit for bit powers the rows itself and passes
 * TH_TEMPER_PREPOWERED (design_utils.sampling_utils does).  The normaliser follows NumPy's pairwise summation order
 * and the running sum is strictly sequential.  q_out (may be NULL) receives the tempered rows.
 * cum_dtype (TH_F64, TH_F32 or TH_F16) is the type the running sum is rounded to after every addition: np.cumsum
 * accumulates in the array's own dtype and the reference passes predict's float16 rows unconverted
 * (sampling_utils.py:82,125); sample.py's own path is float64. */
int th_sampler_load(th_sampler* s, const double* probs, int64_t n_rows, int n_cls, double temperature, int temper_mode,
                    int cum_dtype, double* q_out);
/* Key k owns rows [row_off[k], row_off[k+1]) of the loaded matrix (row_off has n_keys+1 ascending entries).  Draws are
 * numbered in the reference's consumption order — for key: for sample: rand(n_res_key) (sampling_utils.py:118-125):
 *   d(k, s, i) = n_samples*(row_off[k]-row_off[0]) + s*n_res_k + i
 * which is the position of the draw's uniform in `uniforms` (TH_RNG_HOST) or in the device stream (offset by
 * rng_offset), and of its result in idx_out / r_out / letters_out (each may be NULL).  cat_letters[n_cls] maps a
 * category to its one-letter code.  metrics_out (may be NULL) receives, per sampled sequence in the same key-major
 * order, [charge at pH 7.4, isoelectric point, molecular weight, molar extinction at 280 nm] — the tuple
 * calculate_seq_metrics (design_utils/analyse_utils.py:351-371) appends to every sequence at sampling_utils.py:132;
 * computed on the device from a 20-bin residue histogram per sequence (constants: ampal's published tables as
 * restated in design_utils/analyse_utils.py — parity unpinned, ampal is not in the reference tree). */
int th_sampler_draw(th_sampler* s, int64_t n_keys, const int64_t* row_off, int64_t n_samples, int rng_mode, uint64_t seed,
                    uint64_t rng_offset, const double* uniforms, const char* cat_letters, int32_t* idx_out, double* r_out,
                    char* letters_out, double* metrics_out);

/* A whole sample.py run in ONE submission (reference sampling_utils.py:118-133 for every key at once): the rows [n_rows, n_cls]
 * are used as they are (sample.py tempers the matrix first, sample.py:40-41), their running sums (in cum_dtype), every draw, the
 * one-letter codes and the per-sequence metrics are computed by two kernels between ONE host->device copy (rows, offsets and
 * letters travel together) and ONE device->host copy; caller-supplied uniforms (TH_RNG_HOST) add the copy of those.  Keys must
 * cover rows 0..n_rows.  want: bit 0 indices (int32 [total]), bit 1 letters (char [total]), bit 2 metrics (double [n_keys *
 * n_samples][4]) — in th_sampler_draw's draw order.  *block_out points at a page-locked block owned by the sampler (valid until
 * its next call); offsets_out[0..2] = byte offset of indices / letters / metrics in it, -1 when not requested.  Indices and
 * letters are bit-identical to th_sampler_load + th_sampler_draw on the same rows, uniforms / generator settings. */
/* a page-locked buffer of at least `bytes` owned by the sampler (grow-only, valid until the next call of this function on it or
 * th_sampler_free): uniforms or raw generator words written here reach the device by direct DMA instead of through HIP's
 * pageable staging copy */
int th_sampler_uniform_buffer(th_sampler* s, size_t bytes, void** out);
int th_sampler_run(th_sampler* s, const double* probs, int64_t n_rows, int n_cls, int cum_dtype, int64_t n_keys, const int64_t* row_off,
                   int64_t n_samples, int rng_mode, uint64_t seed, uint64_t rng_offset, const double* uniforms, const char* cat_letters,
                   unsigned want, const void** block_out, int64_t* offsets_out);

/* np.random.rand(n) of NumPy's GLOBAL legacy generator — the reference's source of uniforms, r = np.random.rand(n),
 * sampling_utils.py:81 — replayed natively (host code): key = the 624 MT19937 state words and *pos the position in them, as
 * np.random.get_state() returns them; out receives the same n doubles (genrand_res53) NumPy would produce and key / *pos are left
 * as NumPy would leave them, so np.random.set_state((name, key, pos, has_gauss, cached)) continues the stream seamlessly. */
int th_mt19937_rand(uint32_t* key, int* pos, int64_t n, double* out);
/* the same walk through the generator, but `out` receives the 2 n RAW state words the n doubles are made of (untempered, in
 * consumption order: double i = res53(temper(out[2 i]) >> 5, temper(out[2 i + 1]) >> 6)); key / *pos advance exactly as in
 * th_mt19937_rand.  For TH_RNG_MT_WORDS: the recurrence stays on the host (it is sequential), tempering and conversion — two
 * thirds of th_mt19937_rand's time — move into the draw kernel. */
int th_mt19937_words(uint32_t* key, int* pos, int64_t n, uint32_t* out);

/* ---- text output: replaces np.savetxt(f, matrix, delimiter=",") — design_utils/utils.py:768-771 (float16
 * probabilities) and predict.py:145-146 (full-precision rotamer matrix).  Host code only.  Formats the row-major
 * [n, k] matrix exactly as NumPy does (every value '%.18e', ',' between columns, '\n' after each row, NaN as
 * 'nan') into out; dtype is TH_F16 (preformatted table), TH_F32 or TH_F64.  Returns the number of bytes written,
 * or a negative TH_E* code (cap too small: 28 bytes per value always suffice). */
int64_t th_format_csv(const void* data, int dtype, int64_t n, int64_t k, char* out, int64_t cap);
/* The same text for a float32 matrix, formatted ON THE DEVICE (one lane per value): np.savetxt(f, y_pred_batch, delimiter=",") of
 * predict.py:145-146 — the full-precision rotamer matrix, 338 values per residue, ~1 GB of text per 125 000 residues.  rows: host
 * memory, [n, k] float32; out: host memory, cap >= 25 n k.  Every finite non-negative float32 below 2^24 (every probability) is
 * exactly 24 characters in '%.18e' form, so the text is 25 n k bytes (returned).  TH_EUNSUP when a value does not have that form
 * (negative, NaN, infinite, >= 2^24): the caller formats the block with th_format_csv instead — same bytes for every value both
 * accept (csrc/fmt_e18_f32.h is compiled for host and device).  Keeps a stream and two device buffers for `device` between calls;
 * th_format_csv_device_release gives them back.  Serialised internally (one formatter per process). */
int64_t th_format_csv_device(int device, const float* rows, int64_t n, int64_t k, char* out, int64_t cap);
int th_format_csv_device_release(void);
/* dataset-map text -> string table: replaces np.genfromtxt(dataset_map_path, delimiter=",", dtype="str") — predict.py:99.
 * th_csv_shape validates plain ASCII text with the same number of `delim`-separated, non-empty fields on every line and
 * reports (rows, cols, longest field); TH_EUNSUP for anything NumPy treats specially (comments, quotes, '\r', non-ASCII,
 * ragged or blank lines, blank-padded fields): the caller then falls back to NumPy.  th_csv_fill writes the fields as
 * UCS-4 code units into out[rows][cols][width], zero padded — the memory of a NumPy '<U{width}' array.  Host code. */
int th_csv_shape(const char* text, int64_t len, char delim, int64_t* rows_out, int* cols_out, int* width_out);
int th_csv_fill(const char* text, int64_t len, char delim, int64_t rows, int cols, int width, uint32_t* out);
/* argmax + residue letter per row: replaces max_idx = np.argmax(prediction_matrix, axis=1) and the per-residue string
 * appends of extract_sequence_from_pred_matrix — design_utils/utils.py:659, :689-692.  matrix [n, k] of dtype TH_F16 /
 * TH_F32 / TH_F64; np.argmax rules (first maximum; the first NaN of a row wins).  letters_out[i] = col_letters[argmax_i]
 * (col_letters: k bytes, the one-letter code of every probability column); idx_out (int32[n]) optional; either output
 * may be NULL.  Host code, rows split over host threads. */
int th_argmax_letters(const void* matrix, int dtype, int64_t n, int64_t k, const char* col_letters, char* letters_out,
                      int32_t* idx_out);

/* ---- frame ingest: replaces the per-residue h5py reads of load_batch — design_utils/utils.py:514-529.  Host code
 * only.  `file` is the whole HDF5 file in memory (an mmap), `base` its superblock offset.  For n_datasets chunked
 * datasets that share one geometry (shape[rank], chunk[rank], element size, filter pipeline ids in write order:
 * 1 deflate, 2 shuffle, 3 fletcher32) and whose chunk B-trees (version 1) start at btree_addrs[i], inflate every
 * chunk and scatter it into dests[i] (C order, shape[] elements of esz bytes) on nthreads host threads (0 = all
 * usable CPUs — th_host_cpus() —, at most 128).  Unallocated chunks read as zeros.  TH_EUNSUP when the file uses something else (the
 * caller then reads through its generic path). */
int th_h5_read_chunked(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs,
                       void* const* dests, int rank, const int64_t* shape, const int64_t* chunk, int esz, int n_filters,
                       const int* filter_ids, int nthreads);

/* the same with a conversion while the chunks are placed: conv 0 = bytes as stored, 1 = float64 -> float32 (round to
 * nearest even: exactly the cast Keras applies to load_batch's float64 frames; halves the host->device bytes) */
int th_h5_read_chunked_as(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs,
                          void* const* dests, int rank, const int64_t* shape, const int64_t* chunk, int esz, int n_filters,
                          const int* filter_ids, int nthreads, int conv);
/* contiguous (unchunked, unfiltered) datasets: data_addrs[i] = file address of `count` elements of `esz` bytes, -1 = never
 * written (zeros) */
int th_h5_read_contiguous_as(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* data_addrs,
                             void* const* dests, int64_t count, int esz, int conv);
/* The same datasets decoded ON THE DEVICE (f-1: the host's inflate rate — 1.3 k frames/s per core — is what limits predict.py on
 * real gzip .hdf5 datasets, reference design_utils/utils.py:514-529).  Only the chunk B-trees are walked on the host; the
 * COMPRESSED chunk bytes are copied to the GPU as they lie in the file, inflated there one lane per chunk, and placed into d_out —
 * device memory on `device`, [n_datasets][shape...] of float32 when conv = 1 (float64 data: the cast Keras applies) or of the
 * stored element type when conv = 0.  Supports the pipeline aposteriori writes (n_filters = 1, filter id 1 = deflate, every chunk
 * compressed) and shuffle + deflate (filter ids {2, 1}: h5py's compression="gzip", shuffle=True; chunks up to 60 KB); TH_EUNSUP
 * for anything else (use th_h5_read_chunked_as).  Every chunk's Adler-32 is verified as zlib does: a chunk that does not inflate
 * to its declared size or fails the check is TH_EIO.  Never-allocated chunks read as zeros.  Synchronous. */
int th_h5_decode_device(const void* file, int64_t file_len, int64_t base, int64_t n_datasets, const int64_t* btree_addrs, int rank,
                        const int64_t* shape, const int64_t* chunk, int esz, int n_filters, const int* filter_ids, int conv, int device,
                        void* d_out);
/* th_model_free keeps a model's device blocks (weights, arenas, rings) in a per-process cache for the next th_model_load
 * (hipFree synchronises the device: ~13 ms per TIMED handle, paid by every predict.py call that loads and frees a model as
 * reference predict.py:114-121 does); at most min(24 GB, an eighth of the device's memory) are kept per device, the blocks parked
 * longest ago leave first, and every allocator inside the library returns the cache to HIP and retries before it reports
 * TH_ENOMEM.  th_dev_trim returns the cached blocks of `device` (all devices when negative) to HIP; th_dev_cache_info reports
 * what is parked (bytes, the cap, number of blocks). */
int th_dev_trim(int device);
int th_dev_cache_info(int device, uint64_t* cached_bytes, uint64_t* cap_bytes, int* blocks);
/* th_h5_decode_device keeps per-device scratch memory between calls (compressed span, token arena: ~5 bytes per uncompressed
 * byte of the largest batch so far).  When an allocation fails it frees that scratch and returns TH_ENOMEM — decode fewer
 * datasets per call or read through th_h5_read_chunked_as; th_h5_release_scratch frees it on request (end of a run). */
int th_h5_release_scratch(int device);
/* n independent zlib (wrapped = 1) or raw DEFLATE (wrapped = 0) streams inflated on `device`: stream i is
 * comp[src_off[i] .. + src_len[i]) and decodes to exactly dst_len[i] bytes at out[dst_off[i]] (8-byte aligned).  comp / out are
 * host buffers.  status_out[i] (optional): 0, or why stream i failed (1 input exhausted, 2 output overrun, 3 bad zlib header,
 * 4 invalid code, 5 distance too far back, 6 bad code table, 7 bad stored block, 8 ended short of dst_len, 9 Adler-32 of the
 * output differs from the zlib trailer — what zlib reports as "incorrect data check"; raw streams carry none).  TH_EIO if any failed. */
int th_inflate_many(int device, const void* comp, int64_t comp_len, int64_t n, const int64_t* src_off, const int64_t* src_len,
                    const int64_t* dst_off, const int64_t* dst_len, void* out, int64_t out_len, int wrapped, int* status_out);

/* Resolve MANY datasets' object headers in one call — the per-residue `dataset[pdb][chain][res]` header parse and the
 * two attribute reads of load_batch (`encoded_residue`, utils.py:529) and create_flat_dataset_map (`label`,
 * utils.py:375).  ohdr_addrs[n] are object-header addresses (from the chain groups' symbol tables).  Outputs:
 * btree_out[i] (chunk B-tree address or -1), geom_out[40] (rank, shape[7], chunk[7], element size, datatype class,
 * signed, n_filters, filter ids[8], layout class (1 contiguous: btree_out is then the data address; 2 chunked) of the
 * FIRST dataset), status_out[i] bits: 1 = storage with exactly geom_out's geometry, 2 = numeric attribute `num_attr` copied to num_out[i*num_len ..] as doubles, 4 = string
 * attribute `str_attr` copied NUL-terminated to str_out[i*str_len ..].  Either attribute name may be NULL.  A clear
 * bit means "use the general reader for this one"; nothing is guessed. */
int th_h5_resolve(const void* file, int64_t file_len, int64_t base, int64_t n, const int64_t* ohdr_addrs, const char* num_attr,
                  double* num_out, int num_len, const char* str_attr, char* str_out, int str_len, int64_t* btree_out,
                  int64_t* geom_out, int* status_out, int nthreads);

/* Every link of one old-style HDF5 group in one call (create_flat_dataset_map, reference utils.py:357-375, lists each pdb
 * group, its chain groups and every residue name: `for pdb_code in dataset_file`, `.keys()`).  btree_addr: the group B-tree of
 * the symbol-table message; heap_data / heap_size: absolute offset and length of the local heap's data segment.  names receives
 * the link names, NUL-terminated, in B-tree (name) order — *names_len bytes —, addrs[i] the object-header address of link i,
 * *n_out the count.  TH_ENOMEM: a capacity was too small; TH_EINVAL: not a well-formed group B-tree inside the file. */
int th_h5_group_links(const void* file, int64_t file_len, int64_t base, int64_t btree_addr, int64_t heap_data, int64_t heap_size,
                      char* names, int64_t names_cap, int64_t* addrs, int64_t addrs_cap, int64_t* n_out, int64_t* names_len);

/* ---- voxeliser: the producer of the frames — replaces aposteriori.make_frame_dataset as the reference invokes it
 * (ui.py:73-86: frame_edge_length 21.0, voxels_per_side 21, Codec.CNOCACB, voxels_as_gaussian=True; README.md:83-97).
 * PARITY UNPINNED: aposteriori's source is not in the reference tree; the specification implemented here is written out
 * in timed_hip/voxeliser.py and restated by oracle/voxel_oracle.py.  Host arrays in: atoms_xyz [n_atoms,3] (Angstrom),
 * atom_channel [n_atoms] (index into the atom encoder, < 0 = not encoded), atom_sigma [n_atoms] (Gaussian width in
 * Angstrom; may be NULL for boolean frames), frames_rt [n_res,12] = per residue a row-major 3x3 rotation (rows = local
 * x, y, z axes) followed by the origin (the residue's CA).  Out: [n_res, V, V, V, n_channels], float32 when gaussian
 * else uint8 (0/1); `out` is host memory, or device memory of `device` when out_on_device (frames then go straight to
 * th_predict_device without leaving HBM).  One frame holds at most 2048 encodable atoms (TH_EUNSUP beyond). */
int th_voxelise(int device, const float* atoms_xyz, const int32_t* atom_channel, const float* atom_sigma, int64_t n_atoms,
                const float* frames_rt, int64_t n_res, int voxels_per_side, float frame_edge_length, int n_channels, int gaussian,
                void* out, int out_on_device);

/* ---- multi-GPU reassembly (no reference counterpart: the reference is single-process) ----- */
/* one process per GPU; rank 0 creates the id and ships it to the others out of band */
#define TH_COMM_ID_BYTES 128
int th_comm_unique_id(char id[TH_COMM_ID_BYTES]);
int th_comm_init(const char id[TH_COMM_ID_BYTES], int n_ranks, int rank, int device, th_comm** out);
void th_comm_free(th_comm* c);
/* gather contiguous per-rank row shards [counts[r], width] fp32 (device pointers) into
 * d_out [sum(counts), width] on `root` (RCCL send/recv over xGMI; d_out ignored elsewhere) */
int th_comm_gather_rows(th_comm* c, const float* d_local, const int64_t* counts, int width, int root,
                        float* d_out);
int th_comm_barrier(th_comm* c);
/* what this communicator's gathers have issued so far on THIS rank: out = {ncclSend calls, ncclRecv calls, bytes sent, bytes
 * received, device copies of the root's own block}.  A 1-rank gather issues no RCCL transfer (the root's block is a copy) unless
 * TH_COMM_SELF_RCCL=1 was set when th_comm_init ran: then that block goes through a grouped ncclSend/ncclRecv pair to self. */
int th_comm_stats(th_comm* c, int64_t out[5]);

#ifdef __cplusplus
}
#endif
#endif /* TIMED_HIP_H */
