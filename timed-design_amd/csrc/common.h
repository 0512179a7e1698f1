// Shared declarations for the timed_hip runtime (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/timed_hip.h"

// ---- op / activation codes: keep in sync with timed_hip/keras_config.py ------------------
enum : int {
    OP_INPUT = 0, OP_CONV3D, OP_DENSE, OP_BN, OP_ACT, OP_MAXPOOL, OP_AVGPOOL, OP_GAP, OP_GMP,
    OP_FLATTEN, OP_CONCAT, OP_ADD, OP_IDENTITY
};
enum : int { ACT_LINEAR = 0, ACT_RELU, ACT_ELU, ACT_SOFTMAX, ACT_SIGMOID, ACT_TANH, ACT_LEAKY };

// ---- error plumbing ---------------------------------------------------------------------
void th_set_error(const char* fmt, ...);
#define TH_FAIL(code, ...)          \
    do {                            \
        th_set_error(__VA_ARGS__);  \
        return (code);              \
    } while (0)
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            th_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TH_EHIP;                                                                      \
        }                                                                                        \
    } while (0)

// ---- A/B and test knobs ---------------------------------------------------------------------
// Every TH_* environment variable that changes what a MODEL plans or launches is read exactly once, by th_model_load, into the
// handle (th_knobs_read).  Planners take the snapshot of the load in progress (th_knobs_planning) and leave a pointer to it
// in their plan structs, launchers read it from there: no getenv in a launch path, no process-wide static caches — two models
// loaded under different environments keep their own settings, whatever the call order.  `nondefault` lists what was set
// ("TH_WINOGRAD=0 TH_WF_DBG=3"); th_model_knobs() returns it and bench.py prints it.
struct ThKnobs {
    int winograd = 1;          // TH_WINOGRAD: 0 direct kernels, 1 F(3,3)+F(2,3) in-plane (default), 2 F(5,3) (opt-in)
    int wino_split = 1;        // TH_WINO_SPLIT: Winograd GEMMs on bf16 MFMA with exactly split operands (0: fp32-input MFMA)
    int wfused = 1;            // TH_WFUSED: 10^3 layers on conv_wfused.hip
    int wf_split = 1;          // TH_WF_SPLIT: eligible conv_wfused layers on bf16 MFMA with exactly split operands (conv_wfsplit.hip)
    int lanes = 1, lane_lag = 1;   // TH_LANES, TH_LANE_LAG
    int guard = 1;             // TH_GUARD: load-time check of the fast plans against the direct fp32 plan (0: off)
    int first_wino = 1;        // TH_FIRST_WINO=0: k_conv_first instead of k_conv_first_w
    int first_split = 1;       // TH_FIRST_SPLIT: the aposteriori first layer on bf16 MFMA with exactly split operands (conv_first_b3.hip)
    int conv_gl = 1;           // TH_CONV_GL: 1 strided convolutions and those with <= 64 outputs per frame on conv_gl.hip (2: every eligible layer, 0: none)
    int dense_gemm = 1;        // TH_DENSE_GEMM: Dense layers of >= 512 features and 8..128 outputs as a batch GEMM on fp32 MFMA (dense_gemm.hip)
    int first_int = 1;         // TH_FIRST_INT: uint8 / bool frames on the one-piece form of conv_first_b3 (0: the general six-product kernel)
    int first_zb = 0;          // TH_FIRST_ZB: brick depth of the first-layer kernel (tuning)
    int first_dbg = 0;         // TH_FIRST_DBG: timing knock-outs (results wrong)
    int no_pool_first = 0;     // TH_NO_POOL_FIRST: act / BN before the max-pool even when the chain is monotone
    int no_tail_fuse = 0;      // TH_NO_TAIL_FUSE: keep GAP / Dense / Softmax as separate launches
    int conv_nogeo = 0, n16_nogeo = 0, conv_noxc = 0, conv_notail = 0, conv_nozmajor = 0, conv_nocompact = 0, conv_nopw = 0;
    int conv_bmode = 0;        // TH_CONV_BMODE: 1 dbuf, 2 stream8, 3 no16
    int conv_dbg = 0, conv_ldspad = 0, n16_resident = 0;
    int pw_nopipe = 0, pw_noepi = 0, pw_dbg = 0;
    int wf_resident = 0, wf_dbg = 0, wf_noblk = 0;
    long long wino_piece = 0;
    int wino_dbg = 0, wino_var = 3, wino_b3var = 0, wino_nomid = 0;
    std::string nondefault;
};
void th_knobs_read(ThKnobs* k);                 // the process environment, now
const ThKnobs& th_knobs_planning();             // the snapshot of the th_model_load in progress on this thread (defaults outside one)
void th_knobs_set_planning(const ThKnobs* k);   // runtime.hip, around the planner
inline const ThKnobs& th_knobs_of(const ThKnobs* k) { static const ThKnobs dflt; return k ? *k : dflt; }

// hipMalloc for every allocator inside the library: when the device is out of memory the blocks parked in the model block
// cache (runtime.hip: freed arenas kept for the next load) are returned to HIP and the allocation is tried once more — a dead
// block in the cache must never be the reason a decode, a text formatter or a sampler runs out of memory.
hipError_t th_malloc_retry_impl(void** p, size_t bytes);
template <class T> inline hipError_t th_malloc_retry(T** p, size_t bytes) { return th_malloc_retry_impl(reinterpret_cast<void**>(p), bytes); }

// CPUs this process may actually use: min(hardware threads, scheduler affinity, cgroup CPU quota).  Containers often
// expose every host core but enforce a quota (cpu.max); more runnable threads than the quota are throttled in 100 ms
// periods — measured on the GPU box (256 cores visible, quota 16): 20 k frames/s inflated with 16 threads, a bimodal
// 11 k / 100 k with 128.  Host-side thread pools size themselves with this.
int th_usable_cpus();

// ---- device-side views --------------------------------------------------------------------
// A channels-last activation tensor, possibly a channel slice of a wider (concat) buffer.
// element (f, v, c) lives at p[f*fs + v*cs + coff + c], v = (z*H + y)*W + x.
struct TView {
    float* p = nullptr;
    int D = 1, H = 1, W = 1, C = 0;
    int cs = 0;      // floats between consecutive voxels (>= C)
    int coff = 0;    // first channel of this view inside the buffer
    int64_t fs = 0;  // floats between consecutive frames
    // blk = 4: chunk-blocked storage [C / 4][voxel][4] instead of channels-last — element (f, v, c) at
    // p[f*fs + (c >> 2) * V() * 4 + v * 4 + (c & 3)] (cs = C, coff = 0).  The planner hands it only to a tensor whose single
    // producer (conv_pointwise.hip, conv_first.hip) and single consumer (conv_wfused.hip, which reads 4-channel slices of whole
    // frames: contiguous 16 KB instead of 16 bytes out of every voxel's row) both implement it; every other kernel sees 0.
    int blk = 0;
    int V() const { return D * H * W; }
};

#define TH_MAX_POST 4
enum : int { POP_ACT = 1, POP_AFFINE = 2 };
// elementwise epilogue applied to a conv/dense accumulator (after the bias), in order.
struct PostOps {
    int n = 0;
    int type[TH_MAX_POST] = {0, 0, 0, 0};
    int act[TH_MAX_POST] = {0, 0, 0, 0};
    float alpha[TH_MAX_POST] = {0, 0, 0, 0};
    const float* scale[TH_MAX_POST] = {nullptr, nullptr, nullptr, nullptr};  // per output channel
    const float* shift[TH_MAX_POST] = {nullptr, nullptr, nullptr, nullptr};
    // every op is monotone non-decreasing (ReLU/ELU/LeakyReLU with alpha >= 0, sigmoid, tanh, BN-affine with
    // scale >= 0 on every channel): max-pooling then commutes with the whole chain, so a kernel may pool the raw
    // accumulators first and run the chain on 1/8 of the values (set by the planner, runtime.hip)
    int monotone = 0;
};
// elementwise prologue applied to the conv input when it is staged (BN->ReLU->Conv chains)
struct PreOp {
    const float* scale = nullptr;  // per input channel (nullptr: no affine)
    const float* shift = nullptr;
    int act = ACT_LINEAR;
    float alpha = 0.f;
};

// ---- generic kernels (kernels_generic.hip) -------------------------------------------------
struct ConvGeom {
    int kd, kh, kw, sd, sh, sw, dd, dh, dw;  // kernel, stride, dilation
    int pz, py, px;                          // padding before (Keras 'same' rule) or 0
};
int launch_convert_frames(hipStream_t s, const void* src, int dtype, int64_t n, int V, int C, TView dst);
// weights in Keras layout [kd,kh,kw,Cin,Cout]
int launch_conv3d_direct(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, const float* w, const float* bias,
                         PreOp pre, PostOps post);
int launch_pool3d(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, int is_max);
int launch_eltwise(hipStream_t s, int64_t n, TView in, TView out, PostOps ops);
int launch_global_pool(hipStream_t s, int64_t n, TView in, TView out, int is_max);
// dense on a contiguous [n, F] input (in.C = F, V = 1); weights [F, out]
int launch_dense(hipStream_t s, int64_t n, TView in, TView out, const float* w, const float* bias, PostOps post);
// conv_gl.hip: small-volume / strided convolutions as an implicit GEMM with rows across the batch, operands straight from L2
inline bool conv_gl_wanted(int knob, const ConvGeom& g, int Vo) { return knob == 2 || (knob == 1 && (g.sd > 1 || g.sh > 1 || g.sw > 1 || Vo <= 64)); }
bool conv_gl_ok(int Cin, int Cout, int in_cs, int in_coff, int64_t in_fs);
size_t conv_gl_wpk_floats(const ConvGeom& g, int Cin, int Cout);
double conv_gl_exec_flops(const ConvGeom& g, int Cin, int Cout, int Vo);
std::string conv_gl_label(int Cout);
void conv_gl_pack_weights(const ConvGeom& g, int Cin, int Cout, const float* w, float* dst);
int launch_conv_gl(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, int Cin, int Cout, const float* wpk, const float* bias, PreOp pre, PostOps post);
// dense_gemm.hip: the same layer as one fp32-MFMA GEMM over the batch
bool dense_gemm_ok(int F, int O, int64_t xfs);
int launch_dense_gemm(hipStream_t s, int64_t n, TView in, TView out, const float* w, const float* bias, PostOps post);
int launch_softmax(hipStream_t s, int64_t n, TView in, TView out);
// GlobalAveragePooling3D -> Softmax in one launch (<= 512 channels): writes the pooled logits and the probabilities
int launch_gap_softmax(hipStream_t s, int64_t n, TView in, TView logits, TView probs);
int launch_tail_dense(hipStream_t s, int64_t n, TView in, PostOps pre, TView pooled, TView logits, TView probs, const float* w,
                      const float* bias, PostOps post, int softmax);
int launch_copy(hipStream_t s, int64_t n, TView in, TView out);
int launch_add(hipStream_t s, int64_t n, TView a, TView b, TView out);
int launch_synth_frames(hipStream_t s, float* d, int64_t n, int side, int channels, int atoms, uint64_t seed);

// ---- MFMA implicit-GEMM convolution (conv_mfma.hip) ---------------------------------------
struct ConvMfmaPlan {
    int cfg = -1;            // index into the instantiated tile configurations
    int CI = 0, CS = 0;      // input channels per staged chunk, LDS channel stride (floats)
    int BN = 0;              // output channels per workgroup
    int nnb = 0;             // workgroups along Cout
    int nchunks = 0;
    int pool = 0;            // 0 none, 1 max 2x2x2, 2 avg 2x2x2 (stride 2, valid)
    int Dc = 0, Hc = 0, Wc = 0;   // conv-output extent that is computed (even-trimmed when pooled)
    int FB = 1, ZB = 0, nzb = 1;  // frames / conv z-planes per workgroup, z bricks per frame
    int Zp = 0, Hp = 0, Wp = 0;   // staged (haloed) brick extent
    int rows_pf = 0;              // GEMM rows per frame-brick (multiple of 32)
    int bres = 0;                 // all taps' weights resident in LDS
    size_t lds_bytes = 0;
    size_t tab_off = 0;           // byte offset of the row tables inside the LDS allocation
    size_t wpk_floats = 0;        // size of the prepacked weight image
    double exec_flops = 0;        // MFMA FLOPs actually issued per frame
    int first_wino = 0;           // first-layer kernel only: F(2,3) along x (k_conv_first_w), rows are x pairs
    double own_flops = 0;         // first_wino: the algorithm's own multiply-adds x 2 per frame (4 points per x pair and (dz, dy) tap)
    int geo = 0;                  // > 0: the kernel instantiation with Hp = Wp = geo at compile time (tap offsets as immediates)
    const ThKnobs* knobs = nullptr;   // the owning model's A/B knobs (launch-time ones: dbg, resident, ...)
    std::string label;
};
// choose a tiling for this convolution; returns false when the MFMA kernel does not apply
// (stride/dilation != 1, nothing fits in LDS, ...)
bool conv_mfma_plan(const TView& in, const TView& out_conv, const ConvGeom& g, int Cin, int Cout, int pool,
                    ConvMfmaPlan* plan);
// a narrower instantiation for the last Cout block of a layer planned on 128-column blocks (see conv_mfma.hip); on success
// the main launch covers output channels [0, *cout_main) and `tail` the rest
bool conv_mfma_plan_tail(const TView& in, const TView& out_conv, const ConvGeom& g, int Cin, int Cout, int pool,
                         const ConvMfmaPlan& main, ConvMfmaPlan* tail, int* cout_main);
// host-side weight re-layout: Keras [kd,kh,kw,Cin,Cout] -> [nb][chunk][tap][BN][CS]
void conv_mfma_pack_weights(const ConvMfmaPlan& p, const ConvGeom& g, int Cin, int Cout, const float* w_keras,
                            float* dst);
int launch_conv_mfma(hipStream_t s, int64_t n, const ConvMfmaPlan& p, TView in, TView out, ConvGeom g, int Cin,
                     int Cout, const float* wpk, const float* bias, PreOp pre, PostOps post);

// ---- Cook-Toom / Winograd convolution for 3x3x3 'same' layers on 5^3 volumes (conv_wino.hip) ----
struct ConvWinoPlan {
    int P = 9;                                   // evaluation points per in-plane axis: 9 = F(3,3)+F(2,3) (default), 7 = F(5,3)
    int Cin = 0, Cout = 0, Coutp = 0, ncb = 0;   // Coutp: Cout rounded up to the 128-column GEMM block
    int Cinp = 0;                                // Cin rounded up to the GEMM's 32-channel chunk: the channel count of V (padding channels are zero)
    size_t wpk_floats = 0;                       // transformed weights in fragment-stream order
    int64_t v_fpf = 0, m_fpf = 0;                // scratch floats per frame: transformed input V, GEMM output M
    double gemm_flops = 0;                       // the algorithm's multiply-adds x 2 per frame: P^2 positions x 13 z-tap pairs x Cin x Cout
    double exec_flops = 0;                       // MFMA FLOPs issued per frame (Cout padded to the column block)
    int split = 0;                               // 1: the GEMM runs on bf16 MFMA with both operands split exactly into three bf16 pieces
                                                 // (x = h + m + l) and six of the nine piece products summed in fp32 (k_wino_gemm_b3)
    int narrow = 0;                              // split GEMM of a layer with Cout <= 32: one 32-column tile per wave, no LDS (k_wino_gemm_n32)
    const ThKnobs* knobs = nullptr;
    std::string label;
};
// split: 0 = fp32 MFMA (exact fp32 products), 1 = bf16x3 split operands (see ConvWinoPlan::split)
bool conv_wino_plan(const TView& in, const TView& out_conv, const ConvGeom& g, int Cin, int Cout, int scheme, ConvWinoPlan* plan, int split = 0);
void conv_wino_pack_weights(const ConvWinoPlan& p, const float* w_keras, float* dst);
// V, M: scratch for ceil(n / 64) * 64 frames (v_fpf / m_fpf floats each).  Three launches = three plan steps.
int launch_wino_in(hipStream_t s, int64_t n, const ConvWinoPlan& p, TView in, float* V, PreOp pre);
int launch_wino_gemm(hipStream_t s, int64_t n, const ConvWinoPlan& p, const float* V, float* M, const float* wpk);
// gap: `out` is the [n][Cout] tensor of a GlobalAveragePooling3D that is the layer's only reader (the 5^3 activation is not written)
int launch_wino_out(hipStream_t s, int64_t n, const ConvWinoPlan& p, const float* M, TView out, const float* bias, PostOps post, bool gap = false);
// launch_wino_out of layer p + launch_wino_in of the NEXT Winograd layer in one pass (the tensor between them is not written)
int launch_wino_mid(hipStream_t s, int64_t n, const ConvWinoPlan& p, const float* M, float* V_next, const float* bias, PostOps post);

// ---- fused Winograd convolution, everything of the transform domain in LDS (conv_wfused.hip): 3x3x3 'same' on 10^3 volumes ----
struct ConvWfPlan {
    int geo = -1;                    // index into the instantiated volume geometries
    int pool = 0;                    // 0 none, 1 max 2x2x2, 2 avg 2x2x2
    int Cin = 0, Cout = 0, ncb = 0, nchunks = 0;   // ncb: blocks of 16 output channels, nchunks: 4-channel input chunks
    size_t wpk_floats = 0, lds_bytes = 0;
    double own_flops = 0;            // the algorithm's multiply-adds x 2 per frame (16 positions x 3 z taps per 2 x 2 tile)
    double exec_flops = 0;           // MFMA FLOPs issued per frame
    const ThKnobs* knobs = nullptr;
    std::string label;
};
bool conv_wf_plan(const TView& in, const TView& out_conv, const ConvGeom& g, int Cin, int Cout, int pool, ConvWfPlan* plan);
bool conv_wf_view_ok(const TView& in);      // 16-byte aligned channel slices
int conv_wf_pre_kind(const PreOp& pre);
std::string conv_wf_label(const ConvWfPlan& p, const PreOp& pre);    // the plan's label with the kernel's full template argument list
void conv_wf_pack_weights(const ConvWfPlan& p, const float* w_keras, float* dst);
int launch_conv_wf(hipStream_t s, int64_t n, const ConvWfPlan& p, TView in, TView out, const float* wpk, const float* bias, PreOp pre,
                   PostOps post);

// the same layer with its products on bf16 MFMA, both operands split exactly into three bf16 pieces (conv_wfsplit.hip): no input
// prologue, Cin a multiple of 16, more than 32 output channels
struct ConvWfsPlan {
    int geo = -1, pool = 0;
    int Cin = 0, Cout = 0, nkh = 0, ncp = 0;       // nkh: phases of 16 input channels, ncp: passes of 64 output channels
    size_t wpk_floats = 0, lds_bytes = 0;
    double own_flops = 0, exec_flops = 0;
    const ThKnobs* knobs = nullptr;
    std::string label;
};
bool conv_wfs_plan(const ConvWfPlan& base, const TView& in, const PreOp& pre, ConvWfsPlan* plan);
void conv_wfs_pack_weights(const ConvWfsPlan& p, const float* w_keras, float* dst);
int launch_conv_wfs(hipStream_t s, int64_t n, const ConvWfsPlan& p, TView in, TView out, const float* wpk, const float* bias, PostOps post);

// ---- sparse float32 frames expanded on the device (sparse_frames.hip) ----
int launch_sparse_expand(hipStream_t s, int64_t n, const uint32_t* bits, const uint64_t* vidx, const float* values, float* out, int E, int W);

// ---- first-layer convolution (conv_first.hip): Cin <= 8, Cout <= 32, 3x3x3, reads the caller's frames ----
std::string conv_first_label(const ConvMfmaPlan& p, int Cin, const PostOps& post);
bool conv_first_plan(int Din, int Hin, int Win, int Cin, const TView& out_conv, const ConvGeom& g, int Cout, int pool,
                     ConvMfmaPlan* plan);
void conv_first_pack_weights(int Cin, int Cout, const float* w_keras, float* dst);
void conv_first_w_pack_weights(int Cin, int Cout, const float* w_keras, float* dst);    // plans with first_wino set
int launch_conv_first(hipStream_t s, int64_t n, const ConvMfmaPlan& p, const void* frames, int dtype, int Din, int Hin,
                      int Win, int Cin, TView out, ConvGeom g, int Cout, const float* wpk, const float* bias, PostOps post);
// the same layer on the bf16 pipe with split operands (conv_first_b3.hip): 21^3 x (5..6) frames, 'same', max-pool before a monotone chain
bool conv_first_b3_ok(const ConvMfmaPlan& p, int Din, int Hin, int Win, int Cin, int Cout, const ConvGeom& g, const PostOps& post);
size_t conv_first_b3_wpk_floats();
double conv_first_b3_exec_flops();
std::string conv_first_b3_label(int nnb);
void conv_first_b3_pack_weights(int Cin, int Cout, const float* w_keras, float* dst);
int launch_conv_first_b3(hipStream_t s, int64_t n, const ConvMfmaPlan& p, const void* frames, int dtype, int Cin, TView out, int Cout,
                         const float* wpk, const float* bias, PostOps post);

// 5x5x5 'same' convolution on the model input, <= 8 channels -> <= 16 filters, 2^3 max-pool behind it (conv_first5.hip): ProDCoNN's stem
bool conv_first5_ok(int Din, int Hin, int Win, int Cin, int Cout, const ConvGeom& g, int pool);
size_t conv_first5_wpk_floats();
double conv_first5_exec_flops();
std::string conv_first5_label();
void conv_first5_pack_weights(int Cin, int Cout, const float* w_keras, float* dst);
int launch_conv_first5(hipStream_t s, int64_t n, const ThKnobs* knobs, const void* frames, int dtype, int Cin, TView out, int Cout,
                       const float* wpk, const float* bias, PostOps post);

// ---- pointwise (1x1x1) streaming convolution (conv_pointwise.hip); plan.cfg in [300, 309) ----
bool conv_pw_plan(const TView& in, const TView& out_conv, const ConvGeom& g, int Cin, int Cout, int pool, ConvMfmaPlan* plan);
void conv_pw_pack_weights(const ConvMfmaPlan& p, int Cin, int Cout, const float* w_keras, float* dst);
std::string conv_pw_label(const ConvMfmaPlan& p, bool out_blk, const PostOps& post);
int launch_conv_pw(hipStream_t s, int64_t n, const ConvMfmaPlan& p, TView in, TView out, int Cin, int Cout, const float* wpk,
                   const float* bias, PreOp pre, PostOps post);

// ---- sampler (sampler.hip) -----------------------------------------------------------------
int sampler_run(int device, const double* h_probs, int64_t n_res, int n_cls, int64_t n_samples, double temperature,
                int rng_mode, uint64_t seed, uint64_t rng_offset, const double* h_uniforms, int32_t* h_idx, double* h_r_out,
                const char* cat_letters, char* h_letters, double* h_q_out);
