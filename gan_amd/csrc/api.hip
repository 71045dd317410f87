// extern "C" entry points of the contraction family (conv2d, conv2d_transpose, linear)
// and the library-wide error channel.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "bn_tile.h"

namespace gan {

// igemm.hip
enum { MODE_FPROP = 0, MODE_TPROP = 1, MODE_WGRAD = 2 };
struct IgemmArgs;
int conv_geom_check(const gan_conv_geom* g);
int plan_set(int mode, const gan_conv_geom& g, int cfg, int splits);
void plan_get(int mode, const gan_conv_geom& g, int* cfg, int* splits, int* thin);
void plan_clear();
void set_exact_f32(int on);
int get_exact_f32();
size_t igemm_ws_bytes_geom(int mode, const gan_conv_geom& g);
int run_igemm_geom(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                   int accumulate, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what,
                   const ColsumFinal* tail = nullptr, int* tail_done = nullptr);
size_t igemm_ws_bytes_dgrad_reflect(const gan_conv_geom& g);
int run_dgrad_reflect(const gan_conv_geom& g, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes,
                      hipStream_t s, const char* what);
int run_igemm_geom_reflect(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                           int accumulate, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what,
                           const ColsumFinal* tail = nullptr, int* tail_done = nullptr);
int run_igemm_geom_epi(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                       const Epilogue& epi, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what);
// elementwise.hip
size_t colsum_ws_bytes(long rows, int c);
int colsum_partial(const float* x, long rows, int c, void* ws, size_t ws_bytes, hipStream_t st, const char* what,
                   ColsumFinal* job);
int colsum_final(const ColsumFinal& job, hipStream_t st, const char* what);

static thread_local char g_error[512] = "";

char* error_buffer() { return g_error; }

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return 1;
}

static int env_int(const char* name, int unset) {
    const char* e = getenv(name);
    return e ? atoi(e) : unset;
}
static int env_flag(const char* name) {
    const char* e = getenv(name);
    return (e && *e && *e != '0') ? 1 : 0;
}

Knobs& knobs() {
    static Knobs k = [] {
        Knobs r;
        r.force_cfg = env_int("GAN_HIP_FORCE_CFG", -1);
        r.force_splits = env_int("GAN_HIP_FORCE_SPLITS", 0);
        if (r.force_splits < 0) r.force_splits = 0;
        r.no_thin = getenv("GAN_HIP_NO_THIN") != nullptr;
        r.slow_gather = getenv("GAN_HIP_SLOW_GATHER") != nullptr;
        r.scalar_epilogue = getenv("GAN_HIP_SCALAR_EPILOGUE") != nullptr;
        r.three_launch_bn = getenv("GAN_HIP_THREE_LAUNCH_BN") != nullptr;
        r.two_launch_bn = env_int("GAN_HIP_TWO_LAUNCH_BN", 1) != 0;
        r.separate_colsum = getenv("GAN_HIP_SEPARATE_COLSUM") != nullptr;
        r.tile_order = env_int("GAN_HIP_TILE_ORDER", 0);
        r.rgb_gather = env_int("GAN_HIP_RGB_GATHER", 1);
        r.row_gather_any = env_int("GAN_HIP_ROW_GATHER_ANY", 1);
        r.exact_f32 = env_flag("GAN_HIP_EXACT_F32");
        r.no_carry = getenv("GAN_HIP_NO_CARRY") != nullptr;
        r.no_bn_ride = env_int("GAN_HIP_NO_BN_RIDE", 0);
        r.rgb_direct = env_int("GAN_HIP_RGB_DIRECT", 1);
        return r;
    }();
    return k;
}

static gan_conv_geom linear_geom(int rows, int in_dim, int out_dim) {
    gan_conv_geom g;
    g.n = rows;
    g.hb = g.wb = g.hs = g.ws = 1;
    g.cb = in_dim;
    g.cs = out_dim;
    g.kh = g.kw = 1;
    g.sh = g.sw = 1;
    g.pt = g.pl = 0;
    return g;
}

static size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// Weight gradient + bias gradient of one layer.  The bias gradient is the column sum of `dy_rows` ([rows][c]).
// Its first stage runs first, into the workspace behind the split-K slabs; its last stage rides on the slab
// reduction's launch when there is one, so the pair costs one launch more than the weight gradient alone.
static int wgrad_with_bias(const gan_conv_geom& g, const float* a, const float* b, float* dw, float* db,
                           const float* dy_rows, long rows, int c, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                           const char* what) {
    if (db == nullptr)
        return run_igemm_geom(MODE_WGRAD, g, a, b, dw, nullptr, accumulate, g.cs, ws, ws_bytes, st, what);
    if (knobs().separate_colsum) {        // test knob: the column sum as two launches of its own, afterwards
        if (int rc = run_igemm_geom(MODE_WGRAD, g, a, b, dw, nullptr, accumulate, g.cs, ws, ws_bytes, st, what)) return rc;
        ColsumFinal sep;
        if (int rc = colsum_partial(dy_rows, rows, c, ws, ws_bytes, st, what, &sep)) return rc;
        sep.out = db;
        sep.accumulate = accumulate;
        return colsum_final(sep, st, what);
    }
    const size_t off = align256(igemm_ws_bytes_geom(MODE_WGRAD, g));
    GAN_CHECK_ARG(ws && ws_bytes >= off, "%s: workspace of %zu bytes needed, %zu given", what, off, ws_bytes);
    ColsumFinal job;
    if (int rc = colsum_partial(dy_rows, rows, c, static_cast<char*>(ws) + off, ws_bytes - off, st, what, &job)) return rc;
    job.out = db;
    job.accumulate = accumulate;
    int done = 0;
    // the slabs may use only the `off` bytes in front of the partial sums: a plan that needs more (it cannot,
    // the size query covers the aligned and the unaligned plan) fails instead of overwriting them
    if (int rc = run_igemm_geom(MODE_WGRAD, g, a, b, dw, nullptr, accumulate, g.cs, ws, off, st, what, &job, &done))
        return rc;
    return done ? 0 : colsum_final(job, st, what);
}

}  // namespace gan

using namespace gan;

extern "C" const char* gan_last_error(void) { return error_buffer(); }
extern "C" int gan_abi_version(void) { return 1; }

extern "C" size_t gan_conv_workspace_bytes(const gan_conv_geom* g) {
    if (conv_geom_check(g)) return 0;
    const long big_rows = (long)g->n * g->hb * g->wb, small_rows = (long)g->n * g->hs * g->ws;
    size_t b = colsum_ws_bytes(big_rows, g->cb);
    size_t b2 = colsum_ws_bytes(small_rows, g->cs);
    // the weight gradient keeps its slabs and the bias gradient's partial sums side by side
    return max3(igemm_ws_bytes_geom(MODE_FPROP, *g), igemm_ws_bytes_geom(MODE_TPROP, *g),
                align256(igemm_ws_bytes_geom(MODE_WGRAD, *g)) + (b > b2 ? b : b2));
}

extern "C" int gan_conv2d_fwd(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                              void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_conv2d_fwd: NULL tensor");
    return run_igemm_geom(MODE_FPROP, *g, x, w, y, bias, 0, g->cs, ws, ws_bytes, as_stream(stream), "gan_conv2d_fwd");
}

// ---------------------------------------------------------------------------- REFLECT padding folded into the gather
extern "C" int gan_conv2d_fwd_reflect(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                      void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_conv2d_fwd_reflect: NULL tensor");
    return run_igemm_geom_reflect(MODE_FPROP, *g, x, w, y, bias, 0, g->cs, ws, ws_bytes, as_stream(stream),
                                  "gan_conv2d_fwd_reflect");
}

extern "C" int gan_conv2d_wgrad_reflect(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                                        int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && dy && dw, "gan_conv2d_wgrad_reflect: NULL tensor");
    hipStream_t st = as_stream(stream);
    const char* what = "gan_conv2d_wgrad_reflect";
    if (db == nullptr)
        return run_igemm_geom_reflect(MODE_WGRAD, *g, x, dy, dw, nullptr, accumulate, g->cs, ws, ws_bytes, st, what);
    // bias gradient as in wgrad_with_bias: first stage ahead of the contraction, last stage on its slab reduction
    const size_t off = align256(igemm_ws_bytes_geom(MODE_WGRAD, *g));
    GAN_CHECK_ARG(ws && ws_bytes >= off, "%s: workspace of %zu bytes needed, %zu given", what, off, ws_bytes);
    ColsumFinal job;
    if (int rc = colsum_partial(dy, (long)g->n * g->hs * g->ws, g->cs, static_cast<char*>(ws) + off, ws_bytes - off, st, what, &job))
        return rc;
    job.out = db;
    job.accumulate = accumulate;
    int done = 0;
    if (int rc = run_igemm_geom_reflect(MODE_WGRAD, *g, x, dy, dw, nullptr, accumulate, g->cs, ws, off, st, what, &job, &done))
        return rc;
    return done ? 0 : colsum_final(job, st, what);
}

extern "C" size_t gan_conv2d_dgrad_reflect_workspace_bytes(const gan_conv_geom* g) {
    if (conv_geom_check(g)) return 0;
    return igemm_ws_bytes_dgrad_reflect(*g);
}

extern "C" int gan_conv2d_dgrad_reflect(const gan_conv_geom* g, const float* dy, const float* w, float* dx, void* ws,
                                        size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(dy && w && dx, "gan_conv2d_dgrad_reflect: NULL tensor");
    return run_dgrad_reflect(*g, dy, w, dx, ws, ws_bytes, as_stream(stream), "gan_conv2d_dgrad_reflect");
}

// ---------------------------------------------------------------------------- fused activation epilogues
static int act_check(const char* what, int act, float scale) {
    GAN_CHECK_ARG(act >= GAN_ACT_NONE && act <= GAN_ACT_SIGMOID, "%s: unknown activation %d", what, act);
    GAN_CHECK_ARG(scale != 0.f, "%s: activation scale must not be 0", what);
    return 0;
}

extern "C" int gan_conv2d_fwd_act(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                  int act, float act_param, float act_scale, float act_shift, void* ws, size_t ws_bytes,
                                  gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    if (int rc = act_check("gan_conv2d_fwd_act", act, act_scale)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_conv2d_fwd_act: NULL tensor");
    return run_igemm_geom_epi(MODE_FPROP, *g, x, w, y, bias, Epilogue{act, act_param, act_scale, act_shift, nullptr}, g->cs, ws,
                              ws_bytes, as_stream(stream), "gan_conv2d_fwd_act");
}

extern "C" int gan_deconv2d_fwd_act(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                    int act, float act_param, float act_scale, float act_shift, void* ws, size_t ws_bytes,
                                    gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    if (int rc = act_check("gan_deconv2d_fwd_act", act, act_scale)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_deconv2d_fwd_act: NULL tensor");
    return run_igemm_geom_epi(MODE_TPROP, *g, x, w, y, bias, Epilogue{act, act_param, act_scale, act_shift, nullptr}, g->cb, ws,
                              ws_bytes, as_stream(stream), "gan_deconv2d_fwd_act");
}

extern "C" int gan_conv2d_dgrad_gated(const gan_conv_geom* g, const float* dy, const float* w, float* dx, const float* gate,
                                      int act, float act_param, float act_scale, float act_shift, void* ws,
                                      size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    if (int rc = act_check("gan_conv2d_dgrad_gated", act, act_scale)) return rc;
    GAN_CHECK_ARG(dy && w && dx && gate, "gan_conv2d_dgrad_gated: NULL tensor");
    return run_igemm_geom_epi(MODE_TPROP, *g, dy, w, dx, nullptr, Epilogue{act, act_param, act_scale, act_shift, gate}, g->cb,
                              ws, ws_bytes, as_stream(stream), "gan_conv2d_dgrad_gated");
}

extern "C" int gan_deconv2d_dgrad_gated(const gan_conv_geom* g, const float* dy, const float* w, float* dx, const float* gate,
                                        int act, float act_param, float act_scale, float act_shift, void* ws,
                                        size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    if (int rc = act_check("gan_deconv2d_dgrad_gated", act, act_scale)) return rc;
    GAN_CHECK_ARG(dy && w && dx && gate, "gan_deconv2d_dgrad_gated: NULL tensor");
    return run_igemm_geom_epi(MODE_FPROP, *g, dy, w, dx, nullptr, Epilogue{act, act_param, act_scale, act_shift, gate}, g->cs,
                              ws, ws_bytes, as_stream(stream), "gan_deconv2d_dgrad_gated");
}

// ---------------------------------------------------------------------------- entry point that allocates nothing
// The split-K slabs of gan_conv2d_fwd normally live in the caller's workspace.  This variant keeps them in a
// device array that is part of the library's code object, so a caller that cannot allocate (HIP graph capture,
// a serving loop, a harness that forbids hipMalloc) needs no workspace at all.  Every GPU the library runs on
// holds its own copy of the array at its own address until the process ends; the array's name is only a
// host-side handle, so the address is asked for once per device and used from then on.
namespace gan {
constexpr size_t kScratchBytes = 64u << 20;
constexpr int kScratchMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
static std::atomic<void*> g_scratch_ptr[kScratchMaxDevices];

// Device address of `scratch` on the current device: looked up at the first call on that device, read without a lock
// from then on (two threads that race on the first call ask twice and store the same address).
static int scratch_address(void** out, const char* what) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    GAN_CHECK_ARG(e == hipSuccess, "%s: hipGetDevice failed: %s", what, hipGetErrorString(e));
    GAN_CHECK_ARG(dev >= 0 && dev < kScratchMaxDevices, "%s: device index %d outside 0..%d", what, dev, kScratchMaxDevices - 1);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        GAN_CHECK_ARG(e == hipSuccess && p != nullptr, "%s: hipGetSymbolAddress(scratch) failed on device %d: %s", what, dev,
                      hipGetErrorString(e));
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return 0;
}
}  // namespace gan

extern "C" int gan_conv2d_fwd_static_ws(const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                                        float* y, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_conv2d_fwd_static_ws: NULL tensor");
    const size_t need = igemm_ws_bytes_geom(MODE_FPROP, *g);
    GAN_CHECK_ARG(need <= kScratchBytes,
                  "gan_conv2d_fwd_static_ws: this geometry needs %zu bytes of split-K slabs, the built-in scratch array "
                  "holds %zu; use gan_conv2d_fwd with a workspace", need, (size_t)kScratchBytes);
    void* ws = nullptr;
    if (int rc = scratch_address(&ws, "gan_conv2d_fwd_static_ws")) return rc;
    // every slab element a reduction reads is written by the contraction launch before it: nothing has to start at zero
    return run_igemm_geom(MODE_FPROP, *g, x, w, y, bias, 0, g->cs, ws, kScratchBytes, as_stream(stream),
                          "gan_conv2d_fwd_static_ws");
}

extern "C" int gan_conv2d_dgrad(const gan_conv_geom* g, const float* dy, const float* w, float* dx, void* ws,
                                size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(dy && w && dx, "gan_conv2d_dgrad: NULL tensor");
    return run_igemm_geom(MODE_TPROP, *g, dy, w, dx, nullptr, 0, g->cb, ws, ws_bytes, as_stream(stream), "gan_conv2d_dgrad");
}

extern "C" int gan_conv2d_wgrad(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                                int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && dy && dw, "gan_conv2d_wgrad: NULL tensor");
    return wgrad_with_bias(*g, x, dy, dw, db, dy, (long)g->n * g->hs * g->ws, g->cs, accumulate, ws, ws_bytes,
                           as_stream(stream), "gan_conv2d_wgrad");
}

extern "C" int gan_deconv2d_fwd(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_deconv2d_fwd: NULL tensor");
    return run_igemm_geom(MODE_TPROP, *g, x, w, y, bias, 0, g->cb, ws, ws_bytes, as_stream(stream), "gan_deconv2d_fwd");
}

extern "C" int gan_deconv2d_dgrad(const gan_conv_geom* g, const float* dy, const float* w, float* dx, void* ws,
                                  size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(dy && w && dx, "gan_deconv2d_dgrad: NULL tensor");
    return run_igemm_geom(MODE_FPROP, *g, dy, w, dx, nullptr, 0, g->cs, ws, ws_bytes, as_stream(stream), "gan_deconv2d_dgrad");
}

extern "C" int gan_deconv2d_wgrad(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                                  int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && dy && dw, "gan_deconv2d_wgrad: NULL tensor");
    // dy (big side) plays the conv input, x (small side) the conv output gradient
    return wgrad_with_bias(*g, dy, x, dw, db, dy, (long)g->n * g->hb * g->wb, g->cb, accumulate, ws, ws_bytes,
                           as_stream(stream), "gan_deconv2d_wgrad");
}

// ---------------------------------------------------------------------------- batch norm riding on a contraction
// A conv / deconv in front of a training-mode batch norm leaves the norm's block statistics behind; an input gradient
// that feeds a batch norm's backward leaves that backward's block sums behind.  In both cases the reduction runs in
// the launch that sums the contraction's split-K slabs (one launch instead of two, and the finished tensor is not
// read back for it); without a slab reduction it is a launch of its own behind the contraction, so the caller finds
// the partial results either way.
static int bn_ride_shape(const char* what, int op, const gan_conv_geom& g, bool output_big, int groups, bool backward, TileBn* ts) {
    GAN_CHECK_ARG(op >= GAN_OP_CONV2D && op <= GAN_OP_CONV2D_REFLECT, "%s: unknown op %d", what, op);
    const long rows_all = output_big ? (long)g.n * g.hb * g.wb : (long)g.n * g.hs * g.ws;
    const int c = output_big ? g.cb : g.cs;
    GAN_CHECK_ARG(groups > 0 && rows_all % groups == 0, "%s: %ld output rows do not split into %d groups", what, rows_all, groups);
    GAN_CHECK_ARG(rows_all / groups < (1L << 31) && bn_ride_plan(groups, (int)(rows_all / groups), c, backward, ts),
                  "%s: no block decomposition exists for %d x %ld x %d (gan_bn_tile_blocks reports 0)", what, groups,
                  rows_all / groups, c);
    return 0;
}

extern "C" int gan_conv_fwd_bn_stats(int op, const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                                     float* y, int groups, float* bn_partial, void* ws, size_t ws_bytes,
                                     gan_stream_t stream) {
    const char* what = "gan_conv_fwd_bn_stats";
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(bn_partial != nullptr, "%s: NULL tensor", what);
    TileBn ts;
    if (int rc = bn_ride_shape(what, op, *g, op == GAN_OP_DECONV2D, groups, false, &ts)) return rc;
    pending_bn_ride() = BnRide{1, reinterpret_cast<float2*>(bn_partial), ts, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f};
    const int rc = op == GAN_OP_CONV2D ? gan_conv2d_fwd(g, x, w, bias, y, ws, ws_bytes, stream)
                 : op == GAN_OP_DECONV2D ? gan_deconv2d_fwd(g, x, w, bias, y, ws, ws_bytes, stream)
                                         : gan_conv2d_fwd_reflect(g, x, w, bias, y, ws, ws_bytes, stream);
    pending_bn_ride().kind = 0;   // (a call that failed its argument checks never took it)
    return rc;
}

extern "C" int gan_conv_dgrad_bn_sums(int op, const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                                      const gan_bn_saved* bn, float* bn_partial, void* ws, size_t ws_bytes,
                                      gan_stream_t stream) {
    const char* what = "gan_conv_dgrad_bn_sums";
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(bn != nullptr && bn_partial != nullptr && bn->x && bn->gamma && bn->beta && bn->save_mean && bn->save_rstd,
                  "%s: NULL tensor", what);
    GAN_CHECK_ARG(bn->act >= GAN_ACT_NONE && bn->act <= GAN_ACT_SIGMOID, "%s: unknown activation %d", what, bn->act);
    TileBn ts;
    if (int rc = bn_ride_shape(what, op, *g, op != GAN_OP_DECONV2D, bn->groups, true, &ts)) return rc;
    pending_bn_ride() = BnRide{2, reinterpret_cast<float2*>(bn_partial), ts, bn->x, bn->gamma, bn->beta, bn->save_mean,
                               bn->save_rstd, bn->act, bn->act_param};
    const int rc = op == GAN_OP_CONV2D ? gan_conv2d_dgrad(g, dy, w, dx, ws, ws_bytes, stream)
                 : op == GAN_OP_DECONV2D ? gan_deconv2d_dgrad(g, dy, w, dx, ws, ws_bytes, stream)
                                         : gan_conv2d_dgrad_reflect(g, dy, w, dx, ws, ws_bytes, stream);
    pending_bn_ride().kind = 0;
    return rc;
}

// out = a + b with the block sums of a batch norm's backward over `out`: the two addends are read as two "slabs" whose
// distance is the distance of the two tensors (the slab-summing variant of the sums kernel adds slab 0 + slab 1, which
// is the add kernel's a + b), every sum is stored to `out` and taken into the reduction.
extern "C" int gan_add_bn_sums(const float* a, const float* b, float* out, const gan_bn_saved* bn, int rows, int c,
                               float* bn_partial, gan_stream_t stream) {
    const char* what = "gan_add_bn_sums";
    GAN_CHECK_ARG(a && b && out && bn != nullptr && bn_partial != nullptr && bn->x && bn->gamma && bn->beta && bn->save_mean &&
                  bn->save_rstd, "%s: NULL tensor", what);
    GAN_CHECK_ARG(bn->act >= GAN_ACT_NONE && bn->act <= GAN_ACT_SIGMOID, "%s: unknown activation %d", what, bn->act);
    GAN_CHECK_ARG(bn->groups > 0 && rows > 0 && c > 0, "%s: groups, rows and c must be positive", what);
    TileBn ts;
    GAN_CHECK_ARG(bn_ride_plan(bn->groups, rows, c, true, &ts),
                  "%s: no block decomposition exists for %d x %d x %d (gan_bn_tile_blocks reports 0)", what, bn->groups, rows, c);
    GAN_CHECK_ARG((reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0,
                  "%s: the addends must be 16-byte aligned", what);
    const BnRide ride = BnRide{2, reinterpret_cast<float2*>(bn_partial), ts, bn->x, bn->gamma, bn->beta, bn->save_mean,
                               bn->save_rstd, bn->act, bn->act_param};
    const long apart = (long)((reinterpret_cast<intptr_t>(b) - reinterpret_cast<intptr_t>(a)) / (intptr_t)sizeof(float));
    return launch_bn_ride(ride, RowSource{a, out, nullptr, apart, 2}, as_stream(stream), what);
}

extern "C" int gan_conv_wgrad_colsum(int op, const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                                     int accumulate, const float* colsum_partial, int colsum_blocks, void* ws,
                                     size_t ws_bytes, gan_stream_t stream) {
    const char* what = "gan_conv_wgrad_colsum";
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(op >= GAN_OP_CONV2D && op <= GAN_OP_CONV2D_REFLECT, "%s: unknown op %d", what, op);
    GAN_CHECK_ARG(x && dy && dw && db && colsum_partial && colsum_blocks > 0, "%s: NULL tensor", what);
    hipStream_t st = as_stream(stream);
    const int c = op == GAN_OP_DECONV2D ? g->cb : g->cs;
    const ColsumFinal job = ColsumFinal{colsum_partial, db, c, colsum_blocks, accumulate, (c + 3) / 4};
    int done = 0, rc;
    if (op == GAN_OP_CONV2D) rc = run_igemm_geom(MODE_WGRAD, *g, x, dy, dw, nullptr, accumulate, g->cs, ws, ws_bytes, st, what, &job, &done);
    else if (op == GAN_OP_DECONV2D) rc = run_igemm_geom(MODE_WGRAD, *g, dy, x, dw, nullptr, accumulate, g->cs, ws, ws_bytes, st, what, &job, &done);
    else rc = run_igemm_geom_reflect(MODE_WGRAD, *g, x, dy, dw, nullptr, accumulate, g->cs, ws, ws_bytes, st, what, &job, &done);
    if (rc) return rc;
    return done ? 0 : colsum_final(job, st, what);
}

// ---------------------------------------------------------------------------- thin linear
// ops.linear onto a handful of outputs (the discriminator's final logit: 8192 -> 1) is three
// matrix-vector products; the 32-wide MFMA tile would be almost all padding.  They are plain
// bandwidth kernels instead.  OUT <= 4.
namespace gan {

constexpr int kThinLinMaxOut = 4;

// y[r][o] = sum_i x[r][i] * w[i][o] + bias[o]; one block per row
template <int OUT>
__global__ __launch_bounds__(256) void linear_thin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int in_dim, int vec) {
    __shared__ float red[4][OUT];
    const int r = blockIdx.x;
    const float* xr = x + (long)r * in_dim;
    float acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = 0.f;
    if (OUT == 1 && vec) {
        // one output, in_dim % 4 == 0, 16-byte aligned row and vector: 16 bytes of each per lane and step
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        const float4* w4 = reinterpret_cast<const float4*>(w);
#pragma unroll 8
        for (int i = threadIdx.x; i < (in_dim >> 2); i += 256) {
            const float4 xv = x4[i], wv = w4[i];
            acc[0] = fmaf(xv.w, wv.w, fmaf(xv.z, wv.z, fmaf(xv.y, wv.y, fmaf(xv.x, wv.x, acc[0]))));
        }
    } else {
        // (unrolled: the loads of eight steps are in flight together; the sums keep their order)
#pragma unroll 8
        for (int i = threadIdx.x; i < in_dim; i += 256) {
            const float xv = xr[i];
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] = fmaf(xv, w[(long)i * OUT + o], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        const float v = wave_sum(acc[o]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][o] = v;
    }
    __syncthreads();
    if (threadIdx.x < OUT) {
        const int o = threadIdx.x;
        y[(long)r * OUT + o] = red[0][o] + red[1][o] + red[2][o] + red[3][o] + (bias ? bias[o] : 0.f);
    }
}

// dx[r][i] = sum_o dy[r][o] * w[i][o]
template <int OUT>
__global__ void linear_thin_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                         int rows, int in_dim) {
    const long total = (long)rows * in_dim;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / in_dim), i = (int)(e - (long)r * in_dim);
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < OUT; ++o) v = fmaf(dy[(long)r * OUT + o], w[(long)i * OUT + o], v);
        dx[e] = v;
    }
}

// dw[i][o] (+)= sum_r x[r][i] * dy[r][o]; db[o] (+)= sum_r dy[r][o].  A block owns 64 input indices; its four waves
// each sum every fourth row in order (256-byte row segments per wave) and wave 0 adds the four partial sums in
// wave order: in_dim / 64 blocks instead of in_dim / 256, and a quarter of the dependent loads per thread.
template <int OUT>
__global__ __launch_bounds__(256) void linear_thin_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ db, int rows,
                                                                int in_dim, int accumulate) {
    __shared__ float red[4][OUT][64];
    const int il = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = 0.f;
    if (i < in_dim) {
#pragma unroll 8
        for (int r = rl; r < rows; r += 4) {
            const float xv = x[(long)r * in_dim + i];
#pragma unroll
            for (int o = 0; o < OUT; ++o) acc[o] = fmaf(xv, dy[(long)r * OUT + o], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) red[rl][o][il] = acc[o];
    __syncthreads();
    if (rl == 0 && i < in_dim) {
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            float* d = dw + (long)i * OUT + o;
            *d = (accumulate ? *d : 0.f) + (((red[0][o][il] + red[1][o][il]) + red[2][o][il]) + red[3][o][il]);
        }
    }
    if (db != nullptr && blockIdx.x == 0 && threadIdx.x < OUT) {
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < rows; ++r) sum += dy[(long)r * OUT + threadIdx.x];
        db[threadIdx.x] = (accumulate ? db[threadIdx.x] : 0.f) + sum;
    }
}

static bool thin_linear_ok(int in_dim, int out_dim) {
    return out_dim <= kThinLinMaxOut && in_dim >= 256 && !knobs().no_thin;
}

#define GAN_THIN_LIN_DISPATCH(KERNEL, GRID, BLOCK, ...)                                            \
    switch (out_dim) {                                                                             \
        case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, 0, st, __VA_ARGS__); break;             \
        case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, 0, st, __VA_ARGS__); break;             \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, 0, st, __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, 0, st, __VA_ARGS__); break;            \
    }

}  // namespace gan

static int linear_check(const char* what, int rows, int in_dim, int out_dim) {
    GAN_CHECK_ARG(rows > 0 && in_dim > 0 && out_dim > 0, "%s: rows, in_dim and out_dim must be positive", what);
    return 0;
}

extern "C" size_t gan_linear_workspace_bytes(int rows, int in_dim, int out_dim) {
    if (rows <= 0 || in_dim <= 0 || out_dim <= 0) return 0;
    gan_conv_geom g = linear_geom(rows, in_dim, out_dim);
    return gan_conv_workspace_bytes(&g);
}

extern "C" int gan_linear_fwd(int rows, int in_dim, int out_dim, const float* x, const float* matrix,
                              const float* bias, float* y, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = linear_check("gan_linear_fwd", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(x && matrix && y, "gan_linear_fwd: NULL tensor");
    if (thin_linear_ok(in_dim, out_dim)) {
        hipStream_t st = as_stream(stream);
        const int vec = out_dim == 1 && in_dim % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)matrix % 16 == 0);
        GAN_THIN_LIN_DISPATCH(linear_thin_fwd_kernel, dim3(rows), dim3(256), x, matrix, bias, y, in_dim, vec);
        return check_launch("gan_linear_fwd");
    }
    return run_igemm_geom(MODE_FPROP, linear_geom(rows, in_dim, out_dim), x, matrix, y, bias, 0, out_dim, ws, ws_bytes,
                          as_stream(stream), "gan_linear_fwd");
}

extern "C" int gan_linear_dgrad(int rows, int in_dim, int out_dim, const float* dy, const float* matrix, float* dx,
                                void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = linear_check("gan_linear_dgrad", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(dy && matrix && dx, "gan_linear_dgrad: NULL tensor");
    if (thin_linear_ok(in_dim, out_dim)) {
        hipStream_t st = as_stream(stream);
        long blocks = ((long)rows * in_dim + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        GAN_THIN_LIN_DISPATCH(linear_thin_dgrad_kernel, dim3((unsigned)blocks), dim3(256), dy, matrix, dx, rows, in_dim);
        return check_launch("gan_linear_dgrad");
    }
    return run_igemm_geom(MODE_TPROP, linear_geom(rows, in_dim, out_dim), dy, matrix, dx, nullptr, 0, in_dim, ws, ws_bytes,
                          as_stream(stream), "gan_linear_dgrad");
}

extern "C" int gan_linear_wgrad(int rows, int in_dim, int out_dim, const float* x, const float* dy, float* dmatrix,
                                float* dbias, int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = linear_check("gan_linear_wgrad", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(x && dy && dmatrix, "gan_linear_wgrad: NULL tensor");
    if (thin_linear_ok(in_dim, out_dim)) {
        hipStream_t st = as_stream(stream);
        GAN_THIN_LIN_DISPATCH(linear_thin_wgrad_kernel, dim3((in_dim + 63) / 64), dim3(256), x, dy, dmatrix, dbias, rows,
                              in_dim, accumulate);
        return check_launch("gan_linear_wgrad");
    }
    return wgrad_with_bias(linear_geom(rows, in_dim, out_dim), x, dy, dmatrix, dbias, dy, rows, out_dim, accumulate, ws,
                           ws_bytes, as_stream(stream), "gan_linear_wgrad");
}

// ---------------------------------------------------------------------------- plan pinning
static int plan_mode_check(const char* what, int mode) {
    GAN_CHECK_ARG(mode >= MODE_FPROP && mode <= MODE_WGRAD, "%s: mode must be 0 (fprop), 1 (tprop) or 2 (wgrad)", what);
    return 0;
}

extern "C" int gan_conv_plan_set(int mode, const gan_conv_geom* g, int cfg, int splits) {
    if (int rc = plan_mode_check("gan_conv_plan_set", mode)) return rc;
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(cfg < 0 || (cfg % 16 < 10 && cfg / 16 <= 4),
                  "gan_conv_plan_set: cfg must be a tile configuration 0..9 plus 16 x a tile order 0..4 (or negative to unpin)");
    GAN_CHECK_ARG(cfg < 0 || (splits >= 1 && splits <= 512), "gan_conv_plan_set: split factor must be in 1..512");
    return plan_set(mode, *g, cfg, splits);
}

extern "C" int gan_conv_plan_get(int mode, const gan_conv_geom* g, int* cfg, int* splits, int* thin) {
    if (int rc = plan_mode_check("gan_conv_plan_get", mode)) return rc;
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(cfg && splits && thin, "gan_conv_plan_get: NULL output");
    plan_get(mode, *g, cfg, splits, thin);
    return 0;
}

extern "C" void gan_conv_plan_clear(void) { plan_clear(); }

extern "C" int gan_knob_set(const char* name, int value) {
    GAN_CHECK_ARG(name != nullptr, "gan_knob_set: NULL name");
    Knobs& k = knobs();
    if (!strcmp(name, "force_cfg")) k.force_cfg = value < 0 ? -1 : value;
    else if (!strcmp(name, "force_splits")) k.force_splits = value > 0 ? value : 0;
    else if (!strcmp(name, "no_thin")) k.no_thin = value != 0;
    else if (!strcmp(name, "slow_gather")) k.slow_gather = value != 0;
    else if (!strcmp(name, "scalar_epilogue")) k.scalar_epilogue = value != 0;
    else if (!strcmp(name, "three_launch_bn")) k.three_launch_bn = value != 0;
    else if (!strcmp(name, "two_launch_bn")) k.two_launch_bn = value != 0;
    else if (!strcmp(name, "separate_colsum")) k.separate_colsum = value != 0;
    else if (!strcmp(name, "tile_order")) k.tile_order = value;
    else if (!strcmp(name, "rgb_gather")) k.rgb_gather = value;
    else if (!strcmp(name, "row_gather_any")) k.row_gather_any = value;
    else if (!strcmp(name, "no_carry")) k.no_carry = value != 0;
    else if (!strcmp(name, "no_bn_ride")) k.no_bn_ride = value;
    else if (!strcmp(name, "rgb_direct")) k.rgb_direct = value != 0;
    else return fail("gan_knob_set: unknown knob '%s'", name);
    return 0;
}

extern "C" void gan_set_exact_f32(int on) { set_exact_f32(on); }
extern "C" int gan_get_exact_f32(void) { return get_exact_f32(); }

extern "C" int gan_linear_as_conv_geom(int rows, int in_dim, int out_dim, gan_conv_geom* g) {
    if (int rc = linear_check("gan_linear_as_conv_geom", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(g != nullptr, "gan_linear_as_conv_geom: NULL output");
    *g = linear_geom(rows, in_dim, out_dim);
    return 0;
}
