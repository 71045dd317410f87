// extern "C" entry points of the contraction family (conv2d, conv2d_transpose, linear)
// and the library-wide error channel.
#include <cstring>

#include "common.h"

namespace gan {

// igemm.hip
enum { MODE_FPROP = 0, MODE_TPROP = 1, MODE_WGRAD = 2 };
struct IgemmArgs;
int conv_geom_check(const gan_conv_geom* g);
int plan_set(int mode, const gan_conv_geom& g, int cfg, int splits);
void plan_get(int mode, const gan_conv_geom& g, int* cfg, int* splits, int* thin);
void plan_clear();
size_t igemm_ws_bytes_geom(int mode, const gan_conv_geom& g);
int run_igemm_geom(int mode, const gan_conv_geom& g, const float* a, const float* b, float* out, const float* bias,
                   int accumulate, int out_c, void* ws, size_t ws_bytes, hipStream_t s, const char* what);
// elementwise.hip
size_t colsum_ws_bytes(long rows, int c);
int colsum(const float* x, float* out, long rows, int c, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
           const char* what);

static thread_local char g_error[512] = "";

char* error_buffer() { return g_error; }

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return 1;
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

}  // namespace gan

using namespace gan;

extern "C" const char* gan_last_error(void) { return error_buffer(); }
extern "C" int gan_abi_version(void) { return 1; }

extern "C" size_t gan_conv_workspace_bytes(const gan_conv_geom* g) {
    if (conv_geom_check(g)) return 0;
    const long big_rows = (long)g->n * g->hb * g->wb, small_rows = (long)g->n * g->hs * g->ws;
    size_t w = max3(igemm_ws_bytes_geom(MODE_FPROP, *g), igemm_ws_bytes_geom(MODE_TPROP, *g),
                    igemm_ws_bytes_geom(MODE_WGRAD, *g));
    size_t b = colsum_ws_bytes(big_rows, g->cb);
    size_t b2 = colsum_ws_bytes(small_rows, g->cs);
    return max3(w, b, b2);
}

extern "C" int gan_conv2d_fwd(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                              void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(x && w && y, "gan_conv2d_fwd: NULL tensor");
    return run_igemm_geom(MODE_FPROP, *g, x, w, y, bias, 0, g->cs, ws, ws_bytes, as_stream(stream), "gan_conv2d_fwd");
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
    if (int rc = run_igemm_geom(MODE_WGRAD, *g, x, dy, dw, nullptr, accumulate, g->cs, ws, ws_bytes, as_stream(stream),
                                "gan_conv2d_wgrad"))
        return rc;
    if (db) return colsum(dy, db, (long)g->n * g->hs * g->ws, g->cs, accumulate, ws, ws_bytes, as_stream(stream), "gan_conv2d_wgrad");
    return 0;
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
    if (int rc = run_igemm_geom(MODE_WGRAD, *g, dy, x, dw, nullptr, accumulate, g->cs, ws, ws_bytes, as_stream(stream),
                                "gan_deconv2d_wgrad"))
        return rc;
    if (db) return colsum(dy, db, (long)g->n * g->hb * g->wb, g->cb, accumulate, ws, ws_bytes, as_stream(stream), "gan_deconv2d_wgrad");
    return 0;
}

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
    return run_igemm_geom(MODE_FPROP, linear_geom(rows, in_dim, out_dim), x, matrix, y, bias, 0, out_dim, ws, ws_bytes,
                          as_stream(stream), "gan_linear_fwd");
}

extern "C" int gan_linear_dgrad(int rows, int in_dim, int out_dim, const float* dy, const float* matrix, float* dx,
                                void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = linear_check("gan_linear_dgrad", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(dy && matrix && dx, "gan_linear_dgrad: NULL tensor");
    return run_igemm_geom(MODE_TPROP, linear_geom(rows, in_dim, out_dim), dy, matrix, dx, nullptr, 0, in_dim, ws, ws_bytes,
                          as_stream(stream), "gan_linear_dgrad");
}

extern "C" int gan_linear_wgrad(int rows, int in_dim, int out_dim, const float* x, const float* dy, float* dmatrix,
                                float* dbias, int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream) {
    if (int rc = linear_check("gan_linear_wgrad", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(x && dy && dmatrix, "gan_linear_wgrad: NULL tensor");
    if (int rc = run_igemm_geom(MODE_WGRAD, linear_geom(rows, in_dim, out_dim), x, dy, dmatrix, nullptr, accumulate, out_dim,
                                ws, ws_bytes, as_stream(stream), "gan_linear_wgrad"))
        return rc;
    if (dbias) return colsum(dy, dbias, rows, out_dim, accumulate, ws, ws_bytes, as_stream(stream), "gan_linear_wgrad");
    return 0;
}

// ---------------------------------------------------------------------------- plan pinning
static int plan_mode_check(const char* what, int mode) {
    GAN_CHECK_ARG(mode >= MODE_FPROP && mode <= MODE_WGRAD, "%s: mode must be 0 (fprop), 1 (tprop) or 2 (wgrad)", what);
    return 0;
}

extern "C" int gan_conv_plan_set(int mode, const gan_conv_geom* g, int cfg, int splits) {
    if (int rc = plan_mode_check("gan_conv_plan_set", mode)) return rc;
    if (int rc = conv_geom_check(g)) return rc;
    GAN_CHECK_ARG(cfg < 4, "gan_conv_plan_set: tile configuration must be 0..3 (or negative to unpin)");
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

extern "C" int gan_linear_as_conv_geom(int rows, int in_dim, int out_dim, gan_conv_geom* g) {
    if (int rc = linear_check("gan_linear_as_conv_geom", rows, in_dim, out_dim)) return rc;
    GAN_CHECK_ARG(g != nullptr, "gan_linear_as_conv_geom: NULL output");
    *g = linear_geom(rows, in_dim, out_dim);
    return 0;
}
