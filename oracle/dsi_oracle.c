/*
 * dsi_oracle.c -- CPU ORACLE (test infrastructure only; PARITY UNPINNED, see
 * dsi_oracle.h).  Plain C, fp32 arithmetic written operation by operation so
 * that, built with -ffp-contract=off and without -ffast-math, every rounding
 * happens where the reference's SSE2 build rounds.
 */
#include "dsi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- depth planes: depth_vector.hpp:76-163 ------------------------------ */
void orc_depth_planes(float min_depth, float max_depth, int nz, int inverse,
                      float *raw_depths)
{
    /* depth_vector.hpp:33-36: swap when given in the wrong order */
    if (min_depth > max_depth) {
        float tmp = min_depth;
        min_depth = max_depth;
        max_depth = tmp;
    }
    if (!inverse) {
        /* :88  mult = (float)(N / (max-min)), size_t/float -> float */
        const float mult = (float)nz / (max_depth - min_depth);
        for (int i = 0; i < nz; ++i) /* :93 ; cellIndexToDepth :100-103 */
            raw_depths[i] = min_depth + (float)i / mult;
    } else {
        /* :131-140 */
        const float inv_min = 1.f / min_depth;
        const float inv_max = 1.f / max_depth;
        const float mult = (float)nz / (inv_min - inv_max);
        for (int i = 0; i < nz; ++i) {
            const float rho = inv_max + (float)i / mult;
            raw_depths[i] = 1.f / rho; /* :145-148 */
        }
    }
}

/* ---- virtual camera focal: mapper_emvs_stereo.cpp:219-229 --------------- */
float orc_virtual_focal(float cam_fx, float fov_deg, int dim_x)
{
    if (fov_deg < 10.f)
        return cam_fx;
    /* const float dsi_fov_rad = fov * CV_PI / 180.0 (double math, float store) */
    const float dsi_fov_rad = (float)((double)fov_deg * 3.1415926535897932384626433832795 / 180.0);
    /* f = 0.5 * (float)dimX / std::tan(0.5 * dsi_fov_rad)  (double, float store) */
    return (float)(0.5 * (double)(float)dim_x / tan(0.5 * (double)dsi_fov_rad));
}

/* ---- Eigen 3.3 fixed-size helpers (restated; see header) ---------------- */
static inline float dot3_eigen(float a0, float b0, float a1, float b1, float a2, float b2)
{
    /* redux_novec_unroller<.,.,0,3>: x0 + (x1 + x2) */
    const float x0 = a0 * b0, x1 = a1 * b1, x2 = a2 * b2;
    return x0 + (x1 + x2);
}

static void mul3x3_eigen(const float *a, const float *b, float *c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = dot3_eigen(a[3 * i + 0], b[0 + j], a[3 * i + 1], b[3 + j],
                                      a[3 * i + 2], b[6 + j]);
}

static inline float cof3(const float *m, int i, int j)
{
    /* Eigen cofactor_3x3<i,j> */
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

void orc_inverse3x3(const float *m, float *out)
{
    /* Eigen compute_inverse<Matrix3f,3> / compute_inverse_size3_helper */
    const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    const float det = dot3_eigen(c0, m[0], c1, m[3], c2, m[6]);
    const float invdet = 1.f / det;
    out[0] = c0 * invdet;
    out[1] = c1 * invdet;
    out[2] = c2 * invdet;
    out[3] = cof3(m, 0, 1) * invdet;
    out[4] = cof3(m, 1, 1) * invdet;
    out[5] = cof3(m, 2, 1) * invdet;
    out[6] = cof3(m, 0, 2) * invdet;
    out[7] = cof3(m, 1, 2) * invdet;
    out[8] = cof3(m, 2, 2) * invdet;
}

/* ---- per-packet geometry: mapper_emvs_stereo.cpp:101-126 ---------------- */
void orc_packet_geometry(const float *Rt, const float *K, const float *Kv, float z0,
                         float *center, float *H)
{
    const float *R = Rt, *t = Rt + 9;
    /* :108  camera_centers.push_back(-R.transpose() * t) */
    for (int i = 0; i < 3; ++i)
        center[i] = dot3_eigen(-R[0 + i], t[0], -R[3 + i], t[1], -R[6 + i], t[2]);

    /* :114-116  H_z0_inv = R; H_z0_inv *= z0; H_z0_inv.col(2) += t */
    float Hinv[9];
    for (int k = 0; k < 9; ++k)
        Hinv[k] = R[k] * z0;
    Hinv[2] += t[0];
    Hinv[5] += t[1];
    Hinv[8] += t[2];

    /* :46-48 K_ ; geometry_utils.hpp:43-47 virtual K and Kinv_ = K_.inverse() */
    const float Km[9] = {K[0], 0.f, K[2], 0.f, K[1], K[3], 0.f, 0.f, 1.f};
    const float Kvm[9] = {Kv[0], 0.f, Kv[2], 0.f, Kv[1], Kv[3], 0.f, 0.f, 1.f};
    float Kvinv[9];
    orc_inverse3x3(Kvm, Kvinv);

    /* :119  H_z0_inv_px = K_ * H_z0_inv * Kinv_  (left to right) */
    float M1[9], M[9];
    mul3x3_eigen(Km, Hinv, M1);
    mul3x3_eigen(M1, Kvinv, M);
    /* :120  H_z0_px = H_z0_inv_px.inverse() */
    orc_inverse3x3(M, H);
}

/* ---- per-event z0 warp: mapper_emvs_stereo.cpp:129-142 ------------------ */
void orc_warp_z0(const uint16_t *ex, const uint16_t *ey, size_t n_events, const float *H,
                 const float *lut, int W, float *xy_z0)
{
    for (size_t i = 0; i < n_events; ++i) {
        const float *h = H + 9 * (i / ORC_PACKET_SIZE);
        float u, v;
        if (lut) {
            const size_t c = (size_t)ey[i] * (size_t)W + ex[i]; /* :134 */
            u = lut[2 * c];
            v = lut[2 * c + 1];
        } else {
            u = (float)ex[i];
            v = (float)ey[i];
        }
        /* 4x4 * (u,v,1,0): packet product ((c0*u + c1*v) + c2*1) + c3*0, :138 */
        const float px = ((h[0] * u + h[1] * v) + h[2] * 1.f) + 0.f * 0.f;
        const float py = ((h[3] * u + h[4] * v) + h[5] * 1.f) + 0.f * 0.f;
        const float pz = ((h[6] * u + h[7] * v) + h[8] * 1.f) + 0.f * 0.f;
        xy_z0[2 * i] = px / pz; /* :139  p /= p[2] (true division, Eigen >= 3.3) */
        xy_z0[2 * i + 1] = py / pz;
    }
}

/* ---- bilinear vote: cartesian3dgrid.h:253-273 --------------------------- */
void orc_vote(float x_f, float y_f, float *grid, int nx, int ny)
{
    if (x_f >= 0.f && y_f >= 0.f) {
        /* :257 `const int x = x_f` is UB beyond INT range; x86 cvttss2si yields
         * INT_MIN there and the unsigned compare at :258 then rejects.  With
         * x_f >= 0, x+1 < size_[0]  <=>  x_f < size_[0]-1, tested in float so
         * that huge/inf coordinates are rejected the same way. */
        if (x_f < (float)(nx - 1) && y_f < (float)(ny - 1)) {
            const int x = (int)x_f, y = (int)y_f;
            float *g = grid + x + (size_t)y * nx;
            const float fx = x_f - x, fy = y_f - y, fx1 = 1.f - fx, fy1 = 1.f - fy;
            g[0] += fx1 * fy1;
            g[1] += fx * fy1;
            g[nx] += fx1 * fy;
            g[nx + 1] += fx * fy;
        }
    }
}

/* ---- fillVoxelGrid: mapper_emvs_stereo.cpp:151-205 ---------------------- */
void orc_fill_voxel_grid(const float *xy_z0, const float *centers, size_t n_packets,
                         const float *raw_depths, int nz, const float *Kv, int nx, int ny,
                         float *dsi)
{
    enum { N = 128 }; /* :159 */
    const float z0 = raw_depths[0]; /* :163 */
    const float vfx = Kv[0], vfy = Kv[1], vcx = Kv[2], vcy = Kv[3];
    const size_t n_events = n_packets * ORC_PACKET_SIZE;

#pragma omp parallel for if (n_events >= 20000) /* :168 */
    for (int depth_plane = 0; depth_plane < nz; ++depth_plane) {
        const float *pe = xy_z0;
        float *pgrid = dsi + (size_t)depth_plane * nx * ny; /* :172 */
        for (size_t packet = 0; packet < n_packets; ++packet) {
            const float *C = centers + 3 * packet;
            /* :177-182 */
            const float zi = raw_depths[depth_plane];
            const float a = z0 * (zi - C[2]);
            const float bx = (z0 - zi) * (C[0] * vfx + C[2] * vcx);
            const float by = (z0 - zi) * (C[1] * vfy + C[2] * vcy);
            const float d = zi * (z0 - C[2]);
            for (int batch = 0; batch < ORC_PACKET_SIZE / N; ++batch, pe += 2 * N) {
                float X[N], Y[N];
                for (int i = 0; i < N; ++i) { /* :194-195, element-wise *, +, / */
                    X[i] = (pe[2 * i] * a + bx) / d;
                    Y[i] = (pe[2 * i + 1] * a + by) / d;
                }
                for (int i = 0; i < N; ++i) /* :197-201 */
                    orc_vote(X[i], Y[i], pgrid, nx, ny);
            }
        }
    }
}

/* The same loop restricted to rows [row_begin, row_begin + row_count) of every plane: every coordinate
 * and every accept test is computed on the FULL grid exactly as above, only the four += of a vote are
 * kept or dropped by their row, so `strip` ([nz][row_count][nx]) is bit-equal to those rows of the full
 * DSI (same additions in the same order).  Test infrastructure for grids whose full DSI the CPU cannot
 * afford (1024 x 1024 x 256 at 100 M events): the arg-max needs whole columns, not whole planes. */
void orc_fill_voxel_grid_rows(const float *xy_z0, const float *centers, size_t n_packets,
                              const float *raw_depths, int nz, const float *Kv, int nx, int ny,
                              int row_begin, int row_count, float *strip)
{
    enum { N = 128 };
    const float z0 = raw_depths[0];
    const float vfx = Kv[0], vfy = Kv[1], vcx = Kv[2], vcy = Kv[3];
    const int row_end = row_begin + row_count;

#pragma omp parallel for
    for (int depth_plane = 0; depth_plane < nz; ++depth_plane) {
        const float *pe = xy_z0;
        float *pgrid = strip + (size_t)depth_plane * nx * row_count;
        for (size_t packet = 0; packet < n_packets; ++packet) {
            const float *C = centers + 3 * packet;
            const float zi = raw_depths[depth_plane];
            const float a = z0 * (zi - C[2]);
            const float bx = (z0 - zi) * (C[0] * vfx + C[2] * vcx);
            const float by = (z0 - zi) * (C[1] * vfy + C[2] * vcy);
            const float d = zi * (z0 - C[2]);
            for (int batch = 0; batch < ORC_PACKET_SIZE / N; ++batch, pe += 2 * N) {
                float X[N], Y[N];
                for (int i = 0; i < N; ++i) {
                    X[i] = (pe[2 * i] * a + bx) / d;
                    Y[i] = (pe[2 * i + 1] * a + by) / d;
                }
                for (int i = 0; i < N; ++i) {
                    const float x_f = X[i], y_f = Y[i];
                    if (x_f >= 0.f && y_f >= 0.f && x_f < (float)(nx - 1) && y_f < (float)(ny - 1)) { /* orc_vote */
                        const int x = (int)x_f, y = (int)y_f;
                        if (y + 1 < row_begin || y >= row_end)
                            continue;
                        const float fx = x_f - x, fy = y_f - y, fx1 = 1.f - fx, fy1 = 1.f - fy;
                        if (y >= row_begin) {
                            float *g = pgrid + x + (size_t)(y - row_begin) * nx;
                            g[0] += fx1 * fy1;
                            g[1] += fx * fy1;
                        }
                        if (y + 1 < row_end) {
                            float *g = pgrid + x + (size_t)(y + 1 - row_begin) * nx;
                            g[0] += fx1 * fy;
                            g[1] += fx * fy;
                        }
                    }
                }
            }
        }
    }
}

/* ---- packetisation: mapper_emvs_stereo.cpp:67-99 ------------------------ */
long orc_packetize(size_t n_events, const uint8_t *pose_ok, size_t *first_event,
                   size_t *mid_event)
{
    if (n_events < ORC_PACKET_SIZE) /* :71-75 */
        return -1;
    long np = 0;
    size_t cur = 0;
    while (cur + ORC_PACKET_SIZE < n_events) { /* :88 strict < */
        const size_t mid = cur + ORC_PACKET_SIZE / 2; /* :91 */
        if (pose_ok && !pose_ok[mid]) { /* :95-99 */
            cur++;
            continue;
        }
        first_event[np] = cur;
        mid_event[np] = mid;
        np++;
        cur += ORC_PACKET_SIZE; /* :131 current_event_++ x1024 */
    }
    return np;
}

/* ---- fusion: cartesian3dgrid.h:64-192 ----------------------------------- */
int orc_fuse2(float *a, const float *g, size_t n, int op)
{
    switch (op) {
    case 1: /* minTwoGrids :111-117, std::min(a,b) = (b<a)?b:a */
        for (size_t p = 0; p < n; ++p)
            a[p] = (g[p] < a[p]) ? g[p] : a[p];
        return 0;
    case 2: { /* harmonicMeanTwoGrids :119-127, eps = 1e-1 */
        const float eps = 1e-1;
        for (size_t p = 0; p < n; ++p) {
            const float prod = a[p] * g[p];
            const float sum = a[p] + g[p];
            a[p] = 2 * prod / (sum + eps);
        }
        return 0;
    }
    case 3: /* geometricMeanTwoGrids :150-156 */
        for (size_t p = 0; p < n; ++p)
            a[p] = sqrtf(a[p] * g[p]);
        return 0;
    case 4: /* arithmeticMeanTwoGrids :158-164: 0.5 is a double literal */
        for (size_t p = 0; p < n; ++p)
            a[p] = (float)(0.5 * (double)(a[p] + g[p]));
        return 0;
    case 5: /* rmsTwoGrids :141-148: pow(float,int) and 0.5 in double, sqrt of float ms */
        for (size_t p = 0; p < n; ++p) {
            const float ms = (float)(0.5 * (pow((double)a[p], 2) + pow((double)g[p], 2)));
            a[p] = (float)sqrt((double)ms);
        }
        return 0;
    case 6: /* maxTwoGrids :184-190, std::max(a,b) = (a<b)?b:a */
        for (size_t p = 0; p < n; ++p)
            a[p] = (a[p] < g[p]) ? g[p] : a[p];
        return 0;
    default:
        return -1; /* process1.cpp:155-157 "Improper fusion method selected" */
    }
}

void orc_fuse_hm_n(float *a, const float *g, size_t n, int n_maps)
{
    /* cartesian3dgrid.h:130-139 */
    const float eps = 1e-1;
    for (size_t p = 0; p < n; ++p) {
        const float av = a[p] / (float)(n_maps - 1);
        const float prod = av * g[p];
        const float sum = av + g[p];
        a[p] = n_maps * prod / (sum + eps);
    }
}

/* ---- n-ary geometric mean helpers: log and exp spelled out in IEEE double operations
 * (+, *, /, floor, bit moves and explicit fma() calls -- IEEE 754 fusedMultiplyAdd, exact in
 * glibc with or without the instruction; nothing is fused implicitly: -ffp-contract=off).
 * The n-ary forms are NOT in the reference (it has 2-ary ops only and drops a third camera for
 * GM / AM / RMS, process1.cpp:169-191); they restate SURVEY.md 8(d) cfg 5:
 * GM = exp(mean(log v)), 0 if any v is 0. */
static double bits_to_double(uint64_t b)
{
    double d;
    memcpy(&d, &b, sizeof d);
    return d;
}

/* log v = e ln2 + log c_i + log1p(r): v = 2^e m, m in [1, 2); the top 7 mantissa bits pick
 * c_i = 1 + (2 i + 1) / 256, r = (m - c_i) / c_i, |r| <= 2^-8, log1p by its degree-5 Taylor
 * polynomial; {1 / c_i, log c_i} from a 128-entry table built once with the division-based
 * atanh series (full double accuracy).  Same operations in the same order as the engine's
 * det_logf / det_log_series (dvs_mcemvs_amd/csrc/dsi_kernels.hip), which were defined together
 * with this restatement: the n-ary GM has no reference implementation to follow. */
static double log_series(double m) /* m in [1, 2) */
{
    double e = 0.0;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = 1.0;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 0.047619047619047616;
    p = p * z + 0.05263157894736842;
    p = p * z + 0.058823529411764705;
    p = p * z + 0.06666666666666667;
    p = p * z + 0.07692307692307693;
    p = p * z + 0.09090909090909091;
    p = p * z + 0.1111111111111111;
    p = p * z + 0.14285714285714285;
    p = p * z + 0.2;
    p = p * z + 0.3333333333333333;
    p = p * z + 1.0;
    const double t1 = e * 0.6931471805599453;
    const double t2 = 2.0 * s;
    return t1 + t2 * p;
}

static double log_tab_inv[128], log_tab_log[128];
static int log_tab_ready = 0;

static void log_table_init(void) /* call before any parallel region that uses det_logf */
{
    if (log_tab_ready)
        return;
    for (int i = 0; i < 128; ++i) {
        const double c = 1.0 + (double)(2 * i + 1) * 0.00390625;
        log_tab_inv[i] = 1.0 / c;
        log_tab_log[i] = log_series(c);
    }
    log_tab_ready = 1;
}

static float det_logf(float v)
{
    if (!(v > 0.f))
        return v == 0.f ? -INFINITY : NAN;
    if (isinf(v))
        return v;
    const double x = (double)v;
    uint64_t b;
    memcpy(&b, &x, sizeof b);
    const int e = (int)((b >> 52) & 0x7ffull) - 1023;
    const unsigned i = (unsigned)(b >> 45) & 127u;
    const double m = bits_to_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    const double c = bits_to_double(0x3ff0000000000000ull | ((uint64_t)(2u * i + 1u) << 44));
    const double r = (m - c) * log_tab_inv[i];
    double p = fma(r, 0.2, -0.25);             /* fused multiply-adds, spelled out (IEEE 754 fma) */
    p = fma(p, r, 0.3333333333333333);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    p = p * r;
    return (float)fma((double)e, 0.6931471805599453, log_tab_log[i] + p);
}

static float det_expf(double y)
{
    if (y != y)
        return NAN;
    if (y > 89.0)
        return INFINITY;
    if (y < -104.0)
        return 0.f;
    const double kd = floor(y * 1.4426950408889634 + 0.5);
    double r = fma(kd, -0.6931471803691238, y);      /* ln2 split as in fdlibm (hi part has 32 bits) */
    r = fma(kd, -1.9082149292705877e-10, r);
    double p = 2.7557319223985893e-06;           /* 1/9! */
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int64_t k = (int64_t)kd;
    const double sc = bits_to_double((uint64_t)(k + 1023) << 52);
    return (float)(p * sc);
}

void orc_accumulate_begin(float *acc, size_t n, int mode)
{
    const float v = mode == 4 ? INFINITY : (mode == 5 ? -INFINITY : 0.f);
    for (size_t p = 0; p < n; ++p)
        acc[p] = v;
}

void orc_accumulate(float *acc, const float *g, size_t n, int mode)
{
    switch (mode) {
    case 0: /* addTwoGrids :64-70 */
        for (size_t p = 0; p < n; ++p)
            acc[p] += g[p];
        break;
    case 1: { /* addInverseOfTwoGrids :72-78, eps = 1e-2 */
        const float eps = 1e-2;
        for (size_t p = 0; p < n; ++p)
            acc[p] = acc[p] + 1.0f / (eps + g[p]);
        break;
    }
    case 2: /* n-ary GM: sum of log v */
        log_table_init();
        for (size_t p = 0; p < n; ++p)
            acc[p] = acc[p] + det_logf(g[p]);
        break;
    case 3: /* n-ary RMS: sum of v^2 */
        for (size_t p = 0; p < n; ++p)
            acc[p] = acc[p] + g[p] * g[p];
        break;
    case 4: /* n-ary min, std::min as in minTwoGrids :115 */
        for (size_t p = 0; p < n; ++p)
            acc[p] = (g[p] < acc[p]) ? g[p] : acc[p];
        break;
    default: /* 5: n-ary max, std::max as in maxTwoGrids :188 */
        for (size_t p = 0; p < n; ++p)
            acc[p] = (acc[p] < g[p]) ? g[p] : acc[p];
        break;
    }
}

void orc_finalize(float *acc, size_t n, int mode, int n_maps)
{
    switch (mode) {
    case 0: /* computeAMfromSum :87-93 */
        for (size_t p = 0; p < n; ++p)
            acc[p] = acc[p] / (float)n_maps;
        break;
    case 1: /* computeHMfromSumOfInv :80-86 */
        for (size_t p = 0; p < n; ++p)
            acc[p] = (float)n_maps / acc[p];
        break;
    case 2: { /* GM = exp(mean(log v)) */
        const double inv_n = 1.0 / (double)(float)n_maps;
        for (size_t p = 0; p < n; ++p)
            acc[p] = det_expf((double)acc[p] * inv_n);
        break;
    }
    case 3: /* RMS: mean square in double -> float, sqrt of the float like rmsTwoGrids :145-146 */
        for (size_t p = 0; p < n; ++p) {
            const float ms = (float)((double)acc[p] / (double)(float)n_maps);
            acc[p] = (float)sqrt((double)ms);
        }
        break;
    default: /* min / max: nothing */
        break;
    }
}

/* ---- arg-max over Z: cartesian3dgrid.cpp:115-137 ------------------------ */
void orc_collapse_max_z(const float *dsi, int nx, int ny, int nz, float *conf, uint8_t *idx)
{
    const size_t plane = (size_t)nx * ny;
    for (int v = 0; v < ny; ++v) {
        for (int u = 0; u < nx; ++u) {
            const float *col = dsi + (size_t)v * nx + u;
            /* std::max_element: keeps the first of equal maxima; uses operator< */
            float best = col[0];
            int best_k = 0;
            for (int k = 1; k < nz; ++k) {
                const float val = col[(size_t)k * plane];
                if (best < val) {
                    best = val;
                    best_k = k;
                }
            }
            conf[(size_t)v * nx + u] = best;
            idx[(size_t)v * nx + u] = (uint8_t)best_k; /* CV_8U, :120 */
        }
    }
}

void orc_indices_to_depth(const uint8_t *idx, size_t n, const float *raw_depths, float *depth)
{
    for (size_t i = 0; i < n; ++i) /* mapper_emvs_stereo.cpp:302-313 */
        depth[i] = raw_depths[idx[i]];
}

double orc_mean_square(const float *dsi, size_t n)
{
    double result = 0.; /* cartesian3dgrid.cpp:164-174 */
    for (size_t i = 0; i < n; ++i) {
        const double tmp = (double)dsi[i];
        result += tmp * tmp;
    }
    return result / (double)n;
}

/* ---- SE(3) helpers with minkindr / Eigen semantics (double) ------------- */
typedef struct {
    double t[3];
    double q[4]; /* w,x,y,z */
} pose_t;

static pose_t pose_load(const double *p)
{
    pose_t r;
    memcpy(r.t, p, 3 * sizeof(double));
    memcpy(r.q, p + 3, 4 * sizeof(double));
    return r;
}

static void pose_store(const pose_t *p, double *out)
{
    memcpy(out, p->t, 3 * sizeof(double));
    memcpy(out + 3, p->q, 4 * sizeof(double));
}

static void quat_mul(const double *a, const double *b, double *o)
{
    /* Eigen quat_product<.,.,double> */
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3];
    o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
}

static void quat_rotate(const double *q, const double *v, double *o)
{
    /* Eigen QuaternionBase::_transformVector: uv = 2*cross(qv,v); v + w*uv + cross(qv,uv) */
    const double qv[3] = {q[1], q[2], q[3]};
    double uv[3] = {qv[1] * v[2] - qv[2] * v[1], qv[2] * v[0] - qv[0] * v[2],
                    qv[0] * v[1] - qv[1] * v[0]};
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    const double c[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2],
                         qv[0] * uv[1] - qv[1] * uv[0]};
    for (int i = 0; i < 3; ++i)
        o[i] = v[i] + q[0] * uv[i] + c[i];
}

static pose_t pose_mul(const pose_t *a, const pose_t *b)
{
    /* minkindr QuatTransformation::operator*: q = qa*qb ; t = ta + qa.rotate(tb) */
    pose_t r;
    double rt[3];
    quat_mul(a->q, b->q, r.q);
    quat_rotate(a->q, b->t, rt);
    for (int i = 0; i < 3; ++i)
        r.t[i] = a->t[i] + rt[i];
    return r;
}

static pose_t pose_inv(const pose_t *a)
{
    /* minkindr inverse(): q^-1 (conjugate of a unit quaternion), t = -(q^-1).rotate(t) */
    pose_t r;
    double rt[3];
    r.q[0] = a->q[0];
    r.q[1] = -a->q[1];
    r.q[2] = -a->q[2];
    r.q[3] = -a->q[3];
    quat_rotate(r.q, a->t, rt);
    for (int i = 0; i < 3; ++i)
        r.t[i] = -rt[i];
    return r;
}

static double arcsin_x_over_x(double x)
{
    /* minkindr arcSinXOverX */
    if (fabs(x) < 1.220703125e-4 /* ~ eps^(1/4) */)
        return 1.0 + x * x * (1.0 / 6.0);
    return asin(x) / x;
}

static void quat_log(const double *q, double *o)
{
    /* minkindr RotationQuaternion::log */
    const double na = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double eta = q[0];
    double scale;
    if (fabs(eta) < na) {
        scale = (eta >= 0) ? acos(eta) / na : -acos(-eta) / na;
    } else {
        scale = (eta > 0) ? arcsin_x_over_x(na) : -arcsin_x_over_x(na);
    }
    for (int i = 0; i < 3; ++i)
        o[i] = q[1 + i] * (2.0 * scale);
}

static void quat_exp(const double *dx, double *q)
{
    /* minkindr RotationQuaternion::exp (Grassia 1998) */
    const double theta = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
    double na;
    if (theta < 1.220703125e-4)
        na = 0.5 + (theta * theta) * (1.0 / 48.0);
    else
        na = sin(theta * 0.5) / theta;
    q[0] = cos(theta * 0.5);
    q[1] = dx[0] * na;
    q[2] = dx[1] * na;
    q[3] = dx[2] * na;
}

int orc_pose_at(const double *times, const double *poses, size_t n_poses, double t, double *out)
{
    /* trajectory.hpp:98-113: it1 = upper_bound(t) */
    size_t i1 = 0;
    while (i1 < n_poses && !(t < times[i1]))
        ++i1;
    if (i1 == 0 || i1 == n_poses)
        return 0;
    const size_t i0 = i1 - 1;
    const pose_t T0 = pose_load(poses + 7 * i0), T1 = pose_load(poses + 7 * i1);
    /* :123-125 */
    const pose_t T0inv = pose_inv(&T0);
    const pose_t Trel = pose_mul(&T0inv, &T1);
    const double delta_t = (t - times[i0]) / (times[i1] - times[i0]);
    /* minkindr log(): [translation ; rotation log]; exp(): same split */
    double rl[3];
    quat_log(Trel.q, rl);
    pose_t inc;
    for (int i = 0; i < 3; ++i) {
        inc.t[i] = delta_t * Trel.t[i];
        rl[i] = delta_t * rl[i];
    }
    quat_exp(rl, inc.q);
    const pose_t T = pose_mul(&T0, &inc);
    pose_store(&T, out);
    return 1;
}

void orc_event_pose_Rt(const double *T_rv_w, const double *T_w_ev, float *Rt)
{
    /* mapper_emvs_stereo.cpp:101-105 */
    const pose_t a = pose_load(T_rv_w), b = pose_load(T_w_ev);
    const pose_t T_rv_ev = pose_mul(&a, &b);
    const pose_t T = pose_inv(&T_rv_ev);
    /* Eigen Quaternion::toRotationMatrix */
    const double w = T.q[0], x = T.q[1], y = T.q[2], z = T.q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Rt[0] = (float)(1 - (tyy + tzz));
    Rt[1] = (float)(txy - twz);
    Rt[2] = (float)(txz + twy);
    Rt[3] = (float)(txy + twz);
    Rt[4] = (float)(1 - (txx + tzz));
    Rt[5] = (float)(tyz - twx);
    Rt[6] = (float)(txz - twy);
    Rt[7] = (float)(tyz + twx);
    Rt[8] = (float)(1 - (txx + tyy));
    Rt[9] = (float)T.t[0];
    Rt[10] = (float)T.t[1];
    Rt[11] = (float)T.t[2];
}


/* ---- post-arg-max filters: mapper_emvs_stereo.cpp:390-436 ---------------- */
static void gaussian_kernel(int ksize, float *k)
{
    /* cv::getGaussianKernel with sigma <= 0: fixed tables for ksize <= 7, else
     * sigma = ((ksize-1)*0.5 - 1)*0.3 + 0.8 and a normalised exp() in double */
    static const float k1[] = {1.f};
    static const float k3[] = {0.25f, 0.5f, 0.25f};
    static const float k5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float k7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    const float *fixed = ksize == 1 ? k1 : ksize == 3 ? k3 : ksize == 5 ? k5 : ksize == 7 ? k7 : 0;
    if (fixed) {
        memcpy(k, fixed, ksize * sizeof(float));
        return;
    }
    const double sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0, t[64];
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < ksize; ++i)
        k[i] = (float)(t[i] * (1. / sum));
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline uint8_t saturate_u8(float v)
{
    /* cv::saturate_cast<uchar>(float): cvRound (round half to even), clamped */
    const int i = (int)lrintf(v);
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

void orc_depth_map_filters(float *conf, const uint8_t *idx, int nx, int ny, int ksize, double C,
                           int median_size, double max_confidence, const float *raw_depths,
                           uint8_t *conf8, uint8_t *mask, uint8_t *idx_filtered, float *depth)
{
    const size_t n = (size_t)nx * ny;
    /* :393 */
    conf[0] = (float)max_confidence;
    /* :394 cv::normalize(src, dst, 0, 255, NORM_MINMAX): scale/shift in double, applied in float */
    float smin = conf[0], smax = conf[0];
    for (size_t i = 1; i < n; ++i) {
        if (conf[i] < smin) smin = conf[i];
        if (conf[i] > smax) smax = conf[i];
    }
    const double range = (double)smax - (double)smin;
    const double scale = 255.0 * (range > 2.220446049250313e-16 ? 1. / range : 0.);
    const double shift = 0.0 - (double)smin * scale;
    const float a = (float)scale, b = (float)shift;
    for (size_t i = 0; i < n; ++i) {
        float v = conf[i] * a + b;
        if (i == 0) v = 0.f; /* :396 */
        conf8[i] = saturate_u8(v); /* :397 */
    }
    /* :403-409 adaptiveThreshold: Gaussian mean on the float image, BORDER_REPLICATE, rounded
     * back to uchar; dst = 1 where src - mean > -cvCeil(-C) */
    float kern[64];
    if (ksize > 63) ksize = 63;
    gaussian_kernel(ksize, kern);
    const int r = ksize / 2;
    float *tmp = (float *)malloc(n * sizeof(float));
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) { /* row pass */
            float acc = 0.f;
            for (int t = -r; t <= r; ++t)
                acc += kern[t + r] * (float)conf8[(size_t)y * nx + clampi(x + t, 0, nx - 1)];
            tmp[(size_t)y * nx + x] = acc;
        }
    const int idelta = (int)ceil(-C);
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) { /* column pass */
            float acc = 0.f;
            for (int t = -r; t <= r; ++t)
                acc += kern[t + r] * tmp[(size_t)clampi(y + t, 0, ny - 1) * nx + x];
            const int mean = saturate_u8(acc);
            mask[(size_t)y * nx + x] = ((int)conf8[(size_t)y * nx + x] - mean > -idelta) ? 1 : 0;
        }
    free(tmp);
    /* :420-423 huangMedianFilter: median of the masked, in-image window values; the histogram
     * walk of median_filtering.cpp:33-158 visits every pixel with exactly its window */
    const int p = median_size / 2;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            int h[256] = {0}, num = 0;
            for (int yy = y - p; yy <= y + p; ++yy)
                for (int xx = x - p; xx <= x + p; ++xx)
                    if (yy >= 0 && xx >= 0 && yy < ny && xx < nx && mask[(size_t)yy * nx + xx] > 0) {
                        h[idx[(size_t)yy * nx + xx]]++;
                        num++;
                    }
            /* compute_median_histogram, median_filtering.cpp:7-18 */
            const int middle = (num + 1) / 2;
            int m = 0, v = 0;
            for (v = 0; v < 256; ++v) {
                m += h[v];
                if (m >= middle) break;
            }
            idx_filtered[(size_t)y * nx + x] = (uint8_t)v;
        }
    /* :426-427 removeMaskBoundary(mask, max(ksize/2, 1)) (:316-329) */
    const int border = ksize / 2 > 1 ? ksize / 2 : 1;
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x)
            if (x <= border || x >= nx - border || y <= border || y >= ny - border)
                mask[(size_t)y * nx + x] = 0;
    /* :435 */
    for (size_t i = 0; i < n; ++i)
        depth[i] = raw_depths[idx_filtered[i]];
}
