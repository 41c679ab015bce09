// dsi_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DSI engine.
//
// Replaces, on the GPU, the reference CPU code
//   MapperEMVS::evaluateDSI stage A   mapper_emvs_stereo.cpp:101-142
//   MapperEMVS::fillVoxelGrid         mapper_emvs_stereo.cpp:151-205
//   Grid3D::accumulateGridValueAt     cartesian3dgrid.h:253-273
//   Grid3D fusion ops                 cartesian3dgrid.h:64-192
//   Grid3D::collapseMaxZSlice         cartesian3dgrid.cpp:115-137
//   Grid3D::computeMeanSquare         cartesian3dgrid.cpp:164-174
//
// Numerics: the reference is an SSE2 build without FMA.  Every coordinate is
// computed with the same operations in the same order (separate multiply, add,
// IEEE divide); only the order in which votes are summed into a voxel differs.
// This file must be compiled with -ffp-contract=off (the pragma below restates
// it); explicit fmaf() calls are the only fused operations.
#include "dsi_kernels.h"
#include "dsi_vote_asm.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <mutex>
#include <vector>

#pragma clang fp contract(off)

namespace dsi {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ bool finitef(float v) { return __builtin_isfinite(v); }

// Eigen 3.3 fixed-size length-3 dot: x0 + (x1 + x2) (redux_novec_unroller).
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    const float x0 = a0 * b0, x1 = a1 * b1, x2 = a2 * b2;
    return x0 + (x1 + x2);
}

__device__ __forceinline__ float cof3(const float* m, int i, int j)
{
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

// Eigen compute_inverse<Matrix3f> (cofactors; first column drives the determinant).
__device__ void inverse3x3(const float* m, float* out)
{
    const float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
    const float det = dot3(c0, m[0], c1, m[3], c2, m[6]);
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

__device__ void mul3x3(const float* a, const float* b, float* c)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[3 * i + j] = dot3(a[3 * i], b[j], a[3 * i + 1], b[3 + j], a[3 * i + 2], b[6 + j]);
}

// n / d, correctly rounded, given r = RN(1/d): two residual corrections, the same
// recurrence the gfx950 IEEE-divide expansion runs after v_rcp_f32, without its
// v_div_scale / v_div_fixup range handling (callers guarantee 2^-40 <= |d| <= 2^40;
// a non-finite or overflowing n gives NaN/inf, which the vote rejects either way).
__device__ __forceinline__ float div_rc(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

// mapper_emvs_stereo.cpp:177-182
__device__ __forceinline__ void plane_coefficients(float cx_, float cy_, float cz_, float zi,
                                                   const Geom& g, float& a, float& bx, float& by,
                                                   float& d)
{
    a = g.z0 * (zi - cz_);
    bx = (g.z0 - zi) * (cx_ * g.vfx + cz_ * g.vcx);
    by = (g.z0 - zi) * (cy_ * g.vfy + cz_ * g.vcy);
    d = zi * (g.z0 - cz_);
}

// ---------------------------------------------------------------- stage A --
// mapper_emvs_stereo.cpp:101-126: one thread per packet.
// camera centre (3 floats) and H_z0 (9 floats, row-major) of one packet from its pose Rt (12 floats)
__device__ __forceinline__ void packet_geometry_of(const float* __restrict__ Rtk, const Geom& g,
                                                   float* __restrict__ center, float* __restrict__ Hout)
{
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rtk[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Rtk[9 + i];

    // :108  C = -R^T t
#pragma unroll
    for (int i = 0; i < 3; ++i)
        center[i] = dot3(-R[i], t[0], -R[3 + i], t[1], -R[6 + i], t[2]);

    // :114-116
    float Hinv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hinv[i] = R[i] * g.z0;
    Hinv[2] += t[0];
    Hinv[5] += t[1];
    Hinv[8] += t[2];

    const float Km[9] = {g.kfx, 0.f, g.kcx, 0.f, g.kfy, g.kcy, 0.f, 0.f, 1.f};
    const float Kvm[9] = {g.vfx, 0.f, g.vcx, 0.f, g.vfy, g.vcy, 0.f, 0.f, 1.f};
    float Kvinv[9], M1[9], M[9], Hk[9];
    inverse3x3(Kvm, Kvinv);  // geometry_utils.hpp:47
    mul3x3(Km, Hinv, M1);    // :119, left to right
    mul3x3(M1, Kvinv, M);
    inverse3x3(M, Hk);       // :120
#pragma unroll
    for (int i = 0; i < 9; ++i) Hout[i] = Hk[i];
}

__global__ void k_packet_geometry(const float* __restrict__ Rt, int np, Geom g,
                                  float* __restrict__ centers, float* __restrict__ H)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    packet_geometry_of(Rt + 12 * (size_t)k, g, centers + 3 * (size_t)k, H + 9 * (size_t)k);
}

// mapper_emvs_stereo.cpp:129-142 for one event: LUT, 4x4 packet product ((c0*u + c1*v) + c2*1) + c3*0,
// p /= p[2]
// the two halves of it, for callers that keep several events in flight: the rectified pixel (:134; NaN for a pixel
// outside the sensor, which has no LUT entry -- the reference would read past its matrix --: the event then gets a
// non-finite location, which no plane accepts) ...
__device__ __forceinline__ float2 rectified_pixel(unsigned x, unsigned y, const float2* __restrict__ lut, int sensor_w,
                                                  int sensor_h)
{
    if (!lut) return make_float2((float)x, (float)y);
    // (branch-free, so that a thread's look-ups are issued back to back: entry 0 is read in place of a missing one)
    const bool inside = x < (unsigned)sensor_w && y < (unsigned)sensor_h;
    const float2 p = lut[inside ? (size_t)y * sensor_w + x : (size_t)0];
    const float nan = __builtin_nanf("");
    return make_float2(inside ? p.x : nan, inside ? p.y : nan);
}

// ... and the packet's homography applied to it (:135-142)
__device__ __forceinline__ float2 warp_pixel_z0(float2 uv, const float* __restrict__ h)
{
    const float u = uv.x, v = uv.y;
    const float px = ((h[0] * u + h[1] * v) + h[2] * 1.f) + 0.f;
    const float py = ((h[3] * u + h[4] * v) + h[5] * 1.f) + 0.f;
    const float pz = ((h[6] * u + h[7] * v) + h[8] * 1.f) + 0.f;
    return make_float2(px / pz, py / pz);
}

__device__ __forceinline__ float2 warp_event_z0(unsigned x, unsigned y, const float* __restrict__ h,
                                                const float2* __restrict__ lut, int sensor_w, int sensor_h)
{
    if (lut && (x >= (unsigned)sensor_w || y >= (unsigned)sensor_h)) return make_float2(__builtin_nanf(""), __builtin_nanf(""));
    return warp_pixel_z0(rectified_pixel(x, y, lut, sensor_w, sensor_h), h);
}

// mapper_emvs_stereo.cpp:129-142: one thread per event slot of a packet.
__global__ void k_warp_z0(const uint16_t* __restrict__ ex, const uint16_t* __restrict__ ey,
                          const uint32_t* __restrict__ packet_first, int np,
                          const float* __restrict__ H, const float2* __restrict__ lut,
                          int sensor_w, int sensor_h, float2* __restrict__ xy)
{
    const int k = blockIdx.x;  // packet
    const size_t first = packet_first ? (size_t)packet_first[k] : (size_t)k * kPacket;
    const float* h = H + 9 * (size_t)k;
    const float h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6],
                h7 = h[7], h8 = h[8];
    const float hh[9] = {h0, h1, h2, h3, h4, h5, h6, h7, h8};
    for (int j = threadIdx.x; j < kPacket; j += blockDim.x)
        xy[(size_t)k * kPacket + j] = warp_event_z0(ex[first + j], ey[first + j], hh, lut, sensor_w, sensor_h);
}

// ------------------------------------------------- stage B: global atomics --
// cartesian3dgrid.h:253-273 with the int-conversion guard expressed in float
// (x+1 < Nx  <=>  x_f < Nx-1 for x_f >= 0), so that inf/huge/NaN are rejected
// exactly like the x86 build rejects them.
__device__ __forceinline__ void vote_global(float X, float Y, float* __restrict__ plane, int nx,
                                            float xmax, float ymax)
{
    if (X >= 0.f && Y >= 0.f && X < xmax && Y < ymax) {
        const int xi = (int)X, yi = (int)Y;
        const float fx = X - (float)xi, fy = Y - (float)yi, fx1 = 1.f - fx, fy1 = 1.f - fy;
        float* gptr = plane + (size_t)yi * nx + xi;
        unsafeAtomicAdd(gptr, fx1 * fy1);
        unsafeAtomicAdd(gptr + 1, fx * fy1);
        unsafeAtomicAdd(gptr + nx, fx1 * fy);
        unsafeAtomicAdd(gptr + nx + 1, fx * fy);
    }
}

constexpr int kVgPlanes = 8;  // planes per block of the global-atomic kernel

// block = (packet, group of planes); thread t owns events t, t+256, ... of the packet
__global__ __launch_bounds__(256) void k_vote_global(const float2* __restrict__ xy,
                                                     const float* __restrict__ centers,
                                                     const float* __restrict__ planes, Geom g,
                                                     float* __restrict__ dsi)
{
    const int k = blockIdx.x;
    const int zbeg = blockIdx.y * kVgPlanes;
    const int zend = min(g.nz, zbeg + kVgPlanes);
    const float cx_ = centers[3 * k], cy_ = centers[3 * k + 1], cz_ = centers[3 * k + 2];
    float2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = xy[(size_t)k * kPacket + threadIdx.x + 256 * i];
    const float xmax = (float)(g.nx - 1), ymax = (float)(g.ny - 1);
    const size_t plane_sz = (size_t)g.nx * g.ny;
    for (int z = zbeg; z < zend; ++z) {
        float a, bx, by, d;
        plane_coefficients(cx_, cy_, cz_, planes[z], g, a, bx, by, d);
        float* plane = dsi + (size_t)z * plane_sz;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float X = (e[i].x * a + bx) / d;  // :194
            const float Y = (e[i].y * a + by) / d;  // :195
            vote_global(X, Y, plane, g.nx, xmax, ymax);
        }
    }
}

// ------------------------------------------------ stage B: LDS row bands ---
// (1) per packet: drop events that no plane can accept (non-finite z0 location) and group
//     the rest by the row floor(y0) they fall in at z0 (counting sort in LDS; the order inside
//     a row does not matter).  For one (packet, plane) the map y0 -> Y is monotone (each of
//     *a, +by, /d rounds monotonically), so the events landing in a band of output rows come
//     from a contiguous range of z0 rows = one contiguous run of the grouped packet.
//     Rows are binned over [-pad, ny+pad) (events warped to z0 spill well outside the grid
//     when the camera is far from the reference view, and other planes pull them back in);
//     bin 0 holds everything below, the last bin everything above:
//     bin(y0) = clamp(floor(y0), -pad-1, ny+pad) + pad + 1, nb = ny + 2*pad + 2 bins,
//     rowstart[p][b] = number of kept events in bins < b, b = 0..nb.
__device__ __forceinline__ int row_bin(float y0, int ny, int pad)
{
    const float f = fminf(fmaxf(__builtin_floorf(y0), (float)(-pad - 1)), (float)(ny + pad));
    return (int)f + pad + 1;  // finite y0 only
}

// Events of one packet with bit-identical z0 locations (same raw pixel fired again within the
// packet) are merged into one record with a multiplicity: an open-addressing hash set in LDS
// keyed by the 64 bits of (x0, y0); the first event to arrive in a slot represents the rest.
constexpr int kHashSlots = 2048;  // >= 2 * kPacket
constexpr unsigned long long kHashEmpty = ~0ull;  // a NaN pair: never a finite location

// Stage A of a packet, fused in when the events come raw (xy == nullptr): the block computes its
// packet's camera centre and H_z0 (one thread; written to `centers` for k_plane_coef) and warps its
// 1024 events to z0 in registers -- the z0 locations never go through HBM (saves 8 B written +
// 8 B read per event and two launches per evaluateDSI).
struct RawEvents {
    const float* Rt;             // [np][12]
    const uint16_t *ex, *ey;
    const uint32_t* packet_first;  // or nullptr: packet k starts at k * 1024
    const float2* lut;             // or nullptr: identity
    int sensor_w, sensor_h;
    Geom g;
    float* centers;              // out [np][3]
    int unit_multiplicity;       // test hook: every record gets multiplicity 1 (the DSI then counts accepted RECORDS)
};

// one camera's share of a multi-camera preparation launch (k_sort_packets_multi, k_plane_coef_multi)
struct PrepCamera {
    RawEvents raw;
    int np;
    EvRec* sxy;
    uint32_t* nvalid;
    uint16_t* rowstart;
    const float* planes;
    PlaneCoef* coef;
    uint32_t* cuts;
    uint32_t* pair_work;  // [bands][nz]: records the voting kernel will look at per (band, plane), or nullptr
};
struct PrepCameras {
    PrepCamera cam[kFusedMaxCameras];
    int n;
};

// RAW (the events come as sensor pixels, stage A fused in): two events of a packet have the same z0
// location iff they have the same pixel, so the hash set is keyed by the 32-bit pixel and one 64-bit
// word per slot holds key and count -- 16 KB instead of 24 KB of LDS, which is what decides how many
// packets a CU sorts at once (time at 10 M events: 35 us + 260 us / blocks per CU, measured by
// padding the LDS: 5 blocks 86 us, 4: 99, 3: 122, 2: 172; now 8).
template <bool RAW>
__device__ __forceinline__ void sort_packets_body(const int k, const float2* __restrict__ xy, const RawEvents& raw,
                                                  int np, int ny,
                                                  int nz, int pad, EvRec* __restrict__ sxy,
                                                  uint32_t* __restrict__ nvalid,
                                                  uint16_t* __restrict__ rowstart, uint32_t* __restrict__ pair_work,
                                                  int n_pairs)
{
    extern __shared__ uint32_t hist[];  // nb + 1 counters, then scanned in place
    __shared__ float s_H[9];
    __shared__ __attribute__((aligned(16))) unsigned long long hkey[kHashSlots];  // RAW: pixel | count << 32
    __shared__ uint32_t hcnt[RAW ? 1 : kHashSlots];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t big;  // some |x0| or |y0| above 2^40 (never a real pixel)
    const int nb = ny + 2 * pad + 2;
    // nvalid[np + z]: "plane z has a coefficient set that needs the IEEE divide", set by
    // k_plane_coef (next kernel on the stream), read by the packed voting kernel
    if (k == 0) {
        // (+ 8 work counters of the persistent voting kernel behind the per-plane flags, + lane mapping 8's overflow word)
        for (int i = threadIdx.x; i < nz + 9; i += 256) nvalid[np + i] = 0;
        // (+ the records per (band, plane) pair that k_plane_coef counts for the fused kernel's partition)
        if (pair_work)
            for (int i = threadIdx.x; i < n_pairs; i += 256) pair_work[i] = 0;
        if (threadIdx.x == 0) {  // the multiplicity-0 record behind the last packet
            const EvRec none = {0.f, 0.f, 0u};
            sxy[(size_t)np * kPacket] = none;
        }
    }
    if (threadIdx.x == 0) big = 0;
    for (int i = threadIdx.x; i <= nb; i += 256) hist[i] = 0;
    for (int i = threadIdx.x; i < kHashSlots; i += 256) {
        hkey[i] = kHashEmpty;
        if (!RAW) hcnt[i] = 0;
    }
    if (RAW && threadIdx.x == 0) {
        float c3[3], h9[9];
        packet_geometry_of(raw.Rt + 12 * (size_t)k, raw.g, c3, h9);
#pragma unroll
        for (int i = 0; i < 3; ++i) raw.centers[3 * (size_t)k + i] = c3[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) s_H[i] = h9[i];
    }
    // the four events of a thread travel TOGETHER: all coordinate loads, then all look-ups of the rectification table,
    // before the first hash insert (one after the other -- load, look-up, insert, next load -- a block spent four
    // round trips to HBM and four to L2 in a row: 63 us per 10 M events; the loads are issued before the barrier)
    float2 ev[4];
    uint32_t pixel4[4] = {0u, 0u, 0u, 0u};
    if (RAW) {
        const size_t first_ev = raw.packet_first ? (size_t)raw.packet_first[k] : (size_t)k * kPacket;
        unsigned px[4], py[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const size_t e = first_ev + threadIdx.x + 256 * h;
            px[h] = raw.ex[e];
            py[h] = raw.ey[e];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) pixel4[h] = px[h] | (py[h] << 16);
        if (raw.lut) {  // rectified_pixel() x 4 with the four look-ups issued back to back
            bool inside[4];
            float2 p[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                inside[h] = px[h] < (unsigned)raw.sensor_w && py[h] < (unsigned)raw.sensor_h;
                p[h] = raw.lut[inside[h] ? (size_t)py[h] * raw.sensor_w + px[h] : (size_t)0];
            }
            const float nan = __builtin_nanf("");
#pragma unroll
            for (int h = 0; h < 4; ++h) ev[h] = make_float2(inside[h] ? p[h].x : nan, inside[h] ? p[h].y : nan);
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h) ev[h] = make_float2((float)px[h], (float)py[h]);
        }
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) ev[h] = xy[(size_t)k * kPacket + threadIdx.x + 256 * h];
    }
    __syncthreads();
    if (RAW) {
#pragma unroll
        for (int h = 0; h < 4; ++h) ev[h] = warp_pixel_z0(ev[h], s_H);  // (NaN in, NaN out)
    }
    int bin[4], slot[4];
    uint32_t rank[4], mult[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const uint32_t pixel = pixel4[h];
        bin[h] = -1;
        slot[h] = 0;
        rank[h] = 0;
        mult[h] = 0;
        if (finitef(ev[h].x) && finitef(ev[h].y)) {
            bool first;
            if (RAW) {
                // (a claimed word has a count < 2^11 in its high half, the empty word all ones: no pixel
                //  value can be mistaken for an empty slot)
                uint32_t hs = (pixel * 0x9E3779B1u) >> 21;  // 11 bits
                for (;;) {
                    const unsigned long long old = atomicCAS(&hkey[hs], kHashEmpty, (unsigned long long)pixel);
                    if (old == kHashEmpty || (uint32_t)old == pixel) break;
                    hs = (hs + 1) & (kHashSlots - 1);
                }
                slot[h] = (int)hs;
                first = (atomicAdd(&hkey[hs], 1ull << 32) >> 32) == 0ull;
            } else {
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint(ev[h].y) << 32) | __float_as_uint(ev[h].x);
                uint32_t hs = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 53);  // 11 bits
                for (;;) {
                    const unsigned long long old = atomicCAS(&hkey[hs], kHashEmpty, key);
                    if (old == kHashEmpty || old == key) break;
                    hs = (hs + 1) & (kHashSlots - 1);
                }
                slot[h] = (int)hs;
                first = atomicAdd(&hcnt[hs], 1u) == 0u;
            }
            if (first) bin[h] = row_bin(ev[h].y, ny, pad);  // representative
            if (!(fabsf(ev[h].x) <= 0x1p40f && fabsf(ev[h].y) <= 0x1p40f)) big = 1u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (bin[h] >= 0) {
            rank[h] = atomicAdd(&hist[bin[h]], 1u);
            mult[h] = RAW ? (uint32_t)(hkey[slot[h]] >> 32) : hcnt[slot[h]];  // (the records are staged over hkey later)
            if (RAW && raw.unit_multiplicity) mult[h] = 1u;
        }
    __syncthreads();
    // exclusive scan of hist[0..nb): each thread owns a contiguous slice
    const int per = (nb + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    uint32_t local = 0;
    for (int i = b0; i < b1; ++i) local += hist[i];
    uint32_t incl = local;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - local;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
    const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    for (int i = b0; i < b1; ++i) {
        const uint32_t c = hist[i];
        hist[i] = base;
        base += c;
    }
    if (threadIdx.x == 0) hist[nb] = total;
    __syncthreads();
    uint16_t* rs = rowstart + (size_t)k * (nb + 1);
    for (int i = threadIdx.x; i <= nb; i += 256) rs[i] = (uint16_t)hist[i];
    // the grouped records go through LDS (the hash keys are dead by now: 16 KB >= 1024 x 12 B) and
    // leave as one linear 16-byte-per-lane copy; scattered 12-byte stores cost the kernel ~15 %
    static_assert(sizeof(unsigned long long) * kHashSlots >= sizeof(EvRec) * kPacket, "staging area");
    EvRec* stage = reinterpret_cast<EvRec*>(hkey);
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (bin[h] >= 0) {
            EvRec r;
            r.x = ev[h].x;
            r.y = ev[h].y;
            r.m = mult[h];
            stage[hist[bin[h]] + rank[h]] = r;
        }
    __syncthreads();
    {
        const int n16 = (int)((total * (uint32_t)sizeof(EvRec) + 15u) / 16u);  // <= 768
        const uint4* src = reinterpret_cast<const uint4*>(stage);
        uint4* dst = reinterpret_cast<uint4*>(sxy + (size_t)k * kPacket);  // k * 12 KB: 16-byte aligned
        for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    }
    if (threadIdx.x == 0) nvalid[k] = total | (big << 31);  // total <= 1024
}

template <bool RAW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sort_packets(const float2* __restrict__ xy, RawEvents raw,
                                                      int np, int ny,
                                                      int nz, int pad, EvRec* __restrict__ sxy,
                                                      uint32_t* __restrict__ nvalid,
                                                      uint16_t* __restrict__ rowstart)
{
    sort_packets_body<RAW>((int)blockIdx.x, xy, raw, np, ny, nz, pad, sxy, nvalid, rowstart, nullptr, 0);
}

// the packets of up to two cameras in ONE launch (the fused kernel's preparation: a 50 ms window has ~490
// packets per camera, and two launches of ~490 blocks each cost two launch latencies for nothing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sort_packets_multi(PrepCameras cams, int ny, int nz, int pad,
                                                                                                       int n_pairs)
{
    int k = (int)blockIdx.x, c = 0;
    while (c + 1 < cams.n && k >= cams.cam[c].np) k -= cams.cam[c++].np;
    const PrepCamera& pc = cams.cam[c];
    sort_packets_body<true>(k, nullptr, pc.raw, pc.np, ny, nz, pad, pc.sxy, pc.nvalid, pc.rowstart, pc.pair_work, n_pairs);
}

// lane mappings 2 and 4 sort groups of packets together (k_sort_groups)
__device__ __host__ __forceinline__ bool grouped(int packed) { return packed == 2 || packed == 4; }

// (2) per (packet, plane): coefficients + for every band the run [lo,hi) of the grouped
//     packet whose Y can fall into the band.  The band's Y interval is mapped back to z0 rows
//     (in double, widened by one row on each side), and the run is read off the packet's
//     rowstart table.  The run is a superset; the voting kernel re-tests every event exactly.
//     A block covers 16 packets x (blockDim / 16) planes and, when they fit (STAGED), first copies
//     the 16 packets' rowstart tables into LDS with coalesced loads: every thread looks up two
//     entries per band at data-dependent positions, and out of global memory those look-ups, not
//     the table writes, bounded the kernel (1024 x 1024 x 256, 10 M events: 440 us, 16 x 16 tiles
//     best; wider tiles for longer write segments were slower).
constexpr int kCoefTilePackets = 16;

// The z0 row bins a band's rows come from: band rows [r0 - halo, min(r1, ny - 1)] -> y0 = Y * d_a - by_a (fp32; its rounding,
// <= 4e-7 of |y0| + |by/a|, disappears in the widening) -> entries of the packet's row table: the run is
// [rs[*bin_lo], rs[*bin_hi]).  false: a NaN bound -- the run is the whole packet.  ONE function for k_plane_coef's cut
// table and for the vector fill's inline cuts (BandPlan::cuts_inline), so that both take the same runs.
__device__ __forceinline__ bool band_row_bins(float d_a, float by_a, float L, float U, int ny, int pad, int* bin_lo, int* bin_hi)
{
    const float ya = L * d_a - by_a, yb = U * d_a - by_a;
    float ymin = fminf(ya, yb), ymax = fmaxf(ya, yb);
    if (!(ya == ya && yb == yb)) return false;
    // the fp32 forward map (mul, add, divide) is within a few ulps of the real
    // one: in y0 units that is ~2^-22 * (|y0| + |by/a|); widen by 80x that
    const float m = 2e-5f * (fmaxf(fabsf(ymin), fabsf(ymax)) + fabsf(by_a));
    ymin -= m;
    ymax += m;
    const float fa = fminf(fmaxf(__builtin_floorf(ymin), (float)(-pad - 1)), (float)(ny + pad));
    const float fb = fminf(fmaxf(__builtin_floorf(ymax), (float)(-pad - 1)), (float)(ny + pad));
    *bin_lo = (int)fa + pad + 1;  // events in bins below bin(fa)
    *bin_hi = (int)fb + pad + 2;  // events in bins up to and including bin(fb)
    return true;
}
// (the band's accepted Y interval: accepted iff 0 <= Y < ny-1; the band needs floor(Y) in [r0, r1-1]; bp.halo: the fused
//  kernel's bands also take the events of the row above their first)
__device__ __forceinline__ void band_y_interval(int j, const Geom& g, const BandPlan& bp, float* L, float* U)
{
    const int r0 = j * bp.band_rows;
    const int r1 = min(g.ny, r0 + bp.band_rows);
    *L = (float)(r0 - bp.halo) - 0.01f;
    *U = (float)min(r1, g.ny - 1) + 0.01f;
}

template <bool STAGED>
__device__ __forceinline__ void plane_coef_body(const unsigned bid, const float* __restrict__ centers,
                                                const float* __restrict__ planes,
                                                const uint16_t* __restrict__ rowstart,
                                                uint32_t* __restrict__ nvalid, int np,
                                                const Geom& g, const BandPlan& bp,
                                                PlaneCoef* __restrict__ coef,
                                                uint32_t* __restrict__ cuts, uint32_t* __restrict__ pair_work)
{
    // 16 consecutive packets per plane make the plane-major tables (coef[z][p], cuts[band][z][p])
    // 64-byte coalesced writes
    extern __shared__ uint16_t s_rowstart[];  // STAGED: [16][nb + 1]
    // Block b runs on XCD b % 8 (the dispatch rule the voting kernel relies on too).  Eight tiles that
    // are neighbours along the packet axis -- the eight 64-byte pieces of a 512-byte stretch of every
    // table row -- go to eight consecutive blocks of ONE XCD, so that they meet in that XCD's L2 and
    // leave it as whole lines (dealt round-robin over the XCDs, every line was written in pieces:
    // 1024 x 1024 x 256, 10 M events: 422 us; groups of four 365 us; groups of eight 337 us).
    const int tiles_p = (np + kCoefTilePackets - 1) / kCoefTilePackets;
    const unsigned tile = (bid >> 6) * 64u + (bid & 7u) * 8u + ((bid >> 3) & 7u);
    const int planes_per_block = (int)(blockDim.x >> 4);
    if (tile >= (unsigned)tiles_p * (unsigned)((g.nz + planes_per_block - 1) / planes_per_block)) return;
    const int k0 = (int)(tile % (unsigned)tiles_p) * kCoefTilePackets;
    const int k = k0 + (int)(threadIdx.x & 15);
    const int z = (int)(tile / (unsigned)tiles_p) * planes_per_block + (int)(threadIdx.x >> 4);
    const int pad = bp.row_pad;
    const int nb = g.ny + 2 * pad + 2;
    if (STAGED && !(bp.cuts_inline && !pair_work)) {
        // the tile's tables are one contiguous piece of rowstart, 4-byte aligned (k0 is even); the
        // last 32-bit word may reach one element past the piece (the buffer has that slack)
        const int count = min(kCoefTilePackets, np - k0) * (nb + 1);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(rowstart + (size_t)k0 * (nb + 1));
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_rowstart);
        for (int i = threadIdx.x; i < (count + 1) / 2; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    // (with pair_work the threads beyond the tables stay: the row reductions below need whole rows of lanes;
    //  they act as dead packets and store nothing)
    const bool valid = k < np && z < g.nz;
    if (!valid && !pair_work) return;
    const int kq = valid ? k : 0, zq = valid ? z : 0;
    const size_t tid = (size_t)zq * np + kq;
    PlaneCoef c;
    plane_coefficients(centers[3 * kq], centers[3 * kq + 1], centers[3 * kq + 2], planes[zq], g, c.a,
                       c.bx, c.by, c.d);
    c.r = 1.f / c.d;
    const uint32_t nvraw = !valid ? 0u : (grouped(bp.packed) ? 1u : nvalid[kq]);
    const int nv = (int)(nvraw & 0x7fffffffu);
    const bool big_events = (nvraw >> 31) != 0;
    // Which (packet, plane) pairs can vote at all?  NaN anywhere, d == 0, or an
    // infinite a / bx / by make X or Y non-finite for every event.
    const bool dead = !(c.a == c.a) || !(c.bx == c.bx) || !(c.by == c.by) || !(c.d == c.d) ||
                      c.d == 0.f || !finitef(c.a) || !finitef(c.bx) || !finitef(c.by) || nv == 0;
    const float ad = fabsf(c.d);
    // "slow": the residual-corrected division is only proven for 2^-40 <= |d| <= 2^40; with
    // |x0|, |y0|, |a|, |bx|, |by| <= 2^40 as well, no intermediate of the fast path can overflow,
    // so X and Y are finite there and the voting kernel needs no NaN / overflow guards
    const bool slow = !dead && (!(ad >= 0x1p-40f && ad <= 0x1p40f) /* incl. d = +-inf */ ||
                                big_events || fabsf(c.a) > 0x1p40f || fabsf(c.bx) > 0x1p40f ||
                                fabsf(c.by) > 0x1p40f);
    // y0 = (Y*d - by)/a inverts the transfer; unusable when the map is (nearly) constant or
    // the inversion is ill-conditioned -- then the whole packet is the (superset) run
    const double a = (double)c.a, d = (double)c.d, by = (double)c.by;
    const bool invertible = !dead && !slow && fabs(a) > 1e-3 * fabs(d);
    const double inv_a = 1.0 / a;
    // y0 = Y * d_a - by_a, per band in fp32: its rounding (<= 4e-7 of |y0| + |by/a|) disappears in the
    // widening below, and 2 x bands x planes x packets double-precision chains were what the kernel
    // spent its time on
    const float by_a = (float)(by * inv_a), d_a = (float)(d * inv_a);
    // The inline cuts (BandPlan::cuts_inline) widen every band's interval by ONE bound per (packet, plane), M >= the
    // 2e-5 * (|y0| + |by/a|) of band_row_bins for every band (|y0| <= (ny + 1) |d/a| + |by/a|), in 1/256 rows in the high half of
    // flags; they only trust d_a / by_a where every band's y0 stays far inside the int32 range of v_cvt_flr_i32_f32.
    const float margin = 2e-5f * ((float)(g.ny + 1) * fabsf(d_a) + 2.f * fabsf(by_a));
    const bool inline_ok = invertible && fabsf(d_a) <= 1e4f && fabsf(by_a) <= 1e8f && margin <= 200.f;
    const uint32_t margin_q = inline_ok ? (uint32_t)__builtin_ceilf(margin * 256.f) + 1u : 0u;
    c.flags = (dead ? kCoefSkip : (slow ? kCoefSlow : 0u)) | (inline_ok ? kCoefInvertible : 0u) | (margin_q << 16);
    c.d_a = d_a;
    c.by_a = by_a;
    if (valid) coef[tid] = c;
    if (slow) atomicOr(&nvalid[np + zq], 1u);  // (slow implies valid: nv > 0)
    // no cut table: the voting kernel's passes derive their runs themselves (BandPlan::cuts_inline)
    if (bp.cuts_inline && !pair_work) return;

    const uint16_t* rs = STAGED ? s_rowstart + (size_t)(threadIdx.x & 15) * (nb + 1)
                                : rowstart + (grouped(bp.packed) ? 0 : (size_t)kq * (nb + 1));
    for (int j = 0; j < bp.bands; ++j) {
        uint32_t lo = 0, hi = 0;
        if (!dead) {
            hi = (uint32_t)nv;
            if (invertible) {
                float L, U;
                band_y_interval(j, g, bp, &L, &U);
                int bin_lo, bin_hi;
                if (band_row_bins(d_a, by_a, L, U, g.ny, pad, &bin_lo, &bin_hi)) {  // not NaN
                    if (grouped(bp.packed)) {  // grouped mapping: row bins, resolved per group later
                        lo = (uint32_t)bin_lo;
                        hi = (uint32_t)bin_hi;
                    } else {
                        lo = rs[bin_lo];
                        hi = rs[bin_hi];
                    }
                } else if (grouped(bp.packed)) {
                    lo = 0;
                    hi = (uint32_t)nb;
                }
            } else if (grouped(bp.packed)) {
                lo = 0;
                hi = (uint32_t)nb;  // whole packet = all row bins
            }
            if (hi < lo) hi = lo;
        } else if (grouped(bp.packed)) {
            lo = 0xffffu;  // dead packet: contributes no rows
            hi = 0;
        }
        if (valid && !bp.cuts_inline) cuts[((size_t)j * g.nz + zq) * np + kq] = lo | (hi << 16);  // 16 bits each
        if (pair_work) {
            // records of this (band, plane) over the 16 packets of the tile (lanes 16 i .. 16 i + 15 of a wave
            // share the plane): a row reduction, then one atomic per (band, plane, tile)
            int len = (int)hi - (int)lo;
            len += __builtin_amdgcn_update_dpp(0, len, 0x111, 0xf, 0xf, true);  // row_shr:1
            len += __builtin_amdgcn_update_dpp(0, len, 0x112, 0xf, 0xf, true);  // row_shr:2
            len += __builtin_amdgcn_update_dpp(0, len, 0x114, 0xf, 0xf, true);  // row_shr:4
            len += __builtin_amdgcn_update_dpp(0, len, 0x118, 0xf, 0xf, true);  // row_shr:8
            if ((threadIdx.x & 15) == 15 && len > 0 && z < g.nz) atomicAdd(&pair_work[j * g.nz + z], (uint32_t)len);
        }
    }
}

template <bool STAGED>
__global__ __launch_bounds__(1024) void k_plane_coef(const float* __restrict__ centers,
                                                    const float* __restrict__ planes,
                                                    const uint16_t* __restrict__ rowstart,
                                                    uint32_t* __restrict__ nvalid, int np,
                                                    Geom g, BandPlan bp,
                                                    PlaneCoef* __restrict__ coef,
                                                    uint32_t* __restrict__ cuts)
{
    plane_coef_body<STAGED>(blockIdx.x, centers, planes, rowstart, nvalid, np, g, bp, coef, cuts, nullptr);
}

// the coefficient / cut tables of up to four cameras in one launch; camera c's blocks follow camera c - 1's
struct PrepBlocks {
    unsigned n[kFusedMaxCameras];  // blocks of camera c
};

template <bool STAGED>
__global__ __launch_bounds__(1024) void k_plane_coef_multi(PrepCameras cams, BandPlan bp, PrepBlocks blocks)
{
    unsigned bid = blockIdx.x;
    int c = 0;
#pragma unroll
    for (int k = 0; k < kFusedMaxCameras - 1; ++k)
        if (c == k && cams.n > k + 1 && bid >= blocks.n[k]) {
            bid -= blocks.n[k];
            c = k + 1;
        }
    const PrepCamera& pc = cams.cam[c];
    plane_coef_body<STAGED>(bid, pc.raw.centers, pc.planes, pc.rowstart, pc.nvalid, pc.np, pc.raw.g, bp, pc.coef, pc.cuts, pc.pair_work);
}

// (2b) BandPlan::cuts_inline: the packets' row tables TRANSPOSED -- rsT[bin][packet], rows of `stride` packets -- so
//      that a pass of the vector fill (lane = packet, 64 consecutive packets whose poses are microseconds apart: their
//      bins for one band and plane differ by a row or two) reads its two entries per packet from one or two 128-byte
//      pieces.  64 packets x 64 bins per block through LDS.
__global__ __launch_bounds__(256) void k_transpose_rowstart(const uint16_t* __restrict__ rs, int np, int nb1, int stride,
                                                            uint16_t* __restrict__ rsT)
{
    __shared__ uint16_t tile[64][66];
    const int p0 = (int)blockIdx.x * 64, b0 = (int)blockIdx.y * 64;
    {
        const int bin = b0 + (int)(threadIdx.x & 63);
        for (int i = (int)(threadIdx.x >> 6); i < 64; i += 4) {
            const int p = p0 + i;
            tile[i][threadIdx.x & 63] = (p < np && bin < nb1) ? rs[(size_t)p * nb1 + bin] : (uint16_t)0;
        }
    }
    __syncthreads();
    const int p = p0 + (int)(threadIdx.x & 63);
    if (p >= stride) return;
    for (int i = (int)(threadIdx.x >> 6); i < 64; i += 4) {
        const int bin = b0 + i;
        if (bin < nb1) rsT[(size_t)bin * stride + p] = tile[threadIdx.x & 63][i];
    }
}

// (3) the voting kernel.  Work item = (packet chunk c, band j, plane z): the band's
//     rows [r0-1, r1] of plane z live in LDS (two halo rows, so a bilinear vote never
//     needs a row test), every wave walks packets of the chunk with the packet's
//     coefficients in SGPRs, and the owned rows [r0, r1) are written back with plain
//     coalesced stores -- no global atomics, no memset.
//
//     Block b runs on XCD b % 8 (observed dispatch rule): the blocks of one XCD walk
//     the planes of ONE (chunk, band) pair at a time, so its events stream through
//     that XCD's L2 once instead of once per plane.
// LDS accumulators are 64-bit fixed point (Q33.31), voted with ds_add_u64 (no return).
// Measured on gfx950 (tools/lds_atomic_bench2.hip, cycles per wave instruction, random band
// addresses): ds_add_f32 ~185 (lanes serialised), ds_add_u64 ~11, ds_add_u32 ~6.5.  A 32-bit
// Q8.24 word + carry counter was tried (needs the returning form to detect wraps): the wait
// for the returned value cost more than the cheaper atomic saved (3.5 vs 4.3 Gev/s).
// A weight w = fl(fx*fy) in [0,1] is added as trunc(w * 2^31): exact for w >= 2^-8, off by
// < 2^-31 below, so a voxel holds the EXACT sum of its fp32 weights (to 5e-10 per vote),
// independent of vote order, and is rounded to fp32 once at write-back.  (The CPU reference
// rounds after every += instead.)
using acc_t = unsigned long long;
// k_vote_fuse_argmax keeps two fp32 values and a plane index per band cell in registers: 1024-cell stretches per
// thread.  20 cover the whole LDS (160 KB of 8-byte cells); the vector-fill mappings leave fewer registers: 16.
constexpr int kFusedTracePhases = 64;
__host__ __device__ constexpr int fused_cells_per_thread(int mapping) { return (mapping == 5 || mapping == 6) ? 16 : 20; }
constexpr int kFusedCellsFourCameras = 16;  // four cameras keep TWO fp32 arrays per thread beside the running maxima
constexpr int kFusedCellsTwoPerCu = 10;  // k_vote_fuse_argmax_2cu: half the LDS per workgroup = 10 x 1024 cells
constexpr float kFixScale = 2147483648.f;      // 2^31
constexpr double kFixInv = 1.0 / 2147483648.0;  // 2^-31

// Q33.31 sum -> fp32, ONE rounding.  For v < 2^52 the double whose bits are (1044 << 52) | v is exactly
// 2^21 + v * 2^-31; subtracting 2^21 is exact, so two double-rate instructions replace the 64-bit
// integer -> double conversion (two conversions, a scale and an add).  Larger sums (> 2 M votes in one
// voxel) take the general conversion, exact up to 2^53.
__device__ __forceinline__ float fix_to_float(acc_t v)
{
    if (__builtin_expect((v >> 52) == 0ull, 1))
        return (float)(__longlong_as_double((long long)(v | 0x4140000000000000ull)) - 2097152.0);
    return (float)((double)v * kFixInv);
}

// the four bilinear votes of m identical events (cartesian3dgrid.h:261-270) into the band
__device__ __forceinline__ void vote4(acc_t* __restrict__ band, int idx, int nx, float fx, float fy,
                                      uint32_t m)
{
    const float fx1 = 1.f - fx, fy1 = 1.f - fy;
    // scaling one factor by 2^31 scales the rounded product exactly; w * 2^31 <= 2^31 fits u32;
    // m votes of trunc(w * 2^31) are one vote of m * trunc(w * 2^31) (v_mad_u64_u32), exactly
    const float fxs = fx * kFixScale, fx1s = fx1 * kFixScale;
    acc_t* cell = band + idx;
    __hip_atomic_fetch_add(cell, (acc_t)(unsigned int)(fx1s * fy1) * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(cell + 1, (acc_t)(unsigned int)(fxs * fy1) * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(cell + nx, (acc_t)(unsigned int)(fx1s * fy) * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(cell + nx + 1, (acc_t)(unsigned int)(fxs * fy) * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- lane mapping 8: PAIRED 32-bit cells (dsi_vote_asm.h, DSI_ASM_VOTE_PAIRED19) ----
constexpr int kPairedFracBits = 19;                       // Q.19 weights, rounded
constexpr uint32_t kPairedGuard = 0x80000000u;            // a sub-cell at or above this is reported (4,096 full votes)
__host__ __device__ inline int paired_row_words(int nx) { return 2 * ((nx >> 1) + 1); }  // 8-byte words per band row

__device__ __forceinline__ uint32_t cvt_rpi(float x)  // floor(x + 0.5), the instruction the hand-scheduled vote uses
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return (uint32_t)r;
}

// the compiled twin of DSI_ASM_VOTE_PAIRED19's tail (the IEEE-divide planes of mapping 8 use it): same operations, same bits
__device__ __forceinline__ void vote_paired(char* __restrict__ band_bytes, int xi, int yi, int row_bytes, int cbase, float fx,
                                            float fy, uint32_t m)
{
    const float fy1 = 1.f - fy;
    const float S = 524288.f * (float)m;  // m * 2^19, exact (m <= 1024)
    const float fxs = S * fx, fx1s = S - fxs;
    const uint32_t w00 = cvt_rpi(fx1s * fy1), w10 = cvt_rpi(fxs * fy1);
    const uint32_t w01 = cvt_rpi(fx1s * fy), w11 = cvt_rpi(fxs * fy);
    acc_t* word = reinterpret_cast<acc_t*>(band_bytes + (__mul24(yi, row_bytes) + (((xi >> 1) << 3) + cbase) + (xi & 1) * (row_bytes >> 1)));
    __hip_atomic_fetch_add(word, (acc_t)w00 | ((acc_t)w10 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    acc_t* below = reinterpret_cast<acc_t*>(reinterpret_cast<char*>(word) + row_bytes);
    __hip_atomic_fetch_add(below, (acc_t)w01 | ((acc_t)w11 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the raw sum of cell x of a paired row, as the Q33.31 value the exact mappings hold (so that seam rows, partial volumes
// and the one rounding to fp32 are theirs); *guard |= a sub-cell reached kPairedGuard
__device__ __forceinline__ acc_t paired_cell(const uint2* __restrict__ row, int rw, int x, uint32_t* guard)
{
    const int k = x >> 1;
    uint32_t a, b;
    if (x & 1) {
        a = row[k].y;        // votes at xi = x - 1 (even): W_k's upper cell
        b = row[rw + k].x;   // votes at xi = x (odd):      V_k's lower cell
    } else {
        a = row[k].x;                            // votes at xi = x (even): W_k's lower cell
        b = k > 0 ? row[rw + k - 1].y : 0u;      // votes at xi = x - 1 (odd): V_(k-1)'s upper cell
    }
    *guard |= (a | b) & kPairedGuard;
    return ((acc_t)a + (acc_t)b) << (31 - kPairedFracBits);
}

// owned rows of the band -> fp32 volume (a linear, coalesced copy).  A vote of an event in the band's
// last owned row also lands in the row below, which is the next band's first row: that "carry" row,
// and the next band's own sums for its first row (its "head"), leave the CU as the raw 64-bit
// fixed-point sums -- seam[c][z][j][0][x] = head of band j (j >= 1), seam[c][z][j][1][x] = carry of
// band j (j < bands - 1) -- and k_seam_rows writes fl((head + carry) * 2^-31) into the volume: ONE
// rounding of the exact sum, like every other voxel.  The DSI therefore does not depend on the band
// decomposition (round 2 added fl(carry) to fl(head) in fp32: two roundings on seam rows).  A band
// processes exactly the events with floor(Y) in its OWNED rows: no event is processed by two bands.
__device__ __forceinline__ acc_t* seam_rows(acc_t* seam, int c, int z, int j, const Geom& g, const BandPlan& bp)
{
    return seam + ((((size_t)c * g.nz + z) * bp.bands + j) * 2) * g.nx;
}

// `out` + `off`: the band's first owned voxel in the chunk's volume -- fp32 (the DSI itself: one chunk, no
// accumulation) or, raw != 0, the chunk's PARTIAL volume of raw 64-bit sums, which k_reduce_partials adds up
// exactly over the chunks before the one rounding to fp32: the DSI is fl(exact sum of all votes) for any
// number of chunks as well.
// Work item b -> (pair q = (chunk, band), plane z); false: the block has nothing to do.
// Block b runs on XCD b % 8 (observed dispatch rule).  Groups of 8 pairs: XCD x walks the planes of pair 8k + x, so
// that pair's events stream through that XCD's L2 once instead of once per plane.  The pairs % 8 pairs left over:
// 1, 2 or 4 of them share the XCDs evenly (8, 4 or 2 XCDs each, the planes dealt among those: their events
// stream through 8 / L L2s); any other count is dealt plane by plane over all XCDs (round 2's rule for all).
__device__ __forceinline__ bool item_of(int b, int pairs, int nz, int& q, int& z, bool spread_all = false)
{
    const int full_pairs = (pairs / 8) * 8, full = full_pairs * nz;
    if (b < full) {
        const int xcd = b & 7, s = b >> 3;
        q = (s / nz) * 8 + xcd;
        z = s % nz;
        return true;
    }
    const int r = b - full, L = pairs - full_pairs;
    if ((L == 1 || L == 2 || L == 4) && !spread_all) {  // (spread_all: round 2's rule, experiments builds only)
        const int per = 8 / L, xcd = r & 7, s = r >> 3;  // (full is a multiple of 8: r & 7 == b & 7)
        q = full_pairs + xcd / per;
        z = s * per + xcd % per;
        return z < nz;
    }
    q = full_pairs + r / nz;
    z = r % nz;
    return q < pairs;
}

__host__ __device__ inline int item_count(int pairs, int nz, bool spread_all = false)
{
    const int full_pairs = (pairs / 8) * 8, L = pairs - full_pairs;
    if ((L == 1 || L == 2 || L == 4) && !spread_all) return full_pairs * nz + 8 * ((nz + 8 / L - 1) / (8 / L));
    return pairs * nz;
}

template <int BLOCK, bool CLEAR>
__device__ __forceinline__ void flush_band_t(acc_t* __restrict__ band, int nx, int n_out, void* __restrict__ out, size_t off,
                                             int raw, acc_t* __restrict__ seam_j, int j, int bands)
{
    const int head = j >= 1 ? nx : 0;  // cells that go to the seam buffer instead of the volume
    float* __restrict__ dstf = reinterpret_cast<float*>(out) + off;
    acc_t* __restrict__ dstr = reinterpret_cast<acc_t*>(out) + off;
    for (int i = threadIdx.x; i < n_out; i += BLOCK) {
        const acc_t v = band[i];
        if (i < head)
            seam_j[i] = v;
        else if (raw)
            dstr[i] = v;
        else
            dstf[i] = fix_to_float(v);  // exact in f64, one rounding to f32
        if (CLEAR) band[i] = 0;
    }
    acc_t* src = band + n_out;
    for (int i = threadIdx.x; i < nx; i += BLOCK) {
        if (j < bands - 1) seam_j[nx + i] = src[i];
        if (CLEAR) src[i] = 0;
    }
}

template <int BLOCK>
__device__ __forceinline__ void flush_band(acc_t* __restrict__ band, int nx, int n_out, void* __restrict__ out, size_t off,
                                           int raw, acc_t* __restrict__ seam_j, int j, int bands)
{
    flush_band_t<BLOCK, false>(band, nx, n_out, out, off, raw, seam_j, j, bands);
}

// the same, leaving the band zeroed for the workgroup's next work item (persistent kernel)
template <int BLOCK>
__device__ __forceinline__ void flush_band_and_clear(acc_t* __restrict__ band, int nx, int n_out, void* __restrict__ out,
                                                     size_t off, int raw, acc_t* __restrict__ seam_j, int j, int bands)
{
    flush_band_t<BLOCK, true>(band, nx, n_out, out, off, raw, seam_j, j, bands);
}

// flush_band_t for a band of PAIRED cells (lane mapping 8): a cell's two parts are added, scaled to Q33.31 and leave the CU the
// way the exact sums do.  n_rows owned rows + the carry row; the band is always cleared (persistent or not, it is cheap).
// overflow: one word, set when a sub-cell reached the guard (2^31: half its capacity) -- the caller re-runs in an exact mode.
template <int BLOCK>
__device__ __forceinline__ void flush_band_paired(acc_t* __restrict__ band, int nx, int n_rows, void* __restrict__ out, size_t off,
                                                  int raw, acc_t* __restrict__ seam_j, int j, int bands,
                                                  uint32_t* __restrict__ overflow)
{
    const int rw = (nx >> 1) + 1;
    const uint2* __restrict__ words = reinterpret_cast<const uint2*>(band);
    float* __restrict__ dstf = reinterpret_cast<float*>(out) + off;
    acc_t* __restrict__ dstr = reinterpret_cast<acc_t*>(out) + off;
    const int head = j >= 1 ? nx : 0;  // cells that go to the seam buffer instead of the volume
    const int n_out = n_rows * nx;
    uint32_t guard = 0u;
    for (int i = threadIdx.x; i < n_out + nx; i += BLOCK) {  // (the last nx: the carry row)
        const int r = i / nx, x = i - r * nx;
        const acc_t v = paired_cell(words + (size_t)r * 2 * rw, rw, x, &guard);
        if (i >= n_out) {
            if (j < bands - 1) seam_j[nx + (i - n_out)] = v;
        } else if (i < head)
            seam_j[i] = v;
        else if (raw)
            dstr[i] = v;
        else
            dstf[i] = fix_to_float(v);
    }
    if (__builtin_amdgcn_ballot_w64(guard != 0u) != 0ull && (threadIdx.x & 63) == 0) atomicOr(overflow, 1u);
    __syncthreads();  // every cell has been read
    const int all_words = (n_rows + 1) * 2 * rw;
    for (int i = threadIdx.x; i < all_words; i += BLOCK) band[i] = 0;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_vote_bands(const EvRec* __restrict__ sxy,
                                                      const PlaneCoef* __restrict__ coef,
                                                      const uint32_t* __restrict__ cuts, int np,
                                                      Geom g, BandPlan bp,
                                                      void* __restrict__ out,
                                                      acc_t* __restrict__ seam)
{
    extern __shared__ acc_t band[];
    // block -> (pair q = (chunk, band), plane z).  Groups of 8 pairs: XCD x (= block % 8) walks
    // the planes of pair 8k + x.  The last (pairs % 8) pairs are dealt plane by plane over all
    // XCDs so that no XCD idles (their events then stream through every L2).
    const int b = blockIdx.x;
    const int pairs = bp.chunks * bp.bands;
    int q, z;
    if (!item_of(b, pairs, g.nz, q, z, bp.experiment == 200)) return;
    const int c = q / bp.bands, j = q % bp.bands;
    const int r0 = j * bp.band_rows;
    const int r1 = min(g.ny, r0 + bp.band_rows);
    const int nx = g.nx;
    const int cells = (r1 - r0 + 1) * nx;  // owned rows + the carry row
    for (int i = threadIdx.x; i < cells; i += BLOCK) band[i] = 0;
    __syncthreads();

    const int p_begin = (int)(((long long)np * c) / bp.chunks);
    const int p_end = (int)(((long long)np * (c + 1)) / bp.chunks);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const float L = (float)r0, U = (float)min(r1, g.ny - 1);
    const float xmax = (float)(nx - 1);
    const int row_base = r0;

    // Per-packet metadata (coefficients + this band's run) is fetched with VECTOR loads whose
    // address is the same in every lane, one packet ahead of its use: scalar loads would
    // share lgkmcnt with the LDS atomics and serialise flags -> cuts -> coefficients.
    constexpr int kStride = BLOCK / kWave;
    const uint4* __restrict__ coef4 = reinterpret_cast<const uint4*>(coef);
    const size_t coef_base = (size_t)z * np;                 // coef[z][p]
    const size_t cuts_base = ((size_t)j * g.nz + z) * np;    // cuts[band][z][p]
    auto fetch_meta = [&](int p, uint4& m0, uint2& m1, uint32_t& cu) {
        size_t pz = coef_base + p;
        asm volatile("" : "+v"(pz));  // keep the address in VGPRs => global_load, counted by vmcnt
        m0 = coef4[2 * pz];
        m1 = *reinterpret_cast<const uint2*>(coef4 + 2 * pz + 1);
        cu = cuts[cuts_base + (pz - coef_base)];
    };
    int p = p_begin + wave;
    uint4 m0 = make_uint4(0, 0, 0, 0), n0 = m0;
    uint2 m1 = make_uint2(0, kCoefSkip), n1 = m1;
    uint32_t mcu = 0, ncu = 0;
    if (p < p_end) fetch_meta(p, m0, m1, mcu);
    for (; p < p_end; p += kStride) {
        // unpack the current packet first (this is where its loads are waited for), THEN put
        // the next packet's metadata in flight so that it overlaps this packet's votes
        const float ka = __uint_as_float(__builtin_amdgcn_readfirstlane(m0.x));
        const float kbx = __uint_as_float(__builtin_amdgcn_readfirstlane(m0.y));
        const float kby = __uint_as_float(__builtin_amdgcn_readfirstlane(m0.z));
        const float kd = __uint_as_float(__builtin_amdgcn_readfirstlane(m0.w));
        const float kr = __uint_as_float(__builtin_amdgcn_readfirstlane(m1.x));
        const uint32_t flags = __builtin_amdgcn_readfirstlane(m1.y);
        const uint32_t cu = __builtin_amdgcn_readfirstlane(mcu);
        const int lo = (int)(cu & 0xffffu), hi = (int)(cu >> 16);
        const int pn = p + kStride;
        n1 = make_uint2(0, kCoefSkip);
        if (pn < p_end) fetch_meta(pn, n0, n1, ncu);
        if (!(flags & kCoefSkip) && lo < hi) {
            const EvRec* __restrict__ ev = sxy + (size_t)p * kPacket;
            const bool slow = (flags & kCoefSlow) != 0;
            int i = lo + lane;
            EvRec e = {0.f, 0.f, 0u};
            if (i < hi) e = ev[i];
            for (int base = lo; base < hi; base += kWave) {
                // prefetch the next 64 events of the run while this batch is voted
                const int inext = i + kWave;
                EvRec en = e;
                if (inext < hi) en = ev[inext];
                if (i < hi) {
                    const float nxv = e.x * ka + kbx;  // mapper_emvs_stereo.cpp:194-195
                    const float nyv = e.y * ka + kby;
                    float X, Y;
                    if (slow) {
                        X = nxv / kd;
                        Y = nyv / kd;
                    } else {
                        X = div_rc(nxv, kd, kr);
                        Y = div_rc(nyv, kd, kr);
                    }
                    // cartesian3dgrid.h:255-259 restricted to this band's rows.  (Voting rejected
                    // events into a spare cell instead of branching was tried: 12 % slower.)
                    if (X >= 0.f && X < xmax && Y >= L && Y < U) {
                        const float xf = __builtin_floorf(X), yf = __builtin_floorf(Y);
                        const int idx = __mul24((int)yf - row_base, nx) + (int)xf;
                        vote4(band, idx, nx, X - xf, Y - yf, e.m);  // cartesian3dgrid.h:261-270
                    }
                }
                e = en;
                i = inext;
            }
        }
        m0 = n0;
        m1 = n1;
        mcu = ncu;
    }
    __syncthreads();

    // owned rows are contiguous in the [z][y][x] volume: a linear coalesced copy
    const size_t vol = partial_stride((size_t)g.nx * g.ny * g.nz);
    const size_t off = (size_t)c * vol + ((size_t)z * g.ny + r0) * nx;
    flush_band<BLOCK>(band, nx, (r1 - r0) * nx, out, off, bp.raw_out, seam_rows(seam, c, z, j, g, bp), j, bp.bands);
}

// (3b) the same work item decomposition for SHORT runs (tall or wide grids: a band of a
//     1024-wide plane sees ~20 events of a packet).  A wave takes 64 packets at a time,
//     appends their runs back to back into its 64 lanes and votes whenever the lanes are
//     full, so lane utilisation no longer depends on the run length.  Coefficients are then
//     per lane (gathered from the plane-major table), not per wave.
// floor(X) as an integer in one instruction (saturating, NaN -> 0), instead of floor + convert
__device__ __forceinline__ int floor_to_int(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// The stream of one wave.  SLOW = the plane has coefficient sets outside the range where the
// residual-corrected division is proven (or events / coefficients above 2^40): every lane then
// takes the IEEE divide and the accept test also guards against inf / inf.
//
// Issue rates measured on gfx950 (tools/valu_rate_bench.hip, wave-instructions / clk / CU):
// fp32 add/mul/fma and plain integer add/sub/logic/shift ~1.6-2, everything else (conversions,
// floor/fract, three-operand integer ops, compares, v_cndmask, 24-bit and 64-bit multiplies)
// ~0.95, and v_pk_*_f32 0.95 (no gain over two scalar ops).  The loop below is bound by VALU
// issue, so it is written to need few instructions of the second kind.
template <bool SLOW, bool PAIRED = false>
__device__ __forceinline__ void packed_stream(const EvRec* __restrict__ sxy,
                                              const uint4* __restrict__ coef4,
                                              const uint32_t* __restrict__ cutz,
                                              char* __restrict__ band_bytes, int p_first,
                                              int p_end, int lg_group, int stride, int lane,
                                              int nx, int Li, int Ui, int row_base,
                                              uint32_t dummy_eo, int row_bytes = 0)
{
    // ---- the run stream (all wave-uniform).  The wave owns packets i = 0 .. n_my-1:
    //      packet i is p_first + (i >> lg_group) * stride + (i & (group-1)); it contributes the
    //      records [off, hi) of its grouped storage.
    const int group = 1 << lg_group;
    int n_my = 0;
    if (p_first < p_end) {
        const int passes = (p_end - p_first + stride - 1) / stride;         // >= 1
        const int last = p_first + (passes - 1) * stride;
        n_my = (passes - 1) * group + min(group, p_end - last);
    }
    n_my = __builtin_amdgcn_readfirstlane(n_my);  // (the division runs on the vector unit)
    int i = -1;
    int cur = 0, end = 0;  // absolute record indices (packet * 1024 + slot) of the current run
    int pbase = 0;         // current packet * 1024
    uint32_t mycu = 0;     // lane l: cut word of packet l of the current pass
    uint32_t eo = 0;       // per lane: the record the lane takes next
    const int gmask = __builtin_amdgcn_readfirstlane(group - 1);

    // Append runs to the lanes until 64 are taken or the stream ends; returns the lanes taken.
    // The bookkeeping is scalar work, and the CU issues ONE scalar instruction per clock (for
    // all its waves): compiled from C++ this loop cost ~60 scalar instructions per batch and
    // bounded the kernel (PMC: SQ_INSTS_SALU ~ SQ_INSTS_VALU).  Hand-scheduled: 11 per run
    // piece + 13 per packet, and the lane update is one VALU add under a shifted exec mask
    // (lanes [fill, 64) take consecutive records; those beyond the run are overwritten by the
    // next piece, or retired to the multiplicity-0 record when the stream ends).
    auto fill_batch = [&]() -> int {
        int fill = 0;
        for (;;) {
            int status, t0, t1;
            // (all wave-uniform already; the readfirstlanes only pin them to scalar registers)
            cur = __builtin_amdgcn_readfirstlane(cur);
            end = __builtin_amdgcn_readfirstlane(end);
            i = __builtin_amdgcn_readfirstlane(i);
            pbase = __builtin_amdgcn_readfirstlane(pbase);
            fill = __builtin_amdgcn_readfirstlane(fill);
            asm volatile(
                "Ltop%=:\n\t"
                "s_cmp_ge_i32 %0, %1\n\t"
                "s_cbranch_scc1 Lnext%=\n\t"
                "s_sub_i32 %7, 64, %4\n\t"        // room in the batch
                "s_sub_i32 %8, %1, %0\n\t"        // records left in the run
                "s_min_i32 %7, %7, %8\n\t"        // take
                "s_sub_i32 %8, %0, %4\n\t"        // record of lane 0 if the run started there
                "s_lshl_b64 exec, -1, %4\n\t"     // lanes >= fill
                "v_add_u32 %5, %8, %11\n\t"
                "s_mov_b64 exec, -1\n\t"
                "s_add_i32 %4, %4, %7\n\t"
                "s_add_i32 %0, %0, %7\n\t"
                "s_cmp_lt_u32 %4, 64\n\t"
                "s_cbranch_scc0 Lfull%=\n"        // not full => the run is exhausted
                "Lnext%=:\n\t"
                "s_add_i32 %2, %2, 1\n\t"
                "s_cmp_ge_i32 %2, %9\n\t"
                "s_cbranch_scc1 Leos%=\n\t"
                "s_and_b32 %7, %2, %10\n\t"       // packet within the pass
                "s_cmp_eq_u32 %7, 0\n\t"
                "s_cbranch_scc1 Lreload%=\n\t"
                "v_readlane_b32 %8, %12, %7\n\t"  // its cut word
                "s_addk_i32 %3, 0x400\n\t"
                "s_and_b32 %7, %8, 0xffff\n\t"
                "s_lshr_b32 %8, %8, 16\n\t"
                "s_add_i32 %0, %3, %7\n\t"
                "s_add_i32 %1, %3, %8\n\t"
                "s_branch Ltop%=\n"
                "Lfull%=:\n\t"
                "s_mov_b32 %6, 0\n\t"
                "s_branch Ldone%=\n"
                "Lreload%=:\n\t"
                "s_mov_b32 %6, 1\n\t"
                "s_branch Ldone%=\n"
                "Leos%=:\n\t"
                "s_mov_b32 %6, 2\n"
                "Ldone%=:"
                : "+s"(cur), "+s"(end), "+s"(i), "+s"(pbase), "+s"(fill), "+v"(eo), "=&s"(status),
                  "=&s"(t0), "=&s"(t1)
                : "s"(n_my), "s"(gmask), "v"(lane), "v"(mycu)
                : "scc");
            if (status == 0) break;  // 64 lanes taken
            if (status == 2) {       // end of the stream
                if (fill > 0) eo = lane >= fill ? dummy_eo : eo;
                break;
            }
            // first packet of a pass: one coalesced load of the pass's cut words, waited for
            // HERE (a compiler-visible load still pending later would make its next wait a wait
            // for everything in flight)
            const int pp = __builtin_amdgcn_readfirstlane(p_first + (i >> lg_group) * stride);
            mycu = cutz[min(pp + lane, p_end - 1)];
            const uint32_t cu = __builtin_amdgcn_readfirstlane(mycu);
            pbase = pp << 10;
            cur = pbase + (int)(cu & 0xffffu);
            end = pbase + (int)(cu >> 16);
        }
        return fill;
    };
    // Issue the gathers of the filled batch: plain loads the compiler counts.  (Its waitcnt pass
    // then waits for ALL outstanding loads inside fill_batch -- the register allocator reuses
    // destination registers of the set in flight as temporaries there -- so the prefetch is
    // only partly effective.  Hiding the loads from the compiler in inline assembly fixes that
    // but leaves the in-flight registers unprotected against compiler copies; the hand-scheduled
    // stream below owns its registers instead.)
    typedef float v3f __attribute__((ext_vector_type(3)));
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    auto gather = [&](v3f& ev, v4u& va, uint32_t& vr) {
        const EvRec r = sxy[eo];
        ev.x = r.x;
        ev.y = r.y;
        ev.z = __uint_as_float(r.m);
        const uint32_t co = (eo >> 9) & ~1u;  // 2 * packet
        const uint4 c4 = coef4[co];
        va.x = c4.x;
        va.y = c4.y;
        va.z = c4.z;
        va.w = c4.w;
        vr = *reinterpret_cast<const uint32_t*>(coef4 + co + 1u);
    };
    auto arrived = [&](v3f&, v4u&, uint32_t&) {};
    const int nx8 = nx * 8;
    const int cbase = -row_base * nx8;
    auto vote = [&](const v3f& ev, const v4u& va, uint32_t vr) {
        const float ka = __uint_as_float(va.x), kbx = __uint_as_float(va.y);
        const float kby = __uint_as_float(va.z), kd = __uint_as_float(va.w);
        const float nxv = ev.x * ka + kbx;  // mapper_emvs_stereo.cpp:194-195
        const float nyv = ev.y * ka + kby;
        float X, Y;
        if (SLOW) {
            X = nxv / kd;
            Y = nyv / kd;
        } else {
            const float kr = __uint_as_float(vr);
            X = div_rc(nxv, kd, kr);
            Y = div_rc(nyv, kd, kr);
        }
        // cartesian3dgrid.h:255-259 restricted to this band's rows, as ONE integer test:
        // with xi = floor(X), yi = floor(Y) (the conversion saturates, NaN -> 0)
        //   0 <= X < nx-1  <=>  xi >= 0 and nx-2-xi >= 0        (X = -0.0 -> 0, accepted like >= 0.f)
        //   L <= Y < U     <=>  yi-Li >= 0 and Ui-1-yi >= 0     (L, U are integers)
        // all four hold iff the OR of the four values has a clear sign bit.  On the fast path X
        // and Y are finite by construction (k_plane_coef); on the slow path the only NaN source
        // is inf / inf, excluded by asking for finite numerators.
        const int xi = floor_to_int(X), yi = floor_to_int(Y);
        int sgn = xi | (nx - 2 - xi) | (yi - Li) | (Ui - 1 - yi);
        if (SLOW)
            sgn |= (fabsf(nxv) < __builtin_inff() && fabsf(nyv) < __builtin_inff()) ? 0 : -1;
        if (sgn >= 0) {
            // accepted => X, Y >= 0, where v_fract is exactly X - floor(X)
            const float fx = __builtin_amdgcn_fractf(X), fy = __builtin_amdgcn_fractf(Y);
            if (PAIRED) {  // lane mapping 8: two atomics on paired 32-bit cells
                vote_paired(band_bytes, xi, yi, row_bytes, -row_base * row_bytes, fx, fy, __float_as_uint(ev.z));
            } else {
                acc_t* cell = reinterpret_cast<acc_t*>(band_bytes + (__mul24(yi, nx8) + ((xi << 3) + cbase)));
                vote4(cell, 0, nx, fx, fy, __float_as_uint(ev.z));  // cartesian3dgrid.h:261-270
            }
        }
    };

    // two register sets: while the votes of one batch run, the gathers of the next are in flight
    // (the gathers are issued unconditionally -- an empty batch re-reads valid stale records --
    //  so that exactly three loads are newer than the ones a vote waits for)
    v3f eA, eB;
    v4u caA, caB;
    uint32_t crA, crB;
    int nA = fill_batch();
    gather(eA, caA, crA);
    while (nA > 0) {
        const int nB = fill_batch();
        gather(eB, caB, crB);
        arrived(eA, caA, crA);
        vote(eA, caA, crA);
        if (nB == 0) break;
        nA = fill_batch();
        gather(eA, caA, crA);
        arrived(eB, caB, crB);
        vote(eB, caB, crB);
    }
}

// ---- lane mapping 5: VECTOR fill.  The packed streams above append runs to the lanes with
// scalar bookkeeping (11 scalar instructions per run piece + 13 per packet, ~3 taken branches
// each); with short runs -- wide grids: ~19 records per (packet, band) at 1024 x 1024, 4.4 run
// pieces per batch -- a wave then spends ~80 dependent scalar instructions per batch and, with
// only 16 waves per CU (the band fills the LDS), nothing hides that latency (PMC at
// 1024x1024x256: scalar unit 67 % busy, LDS 37 %, 123 CU-clocks per batch).  Here the slot ->
// record mapping of a whole pass of up to 64 packets is computed with vector operations:
//   lane l = packet l of the pass: run length len_l, inclusive prefix incl_l (wave scan);
//   the non-empty packets are compacted (ballot + mbcnt + ds_permute) into tables
//   D_c = first record - first slot, P_c = packet index;
//   the LAST slot of every run sets one bit in a per-wave LDS bit array (ds_or_b64), read back
//   as one 64-bit word per batch (lane j = word j);
//   for batch b, slot lane i belongs to compacted packet  c = #tail bits before slot i
//   = (tails before the batch, a scalar) + v_mbcnt(word_b), and its record is D_c + slot.
// Per batch: 3 v_readlane, 2 v_mbcnt, 2 ds_bpermute, a few adds -- no loop over packets.
// The arithmetic of a vote is the function the other compiled streams use, so all mappings give
// the same bits.
template <bool SLOW>
__device__ __forceinline__ void vote_record(char* __restrict__ band_bytes, float ex, float ey,
                                            uint32_t em, uint4 va, uint32_t vr, int nx, int nx8,
                                            int cbase, int Li, int Ui)
{
    const float ka = __uint_as_float(va.x), kbx = __uint_as_float(va.y);
    const float kby = __uint_as_float(va.z), kd = __uint_as_float(va.w);
    const float nxv = ex * ka + kbx;  // mapper_emvs_stereo.cpp:194-195
    const float nyv = ey * ka + kby;
    float X, Y;
    if (SLOW) {
        X = nxv / kd;
        Y = nyv / kd;
    } else {
        const float kr = __uint_as_float(vr);
        X = div_rc(nxv, kd, kr);
        Y = div_rc(nyv, kd, kr);
    }
    // cartesian3dgrid.h:255-259 restricted to the band's rows, as one integer sign test (see
    // packed_stream)
    const int xi = floor_to_int(X), yi = floor_to_int(Y);
    int sgn = xi | (nx - 2 - xi) | (yi - Li) | (Ui - 1 - yi);
    if (SLOW)
        sgn |= (fabsf(nxv) < __builtin_inff() && fabsf(nyv) < __builtin_inff()) ? 0 : -1;
    if (sgn >= 0) {
        const float fx = __builtin_amdgcn_fractf(X), fy = __builtin_amdgcn_fractf(Y);
        acc_t* cell = reinterpret_cast<acc_t*>(band_bytes + (__mul24(yi, nx8) + ((xi << 3) + cbase)));
        vote4(cell, 0, nx, fx, fy, em);  // cartesian3dgrid.h:261-270
    }
}

// inclusive prefix sum over the 64 lanes with DPP moves only (no LDS round trips: the per-pass set-up
// of the vector fill sits on a wave's critical path while the LDS queue is full of votes)
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/)
{
    // row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143; lanes a mask disables
    // (and reads beyond a row) contribute `old` = 0
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, true);
    return s;
}

// BandPlan::cuts_inline: what a pass of the vector fill needs to find its packets' runs without a cut table.  Per item
// (band, plane): the band's Y interval; per packet two steps -- (1) the second half of its PlaneCoef {r, flags, d_a, by_a},
// (2) two entries of the transposed row table.  The streams issue (1) two stretches and (2) one stretch ahead of the votes.
struct InlineCuts {
    const uint16_t* rsT;  // nullptr: cut table
    uint32_t stride2;     // BYTES per row of rsT (the whole table is < 4 GB: vote_device)
    int nb;               // last entry of a packet's table: all its records
    int pad;
    float L, U;
};

__device__ __forceinline__ InlineCuts inline_cuts_of(const uint32_t* cuts, const Geom& g, const BandPlan& bp, int j)
{
    InlineCuts ic{};
    if (!bp.cuts_inline) return ic;
    ic.rsT = reinterpret_cast<const uint16_t*>(cuts);
    ic.stride2 = 2u * (uint32_t)bp.rs_stride;
    ic.pad = bp.row_pad;
    ic.nb = g.ny + 2 * bp.row_pad + 2;
    band_y_interval(j, g, bp, &ic.L, &ic.U);
    return ic;
}

// step 2 for the packet at byte column p2 (= 2 * packet; a valid packet of the plane): the byte offsets of its two table
// entries.  c1 = {r, flags | margin << 16, d_a, by_a}.  A vector instruction here is paid 1,500 times per wave and work
// item at configs[4]'s size, beside ~50 per batch of votes: the bins are band_row_bins' with the widening taken from the
// packet's own bound (PlaneCoef::flags >> 16, 1/256 rows: >= every band's), fused multiply-adds, one v_cvt_flr each and
// the clamps on integers -- a superset of that superset; the vote re-tests every event exactly.
__device__ __forceinline__ void inline_cut_entries(const InlineCuts& ic, uint4 c1, uint32_t p2, uint32_t* at_lo, uint32_t* at_hi)
{
    const float d_a = __uint_as_float(c1.z), by_a = __uint_as_float(c1.w);
    const float M = (float)(c1.y >> 16) * (1.f / 256.f);
    const float ya = __builtin_fmaf(ic.L, d_a, -by_a), yb = __builtin_fmaf(ic.U, d_a, -by_a);  // (finite: kCoefInvertible's range)
    const int lo = floor_to_int(fminf(ya, yb) - M) + (ic.pad + 1);  // events in bins below bin(floor(ymin))
    const int hi = floor_to_int(fmaxf(ya, yb) + M) + (ic.pad + 2);  // events in bins up to and including bin(floor(ymax))
    int bin_lo = min(max(lo, 0), ic.nb - 1), bin_hi = min(max(hi, 1), ic.nb);
    if (!(c1.y & kCoefInvertible)) {  // entry 0 is 0, entry nb the packet's records: the whole packet
        bin_lo = 0;
        bin_hi = ic.nb;
    }
    *at_lo = (uint32_t)bin_lo * ic.stride2 + p2;
    *at_hi = (uint32_t)bin_hi * ic.stride2 + p2;
}

__device__ __forceinline__ uint32_t inline_cut_load(const InlineCuts& ic, uint32_t byte_offset)
{
    return *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(ic.rsT) + byte_offset);
}

// the cut word lo | hi << 16 of what the two loads returned (k_plane_coef's rules: a dead pair has no run, hi >= lo)
__device__ __forceinline__ uint32_t inline_cut_word(uint32_t flags, uint32_t lo, uint32_t hi)
{
    if (flags & kCoefSkip) return 0u;
    return lo | (max(hi, lo) << 16);
}

template <bool SLOW>
__device__ __forceinline__ void vfill_stream(const EvRec* __restrict__ sxy,
                                             const uint4* __restrict__ coef4,
                                             const uint32_t* __restrict__ cutz,
                                             char* __restrict__ band_bytes,
                                             unsigned long long* __restrict__ scratch /* 64 words of this wave */,
                                             int p_first, int p_end, int lg_pass, int stride,
                                             int lane, int nx, int Li, int Ui, int row_base,
                                             uint32_t dummy_eo, const InlineCuts& ic)
{
    if (Ui - 1 < Li) return;  // the band accepts no row
    const int pass = 1 << lg_pass;
    const int nx8 = nx * 8;
    const int cbase = -row_base * nx8;
    // ---- state of the batch generator (wave-uniform unless noted)
    int pass_base = p_first - stride;  // advanced before use
    int T = 0;                          // records of the current pass
    int rbase = 0;                      // first slot of the current range of 4096 slots
    int b = 0, nb = 0;                  // batch within the range, batches of the range
    int Cbase = 0;                      // compacted packets that end before the range
    int Dc = 0, Pc = 0;                 // per lane c: tables of the c-th non-empty packet
    int incl = 0, len = 0;              // per lane l: packet l of the pass
    unsigned long long w = 0;           // per lane j: tail bits of batch j of the range
    int cbefore = 0;                    // per lane j: tails of the range before batch j

    int rtot = 0;                       // tails of the current range
    auto build_range = [&]() {
        scratch[lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int ts = incl - 1 - rbase;  // tail slot of packet `lane`, relative to the range
        if (len > 0 && ts >= 0 && ts < 4096)
            atomicOr(&scratch[ts >> 6], 1ull << (ts & 63));
        // (same wave: its LDS operations complete in order; the fences only pin the compiler)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        w = scratch[lane];
        const int pc = __builtin_popcountll(w);
        const int inc = wave_incl_scan(pc, lane);
        cbefore = inc - pc;
        rtot = __builtin_amdgcn_readlane(inc, 63);
        nb = min(64, (T - rbase + 63) >> 6);
        b = 0;
    };
    // advance to the next batch; false at the end of the stream.  rec: record index per lane,
    // pk: packet whose coefficients the lane needs
    auto next_batch = [&](uint32_t& rec, int& pk) -> bool {
        while (b >= nb) {
            if (T > 0 && rbase + 4096 < T) {
                Cbase += rtot;
                rbase += 4096;
                build_range();
                continue;
            }
            pass_base += stride;
            if (pass_base >= p_end) return false;
            const int p = pass_base + lane;
            uint32_t cu = 0;
            if (lane < pass && p < p_end) {
                if (ic.rsT) {  // (this compiled stream is the A/B twin and the IEEE-divide planes' path: no prefetch)
                    const uint4 c1 = coef4[2 * (size_t)p + 1];
                    uint32_t at_lo, at_hi;
                    inline_cut_entries(ic, c1, 2u * (uint32_t)p, &at_lo, &at_hi);
                    cu = inline_cut_word(c1.y, inline_cut_load(ic, at_lo), inline_cut_load(ic, at_hi));
                } else {
                    cu = cutz[p];
                }
            }
            const int lo = (int)(cu & 0xffffu), hi = (int)(cu >> 16);
            len = max(hi - lo, 0);
            incl = wave_incl_scan(len, lane);
            T = __builtin_amdgcn_readlane(incl, 63);
            rbase = 0;
            Cbase = 0;
            b = nb = 0;
            if (T == 0) continue;
            const int D = (p << 10) + lo - (incl - len);
            const unsigned long long ne = __builtin_amdgcn_ballot_w64(len > 0);
            const int c = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ne >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)ne, 0u));
            // push (D, p) of the non-empty packets to lanes 0, 1, 2, ...; empty ones to an unused lane
            const int dst = len > 0 ? c : 63 - (lane - c);
            Dc = __builtin_amdgcn_ds_permute(dst << 2, D);
            Pc = __builtin_amdgcn_ds_permute(dst << 2, p);
            build_range();
        }
        const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)w, b);
        const uint32_t whi = (uint32_t)__builtin_amdgcn_readlane((int)(w >> 32), b);
        const int cb = __builtin_amdgcn_readlane(cbefore, b) + Cbase;
        const int c = cb + (int)__builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
        const int slot = rbase + (b << 6) + lane;
        const int Dl = __builtin_amdgcn_ds_bpermute(c << 2, Dc);
        const int Pl = __builtin_amdgcn_ds_bpermute(c << 2, Pc);
        const bool live = slot < T;
        rec = live ? (uint32_t)(Dl + slot) : dummy_eo;
        pk = live ? Pl : p_first;
        ++b;
        return true;
    };
    struct Loaded {
        EvRec e;
        uint4 va;
        uint32_t vr;
    };
    auto gather = [&](uint32_t rec, int pk, Loaded& L) {
        L.e = sxy[rec];
        L.va = coef4[2 * (size_t)pk];
        L.vr = *reinterpret_cast<const uint32_t*>(coef4 + 2 * (size_t)pk + 1);
    };
    auto vote = [&](const Loaded& L) {
        vote_record<SLOW>(band_bytes, L.e.x, L.e.y, L.e.m, L.va, L.vr, nx, nx8, cbase, Li, Ui);
    };
    uint32_t rec;
    int pk;
    Loaded A, B;
    if (!next_batch(rec, pk)) return;
    gather(rec, pk, A);
    for (;;) {
        const bool haveB = next_batch(rec, pk);
        if (haveB) gather(rec, pk, B);
        vote(A);
        if (!haveB) break;
        const bool haveA = next_batch(rec, pk);
        if (haveA) gather(rec, pk, A);
        vote(B);
        if (!haveA) break;
    }
}

// The same stream, fast path only (no IEEE-divide planes), written in gfx950 assembly.
//
// Why: PMC counters of the compiled stream (10 M events x 100 planes) show ~61 scalar and ~63
// vector instructions per 64-lane batch at ~78 CU-clocks per batch: the scalar unit (one
// instruction per clock per CU, shared by all its waves) is ~80 % busy, mostly with compiler
// glue around the run bookkeeping, and the prefetch registers get shuffled.  Written by hand
// the batch costs ~34 scalar + ~46 vector instructions + 4 ds_add_u64, which leaves the LDS
// atomic unit (~11 clocks per ds_add_u64 wave instruction) as the next bound.
//
// Register map (all listed as clobbers; the operands are read-only):
//   s40 cur  s41 end  s42 packet counter i  s43 packet*1024  s44 fill  s45,s46 temporaries
//   s47 lanes of the batch in set A, s48 of set B
//   v40 record index per lane (eo)   v41 cut words of the pass
//   set A: v[42:44] record (x0,y0,m)  v45 r  v[46:49] a,bx,by,d      set B: v[50:52] v53 v[54:57]
//   (register tuples must be even-aligned on gfx950)
//   v36-v39, v58-v63 temporaries
// The arithmetic of VOTE is instruction for instruction what packed_stream<false> compiles to
// (mapper_emvs_stereo.cpp:194-195, div_rc, cartesian3dgrid.h:255-270), so both give the same bits.
#define DSI_ASM_FILL(NOUT, L)                                                                      \
    "s_mov_b32 s44, 0\n"                                                                            \
    "Ltop" L "%=:\n\t"                                                                              \
    "s_cmp_ge_i32 s40, s41\n\t"                                                                     \
    "s_cbranch_scc1 Lnext" L "%=\n\t"                                                               \
    "s_sub_i32 s45, 64, s44\n\t"          /* room in the batch */                                   \
    "s_sub_i32 s46, s41, s40\n\t"         /* records left in the run */                             \
    "s_min_i32 s45, s45, s46\n\t"         /* take */                                                \
    "s_sub_i32 s46, s40, s44\n\t"         /* record of lane 0 if the run started there */           \
    "s_lshl_b64 exec, -1, s44\n\t"        /* lanes >= fill */                                       \
    "v_add_u32 v40, s46, %15\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                        \
    "s_add_i32 s44, s44, s45\n\t"                                                                   \
    "s_add_i32 s40, s40, s45\n\t"                                                                   \
    "s_cmp_lt_u32 s44, 64\n\t"                                                                      \
    "s_cbranch_scc0 Ldone" L "%=\n"       /* not full => the run is exhausted */                    \
    "Lnext" L "%=:\n\t"                                                                             \
    "s_add_i32 s42, s42, 1\n\t"                                                                     \
    "s_cmp_ge_i32 s42, %3\n\t"                                                                      \
    "s_cbranch_scc1 Leos" L "%=\n\t"                                                                \
    "s_and_b32 s45, s42, %4\n\t"          /* packet within the pass */                              \
    "s_cmp_eq_u32 s45, 0\n\t"                                                                       \
    "s_cbranch_scc1 Lreload" L "%=\n\t"                                                             \
    "v_readlane_b32 s46, v41, s45\n\t"    /* its cut word */                                        \
    "s_addk_i32 s43, 0x400\n"                                                                       \
    "Lhave" L "%=:\n\t"                                                                             \
    "s_and_b32 s45, s46, 0xffff\n\t"                                                                \
    "s_lshr_b32 s46, s46, 16\n\t"                                                                   \
    "s_add_i32 s40, s43, s45\n\t"                                                                   \
    "s_add_i32 s41, s43, s46\n\t"                                                                   \
    "s_branch Ltop" L "%=\n"                                                                        \
    "Lreload" L "%=:\n\t"                 /* first packet of a pass: its cut words were prefetched */ \
    "s_lshr_b32 s45, s42, %7\n\t"         /* into v35 one pass ago (loads return in order: once a    */ \
    "s_mul_i32 s45, s45, %6\n\t"          /* GATHER has been issued since, all but the 3 newest      */ \
    "s_add_i32 s45, s45, %5\n\t"          /* loads include them; first packet of the pass:)          */ \
    "s_lshl_b32 s43, s45, 10\n\t"                                                                   \
    "s_cmp_eq_u32 s50, 0\n\t"                                                                       \
    "s_cbranch_scc1 Lwall" L "%=\n\t"                                                               \
    "s_waitcnt vmcnt(3)\n\t"                                                                        \
    "s_branch Lgot" L "%=\n"                                                                        \
    "Lwall" L "%=:\n\t"                                                                             \
    "s_waitcnt vmcnt(0)\n"                                                                          \
    "Lgot" L "%=:\n\t"                                                                              \
    "v_mov_b32 v41, v35\n\t"                                                                        \
    "s_add_i32 s45, s45, %6\n\t"          /* the next pass's cut words travel during this pass */   \
    "v_add_u32 v58, s45, %15\n\t"                                                                   \
    "v_min_i32 v58, %8, v58\n\t"                                                                    \
    "v_lshlrev_b32 v58, 2, v58\n\t"                                                                 \
    "global_load_dword v35, v58, %2\n\t"                                                            \
    "s_mov_b32 s50, 0\n\t"                                                                          \
    "v_readfirstlane_b32 s46, v41\n\t"                                                              \
    "s_branch Lhave" L "%=\n"                                                                       \
    "Leos" L "%=:\n\t"                    /* stream over: unreached lanes -> multiplicity-0 record */ \
    "s_cmp_eq_u32 s44, 0\n\t"                                                                       \
    "s_cbranch_scc1 Ldone" L "%=\n\t"                                                               \
    "s_lshl_b64 exec, -1, s44\n\t"                                                                  \
    "v_mov_b32 v40, %14\n\t"                                                                        \
    "s_mov_b64 exec, -1\n"                                                                          \
    "Ldone" L "%=:\n\t"                                                                             \
    "s_mov_b32 " NOUT ", s44\n\t"

// DSI_ASM_GATHER, DSI_ASM_VOTE: dsi_vote_asm.h

__device__ __forceinline__ void packed_stream_asm(const EvRec* sxy, const uint4* coef4,
                                                  const uint32_t* cutz, char* band_bytes,
                                                  int p_first, int p_end, int lg_group, int stride,
                                                  int lane, int nx, int Li, int Ui, int row_base,
                                                  uint32_t dummy_eo)
{
    const int group = 1 << lg_group;
    int n_my = 0;
    if (p_first < p_end) {
        const int passes = (p_end - p_first + stride - 1) / stride;
        const int last = p_first + (passes - 1) * stride;
        n_my = (passes - 1) * group + min(group, p_end - last);
    }
    if (Ui - 1 < Li) n_my = 0;  // no acceptable row (the unsigned range test needs Ui-1-Li >= 0)
    // every operand is wave-uniform; the readfirstlanes pin them to scalar registers
    const int s_n_my = __builtin_amdgcn_readfirstlane(n_my);
    const int s_gmask = __builtin_amdgcn_readfirstlane(group - 1);
    const int s_p_first = __builtin_amdgcn_readfirstlane(p_first);
    const int s_stride = __builtin_amdgcn_readfirstlane(stride);
    const int s_lg = __builtin_amdgcn_readfirstlane(lg_group);
    const int s_p_last = __builtin_amdgcn_readfirstlane(p_end - 1);
    const int s_nx8 = __builtin_amdgcn_readfirstlane(nx * 8);
    const int lds_base = (int)(uintptr_t)band_bytes;  // LDS byte offset of the band
    const int s_cbase = __builtin_amdgcn_readfirstlane(lds_base - row_base * nx * 8);
    const int s_nxm2 = __builtin_amdgcn_readfirstlane(nx - 2);
    const int s_Li = __builtin_amdgcn_readfirstlane(Li);
    // rows the band accepts: 0 <= yi - Li <= Ui - 1 - Li  (a band with no acceptable row gets no stream)
    const int s_Uim1 = __builtin_amdgcn_readfirstlane(Ui - 1 - Li);
    const uint32_t s_dummy = __builtin_amdgcn_readfirstlane(dummy_eo);
    asm volatile(
        "s_mov_b32 s42, -1\n\t"
        "s_mov_b32 s40, 0\n\t"
        "s_mov_b32 s41, 0\n\t"
        "s_mov_b32 s43, 0\n\t"
        "v_mov_b32 v40, 0\n\t"
        "v_mov_b32 v41, 0\n\t"
        "s_mov_b32 s50, 0\n\t"
        "v_add_u32 v58, %5, %15\n\t"        // the first pass's cut words
        "v_min_i32 v58, %8, v58\n\t"
        "v_max_i32 v58, 0, v58\n\t"          // (a chunk without packets has p_end - 1 = -1)
        "v_lshlrev_b32 v58, 2, v58\n\t"
        "global_load_dword v35, v58, %2\n\t"
        DSI_ASM_FILL("s47", "a")
        DSI_ASM_GATHER("v[42:44]", "v[46:49]", "v45")
        "s_cmp_eq_u32 s47, 0\n\t"
        "s_cbranch_scc1 Lend%=\n"
        "Lloop%=:\n\t"
        DSI_ASM_FILL("s48", "b")
        DSI_ASM_GATHER("v[50:52]", "v[54:57]", "v53")
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "s_cmp_eq_u32 s48, 0\n\t"
        "s_cbranch_scc1 Lend%=\n\t"
        DSI_ASM_FILL("s47", "c")
        DSI_ASM_GATHER("v[42:44]", "v[46:49]", "v45")
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "s_cmp_lg_u32 s47, 0\n\t"
        "s_cbranch_scc1 Lloop%=\n"
        "Lend%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        :
        : "s"(sxy), "s"(coef4), "s"(cutz), "s"(s_n_my), "s"(s_gmask), "s"(s_p_first), "s"(s_stride),
          "s"(s_lg), "s"(s_p_last), "s"(s_nx8), "s"(s_cbase), "s"(s_nxm2), "s"(s_Li), "s"(s_Uim1),
          "s"(s_dummy), "v"(lane)
        : "memory", "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50",
          "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
          "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",
          "v62", "v63");
}

// ---- the packed stream with DEALT passes (the fused kernel: one workgroup per CU, nothing else hides a
// workgroup's tail).  With fixed assignments (wave w takes passes w, w + 16, ...) the waves of a workgroup
// finish a work item at very different times -- 512 x 512 x 200, 489 packets in passes of 8: 61 passes over
// 16 waves is 3 or 4 each, and waves with EQUAL work still spread by 20 % (time stamps of
// tools/fused_trace.py: first wave done after 11.3 us, median 13.7, last 16.8) -- and everybody waits at the
// item's barrier.  Here wave w starts with passes w and w + 16 and draws every further pass from a counter in
// LDS (set to 32 by the item's set-up): ds_add_rtn_u32 by one lane, two passes ahead, so that neither the
// atomic's round trip through the vote-filled LDS queue nor the next pass's cut words are ever waited for:
//   at the start of pass k:  its cut words (requested at pass k-1) arrive; the index of pass k+1 (drawn at
//   pass k-1) is read; pass k+1's cut words are requested; pass k+2 is drawn.
// The sums are integer, so which wave votes a pass changes no bit.
//   s53 current pass, s55 next pass, s54 packets of the current pass, s42 packet within the pass,
//   v34 the drawn pass index (lane 0); everything else as in packed_stream_asm.
#define DSI_ASM_FILL_DEAL(NOUT, L)                                                                 \
    "s_mov_b32 s44, 0\n"                                                                            \
    "Ltop" L "%=:\n\t"                                                                              \
    "s_cmp_ge_i32 s40, s41\n\t"                                                                     \
    "s_cbranch_scc1 Lnext" L "%=\n\t"                                                               \
    "s_sub_i32 s45, 64, s44\n\t"          /* room in the batch */                                   \
    "s_sub_i32 s46, s41, s40\n\t"         /* records left in the run */                             \
    "s_min_i32 s45, s45, s46\n\t"         /* take */                                                \
    "s_sub_i32 s46, s40, s44\n\t"         /* record of lane 0 if the run started there */           \
    "s_lshl_b64 exec, -1, s44\n\t"        /* lanes >= fill */                                       \
    "v_add_u32 v40, s46, %15\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                        \
    "s_add_i32 s44, s44, s45\n\t"                                                                   \
    "s_add_i32 s40, s40, s45\n\t"                                                                   \
    "s_cmp_lt_u32 s44, 64\n\t"                                                                      \
    "s_cbranch_scc0 Ldone" L "%=\n"       /* not full => the run is exhausted */                    \
    "Lnext" L "%=:\n\t"                                                                             \
    "s_add_i32 s42, s42, 1\n\t"                                                                     \
    "s_cmp_ge_i32 s42, s54\n\t"                                                                     \
    "s_cbranch_scc1 Lreload" L "%=\n\t"   /* the pass is exhausted */                               \
    "v_readlane_b32 s46, v41, s42\n\t"    /* the packet's cut word */                               \
    "s_addk_i32 s43, 0x400\n"                                                                       \
    "Lhave" L "%=:\n\t"                                                                             \
    "s_and_b32 s45, s46, 0xffff\n\t"                                                                \
    "s_lshr_b32 s46, s46, 16\n\t"                                                                   \
    "s_add_i32 s40, s43, s45\n\t"                                                                   \
    "s_add_i32 s41, s43, s46\n\t"                                                                   \
    "s_branch Ltop" L "%=\n"                                                                        \
    "Lreload" L "%=:\n\t"                                                                           \
    "s_mov_b32 s53, s55\n\t"              /* the pass whose cut words are in flight */              \
    "s_cmp_ge_i32 s53, %3\n\t"                                                                      \
    "s_cbranch_scc1 Leos" L "%=\n\t"                                                                \
    "s_lshl_b32 s45, s53, %7\n\t"                                                                   \
    "s_add_i32 s45, s45, %5\n\t"          /* its first packet */                                    \
    "s_sub_i32 s54, %6, s45\n\t"                                                                    \
    "s_min_i32 s54, s54, %4\n\t"          /* its packets (the last pass may be short) */            \
    "s_lshl_b32 s43, s45, 10\n\t"                                                                   \
    "s_cmp_eq_u32 s50, 0\n\t"                                                                       \
    "s_cbranch_scc1 Lwall" L "%=\n\t"                                                               \
    "s_waitcnt vmcnt(3)\n\t"              /* (a GATHER has been issued since: all but the 3 newest) */ \
    "s_branch Lgot" L "%=\n"                                                                        \
    "Lwall" L "%=:\n\t"                                                                             \
    "s_waitcnt vmcnt(0)\n"                                                                          \
    "Lgot" L "%=:\n\t"                                                                              \
    "v_mov_b32 v41, v35\n\t"                                                                        \
    "s_waitcnt lgkmcnt(0)\n\t"            /* the draw of one pass ago */                            \
    "v_readfirstlane_b32 s55, v34\n\t"                                                              \
    "s_lshl_b32 s45, s55, %7\n\t"                                                                   \
    "s_add_i32 s45, s45, %5\n\t"                                                                    \
    "v_add_u32 v58, s45, %15\n\t"         /* the next pass's cut words travel during this pass */   \
    "v_min_i32 v58, %8, v58\n\t"                                                                    \
    "v_lshlrev_b32 v58, 2, v58\n\t"                                                                 \
    "global_load_dword v35, v58, %2\n\t"                                                            \
    "s_mov_b32 s50, 0\n\t"                                                                          \
    "s_mov_b64 exec, 1\n\t"                                                                         \
    "ds_add_rtn_u32 v34, %16, %17\n\t"    /* draw the pass after the next */                        \
    "s_mov_b64 exec, -1\n\t"                                                                        \
    "s_mov_b32 s42, 0\n\t"                                                                          \
    "v_readfirstlane_b32 s46, v41\n\t"                                                              \
    "s_branch Lhave" L "%=\n"                                                                       \
    "Leos" L "%=:\n\t"                    /* stream over: unreached lanes -> multiplicity-0 record */ \
    "s_mov_b32 s40, 0\n\t"                                                                          \
    "s_mov_b32 s41, 0\n\t"                                                                          \
    "s_mov_b32 s42, 0\n\t"                                                                          \
    "s_mov_b32 s54, 0\n\t"                /* (a later fill comes straight back here) */             \
    "s_cmp_eq_u32 s44, 0\n\t"                                                                       \
    "s_cbranch_scc1 Ldone" L "%=\n\t"                                                               \
    "s_lshl_b64 exec, -1, s44\n\t"                                                                  \
    "v_mov_b32 v40, %14\n\t"                                                                        \
    "s_mov_b64 exec, -1\n"                                                                          \
    "Ldone" L "%=:\n\t"                                                                             \
    "s_mov_b32 " NOUT ", s44\n\t"

// passes of `1 << lg_group` packets of [p_begin, p_end); wave `wave` of `n_waves`; pass_counter: LDS word set
// to 2 * n_waves before the workgroup's waves enter
// (the whole stream as one asm statement; VOTE = DSI_ASM_VOTE or, lane mapping 8, DSI_ASM_VOTE_PAIRED19)
#define DSI_DEALT_STREAM_ASM(VOTE)                                                                                      \
    asm volatile(                                                                                                       \
        "s_mov_b32 s42, 0\n\t"                                                                                          \
        "s_mov_b32 s54, 0\n\t"                                                                                          \
        "s_mov_b32 s40, 0\n\t"                                                                                          \
        "s_mov_b32 s41, 0\n\t"                                                                                          \
        "s_mov_b32 s43, 0\n\t"                                                                                          \
        "s_mov_b32 s55, %18\n\t"             /* the first pass: the wave's index ... */                                 \
        "v_mov_b32 v34, %19\n\t"             /* ... the second: n_waves further */                                      \
        "v_mov_b32 v40, 0\n\t"                                                                                          \
        "v_mov_b32 v41, 0\n\t"                                                                                          \
        "s_mov_b32 s50, 0\n\t"                                                                                          \
        "v_mov_b32 v35, %20\n\t"             /* the first pass's cut words (already here) */                            \
        DSI_ASM_FILL_DEAL("s47", "a")                                                                                   \
        DSI_ASM_GATHER("v[42:44]", "v[46:49]", "v45")                                                                   \
        "s_cmp_eq_u32 s47, 0\n\t"                                                                                       \
        "s_cbranch_scc1 Lend%=\n"                                                                                       \
        "Lloop%=:\n\t"                                                                                                  \
        DSI_ASM_FILL_DEAL("s48", "b")                                                                                   \
        DSI_ASM_GATHER("v[50:52]", "v[54:57]", "v53")                                                                   \
        "s_waitcnt vmcnt(3)\n\t"                                                                                        \
        VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")                                                    \
        "s_cmp_eq_u32 s48, 0\n\t"                                                                                       \
        "s_cbranch_scc1 Lend%=\n\t"                                                                                     \
        DSI_ASM_FILL_DEAL("s47", "c")                                                                                   \
        DSI_ASM_GATHER("v[42:44]", "v[46:49]", "v45")                                                                   \
        "s_waitcnt vmcnt(3)\n\t"                                                                                        \
        VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")                                                    \
        "s_cmp_lg_u32 s47, 0\n\t"                                                                                       \
        "s_cbranch_scc1 Lloop%=\n"                                                                                      \
        "Lend%=:\n\t"                                                                                                   \
        "s_waitcnt vmcnt(0) lgkmcnt(0)"                                                                                 \
        :                                                                                                               \
        : "s"(sxy), "s"(coef4), "s"(cutz), "s"(s_npass), "s"(s_group), "s"(s_p_begin), "s"(s_p_end),                    \
          "s"(s_lg), "s"(s_p_last), "s"(s_nx8), "s"(s_cbase), "s"(s_nxm2), "s"(s_Li), "s"(s_Uim1),                      \
          "s"(s_dummy), "v"(lane), "v"(ctr_addr), "v"(1), "s"(s_first), "v"(wave + n_waves), "v"(first_cuts), "s"(s_halfb) \
        : "memory", "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50", "s53", "s54", "s55", \
          "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",     \
          "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",                   \
          "v62", "v63")

// PAIRED (lane mapping 8): the band's rows are rows of paired 32-bit cells (dsi_vote_asm.h), row_bytes = 16 * rw apart
template <bool PAIRED = false>
__device__ __forceinline__ void packed_stream_asm_dealt(const EvRec* sxy, const uint4* coef4,
                                                        const uint32_t* cutz, char* band_bytes,
                                                        int p_begin, int p_end, int lg_group, int wave, int n_waves,
                                                        int lane, int nx, int Li, int Ui, int row_base,
                                                        uint32_t dummy_eo, int* pass_counter, uint32_t first_cuts,
                                                        int row_bytes = 0)
{
    // first_cuts: the cut words of the wave's first pass, lane l = packet p_begin + wave * group + l (clamped to
    // p_end - 1), loaded by the caller -- in the fused kernel BEFORE the previous phase's barrier and read-back,
    // which hide the round trip that would otherwise open every phase
    const int group = 1 << lg_group;
    int npass = (p_end - p_begin + group - 1) >> lg_group;
    if (Ui - 1 < Li) npass = 0;  // no acceptable row (the unsigned range test needs Ui-1-Li >= 0)
    if (!PAIRED) row_bytes = nx * 8;
    const int s_npass = __builtin_amdgcn_readfirstlane(npass);
    const int s_group = __builtin_amdgcn_readfirstlane(group);
    const int s_p_begin = __builtin_amdgcn_readfirstlane(p_begin);
    const int s_p_end = __builtin_amdgcn_readfirstlane(p_end);
    const int s_lg = __builtin_amdgcn_readfirstlane(lg_group);
    const int s_p_last = __builtin_amdgcn_readfirstlane(p_end - 1);
    const int s_nx8 = __builtin_amdgcn_readfirstlane(row_bytes);
    const int lds_base = (int)(uintptr_t)band_bytes;  // LDS byte offset of the band
    const int s_cbase = __builtin_amdgcn_readfirstlane(lds_base - row_base * row_bytes);
    const int s_nxm2 = __builtin_amdgcn_readfirstlane(nx - 2);
    const int s_Li = __builtin_amdgcn_readfirstlane(Li);
    const int s_Uim1 = __builtin_amdgcn_readfirstlane(Ui - 1 - Li);
    const uint32_t s_dummy = __builtin_amdgcn_readfirstlane(dummy_eo);
    const int ctr_addr = (int)(uintptr_t)pass_counter;
    const int s_first = __builtin_amdgcn_readfirstlane(wave);
    const int s_halfb = __builtin_amdgcn_readfirstlane(row_bytes >> 1);  // (PAIRED: byte offset of a row's V words)
    if constexpr (PAIRED)
        DSI_DEALT_STREAM_ASM(DSI_ASM_VOTE_PAIRED19);
    else
        DSI_DEALT_STREAM_ASM(DSI_ASM_VOTE);
}

// ---- lane mapping 5, hand-scheduled: the batches of one range (<= 64 batches = 4096 slots) of a
// vector-fill pass (see vfill_stream for the slot -> record mapping; the per-pass tables are built
// by compiled code and handed over through the wave's LDS scratch).  Per batch: 3 v_readlane +
// 2 v_mbcnt + 1 ds_read_b64 of the run table find every lane's record and packet (two ds_bpermute of
// register tables at first: one LDS instruction fewer per batch was worth 7 % at 1024 x 1024 x 256,
// 4.96 -> 4.60 ms), then the same GATHER / VOTE as the packed
// stream.  Two batches of gathers are in flight while one is voted: the table look-up of batch k+2
// is issued BEFORE the votes of batch k (its latency hides behind ~45 vector instructions),
// waited for with lgkmcnt(4) (the four ds_add_u64 of batch k are younger), and the gathers of
// batch k+2 go into the register set batch k has just released.
//   s42 batch counter k   s45 k+2   s46,s49,s50 tail-bit words / tails before the batch   s51 first slot of batch k+2
//   v32 slot per lane   v34 12 * (record - slot)   v35 table address, then the coefficient byte offset
//   sets A / B and the temporaries as in packed_stream_asm
#define DSI_ASM_VPREP1                                                                             \
    "v_readlane_b32 s46, %5, s45\n\t"     /* tail bits of the batch, low / high half */             \
    "v_readlane_b32 s49, %6, s45\n\t"                                                               \
    "v_readlane_b32 s50, %7, s45\n\t"     /* runs that end before the batch */                      \
    "v_mbcnt_lo_u32_b32 v35, s46, 0\n\t"                                                            \
    "v_mbcnt_hi_u32_b32 v35, s49, v35\n\t" /* + runs that end before this lane's slot */            \
    "v_add_u32 v35, v35, s50\n\t"                                                                   \
    "v_lshl_add_u32 v35, v35, 3, %8\n\t"                                                            \
    "ds_read_b64 v[34:35], v35\n\t"       /* the run's {D, coefficient byte offset} */              \
    "s_lshl_b32 s51, s45, 6\n\t"                                                                    \
    "s_add_i32 s51, s51, %3\n\t"          /* first slot of the batch */

#define DSI_ASM_VPREP2(EV, CA, CR)                                                                 \
    "v_add_u32 v32, s51, %15\n\t"         /* slot */                                                \
    "s_mul_i32 s46, s51, 12\n\t"                                                                    \
    "v_cmp_gt_i32 vcc, %4, v32\n\t"       /* slot < T */                                            \
    "v_add3_u32 v58, v34, s46, %17\n\t"   /* byte offset of the record: 12 * (D + slot) */          \
    "v_cndmask_b32 v58, %14, v58, vcc\n\t" /* beyond the pass: the multiplicity-0 record */         \
    "global_load_dwordx3 " EV ", v58, %0\n\t"                                                       \
    "global_load_dwordx4 " CA ", v35, %1\n\t"                                                       \
    "global_load_dword " CR ", v35, %1 offset:16\n\t"

__device__ __forceinline__ void vfill_range_asm(const EvRec* sxy, const uint4* coef4, int nb, int slot0,
                                                int T, uint32_t wlo, uint32_t whi, int cb, const void* tab,
                                                char* band_bytes, int lane, int nx, int Li,
                                                int Ui, int row_base, uint32_t dummy_eo)
{
    const int s_tab = __builtin_amdgcn_readfirstlane((int)(uintptr_t)tab);  // LDS address of the run table
    const int s_nb = __builtin_amdgcn_readfirstlane(nb);
    const int s_slot0 = __builtin_amdgcn_readfirstlane(slot0);
    const int s_T = __builtin_amdgcn_readfirstlane(T);
    const int s_nx8 = __builtin_amdgcn_readfirstlane(nx * 8);
    const int lds_base = (int)(uintptr_t)band_bytes;
    const int s_cbase = __builtin_amdgcn_readfirstlane(lds_base - row_base * nx * 8);
    const int s_nxm2 = __builtin_amdgcn_readfirstlane(nx - 2);
    const int s_Li = __builtin_amdgcn_readfirstlane(Li);
    const int s_Uim1 = __builtin_amdgcn_readfirstlane(Ui - 1 - Li);
    asm volatile(
        // prologue: batches 0 and 1 (batch 1 may lie beyond the range: all its lanes then take the
        // multiplicity-0 record only if it is also beyond the pass; it is never voted)
        "s_mov_b32 s42, 0\n\t"
        "s_mov_b32 s45, 0\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt lgkmcnt(0)\n\t"
        DSI_ASM_VPREP2("v[42:44]", "v[46:49]", "v45")
        "s_cmp_lt_i32 1, %2\n\t"
        "s_cbranch_scc0 Llast1%=\n\t"        // nb == 1
        "s_mov_b32 s45, 1\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt lgkmcnt(0)\n\t"
        DSI_ASM_VPREP2("v[50:52]", "v[54:57]", "v53")
        "Lloop%=:\n\t"
        // ---- batch k in set A
        "s_add_i32 s45, s42, 2\n\t"
        "s_cmp_lt_i32 s45, %2\n\t"
        "s_cbranch_scc0 LtailA%=\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "s_waitcnt lgkmcnt(4)\n\t"
        DSI_ASM_VPREP2("v[42:44]", "v[46:49]", "v45")
        "s_add_i32 s42, s42, 1\n\t"
        // ---- batch k+1 in set B
        "s_add_i32 s45, s42, 2\n\t"
        "s_cmp_lt_i32 s45, %2\n\t"
        "s_cbranch_scc0 LtailB%=\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "s_waitcnt lgkmcnt(4)\n\t"
        DSI_ASM_VPREP2("v[50:52]", "v[54:57]", "v53")
        "s_add_i32 s42, s42, 1\n\t"
        "s_branch Lloop%=\n"
        "LtailA%=:\n\t"                      // batches k (A) and k+1 (B) remain, both in flight
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "s_waitcnt vmcnt(0)\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "s_branch Lend%=\n"
        "LtailB%=:\n\t"                      // batches k (B) and k+1 (A) remain
        "s_waitcnt vmcnt(3)\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "s_waitcnt vmcnt(0)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "s_branch Lend%=\n"
        "Llast1%=:\n\t"                      // a single batch
        "s_waitcnt vmcnt(0)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "Lend%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        :
        : "s"(sxy), "s"(coef4), "s"(s_nb), "s"(s_slot0), "s"(s_T), "v"(wlo), "v"(whi), "v"(cb), "s"(s_tab),
          "s"(s_nx8), "s"(s_cbase), "s"(s_nxm2), "s"(s_Li), "s"(s_Uim1), "v"(dummy_eo * 12u), "v"(lane),
          "v"(0), "v"(lane * 12)
        : "memory", "scc", "vcc", "s42", "s45", "s46", "s49", "s50", "s51", "v32", "v33", "v34", "v35",
          "v36", "v37", "v38", "v39", "v40", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// The same range with THREE register sets (k_vote_bands_vfill: one workgroup per CU, 128 VGPRs): the
// gathers of batch k+3 are issued while batch k is voted, i.e. two whole iterations of other work
// cover a gather's round trip instead of one -- with 16 waves per CU a wave's iteration is ~1300
// clocks and the loaded L2 / fabric takes about that long.  Set C = v[24:26] record, v27 r,
// v[28:31] a,bx,by,d.  The closing votes wait for everything in flight at once (vmcnt(0)) and then
// run in any order, so one drain serves all three phases: s52 = bit mask of the sets still to vote.
__device__ __forceinline__ void vfill_range_asm3(const EvRec* sxy, const uint4* coef4, int nb, int slot0,
                                                 int T, uint32_t wlo, uint32_t whi, int cb, const void* tab,
                                                 char* band_bytes, int lane, int nx, int Li,
                                                 int Ui, int row_base, uint32_t dummy_eo)
{
    const int s_tab = __builtin_amdgcn_readfirstlane((int)(uintptr_t)tab);  // LDS address of the run table
    const int s_nb = __builtin_amdgcn_readfirstlane(nb);
    const int s_slot0 = __builtin_amdgcn_readfirstlane(slot0);
    const int s_T = __builtin_amdgcn_readfirstlane(T);
    const int s_nx8 = __builtin_amdgcn_readfirstlane(nx * 8);
    const int lds_base = (int)(uintptr_t)band_bytes;
    const int s_cbase = __builtin_amdgcn_readfirstlane(lds_base - row_base * nx * 8);
    const int s_nxm2 = __builtin_amdgcn_readfirstlane(nx - 2);
    const int s_Li = __builtin_amdgcn_readfirstlane(Li);
    const int s_Uim1 = __builtin_amdgcn_readfirstlane(Ui - 1 - Li);
    asm volatile(
        // prologue: up to three batches in flight
        "s_mov_b32 s42, 0\n\t"
        "s_mov_b32 s45, 0\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt lgkmcnt(0)\n\t"
        DSI_ASM_VPREP2("v[42:44]", "v[46:49]", "v45")
        "s_mov_b32 s52, 1\n\t"
        "s_cmp_lt_i32 1, %2\n\t"
        "s_cbranch_scc0 Ldrain%=\n\t"
        "s_mov_b32 s45, 1\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt lgkmcnt(0)\n\t"
        DSI_ASM_VPREP2("v[50:52]", "v[54:57]", "v53")
        "s_mov_b32 s52, 3\n\t"
        "s_cmp_lt_i32 2, %2\n\t"
        "s_cbranch_scc0 Ldrain%=\n\t"
        "s_mov_b32 s45, 2\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt lgkmcnt(0)\n\t"
        DSI_ASM_VPREP2("v[24:26]", "v[28:31]", "v27")
        "Lloop%=:\n\t"
        // ---- batch k in set A; B and C in flight behind it
        "s_add_i32 s45, s42, 3\n\t"
        "s_cmp_lt_i32 s45, %2\n\t"
        "s_mov_b32 s52, 7\n\t"
        "s_cbranch_scc0 Ldrain%=\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt vmcnt(6)\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "s_waitcnt lgkmcnt(4)\n\t"
        DSI_ASM_VPREP2("v[42:44]", "v[46:49]", "v45")
        "s_add_i32 s42, s42, 1\n\t"
        // ---- batch k in set B; C and A behind it
        "s_add_i32 s45, s42, 3\n\t"
        "s_cmp_lt_i32 s45, %2\n\t"
        "s_cbranch_scc0 Ldrain%=\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt vmcnt(6)\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "s_waitcnt lgkmcnt(4)\n\t"
        DSI_ASM_VPREP2("v[50:52]", "v[54:57]", "v53")
        "s_add_i32 s42, s42, 1\n\t"
        // ---- batch k in set C; A and B behind it
        "s_add_i32 s45, s42, 3\n\t"
        "s_cmp_lt_i32 s45, %2\n\t"
        "s_cbranch_scc0 Ldrain%=\n\t"
        DSI_ASM_VPREP1
        "s_waitcnt vmcnt(6)\n\t"
        DSI_ASM_VOTE("v24", "v25", "v26", "v28", "v29", "v30", "v31", "v27")
        "s_waitcnt lgkmcnt(4)\n\t"
        DSI_ASM_VPREP2("v[24:26]", "v[28:31]", "v27")
        "s_add_i32 s42, s42, 1\n\t"
        "s_branch Lloop%=\n"
        // ---- everything that is still in flight (1..3 batches: exactly nb - k of them, and when the
        //      loop has run all three sets hold one) is voted after one wait, in any order
        "Ldrain%=:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_bitcmp1_b32 s52, 0\n\t"
        "s_cbranch_scc0 LnoA%=\n\t"
        DSI_ASM_VOTE("v42", "v43", "v44", "v46", "v47", "v48", "v49", "v45")
        "LnoA%=:\n\t"
        "s_bitcmp1_b32 s52, 1\n\t"
        "s_cbranch_scc0 LnoB%=\n\t"
        DSI_ASM_VOTE("v50", "v51", "v52", "v54", "v55", "v56", "v57", "v53")
        "LnoB%=:\n\t"
        "s_bitcmp1_b32 s52, 2\n\t"
        "s_cbranch_scc0 LnoC%=\n\t"
        DSI_ASM_VOTE("v24", "v25", "v26", "v28", "v29", "v30", "v31", "v27")
        "LnoC%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        :
        : "s"(sxy), "s"(coef4), "s"(s_nb), "s"(s_slot0), "s"(s_T), "v"(wlo), "v"(whi), "v"(cb), "s"(s_tab),
          "s"(s_nx8), "s"(s_cbase), "s"(s_nxm2), "s"(s_Li), "s"(s_Uim1), "v"(dummy_eo * 12u), "v"(lane),
          "v"(0), "v"(lane * 12)
        : "memory", "scc", "vcc", "s42", "s45", "s46", "s49", "s50", "s51", "s52", "v24", "v25", "v26", "v27",
          "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v42", "v43",
          "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
          "v59", "v60", "v61", "v62", "v63");
}

// the passes of a wave: per-pass tables by compiled code, the batches by vfill_range_asm
__device__ __forceinline__ void vfill_stream_asm(const EvRec* __restrict__ sxy,
                                                 const uint4* __restrict__ coef4,
                                                 const uint32_t* __restrict__ cutz,
                                                 char* __restrict__ band_bytes,
                                                 unsigned long long* __restrict__ scratch,
                                                 int p_first, int p_end, int lg_pass, int stride,
                                                 int lane, int nx, int Li, int Ui, int row_base,
                                                 uint32_t dummy_eo, int variant, int p_begin_of_wg,
                                                 int* __restrict__ pass_counter, const InlineCuts& ic)
{
    if (Ui - 1 < Li) return;  // the band accepts no row (the unsigned range test needs Ui-1-Li >= 0)
    const int pass = 1 << lg_pass;
    const int p_begin = p_begin_of_wg;
    (void)stride;
    // Passes are DEALT, not pre-assigned, in UNITS of half a pass: wave w starts with units 2w and 2w + 1 and
    // draws every further stretch from a counter in LDS (set to twice the number of waves by the item's set-up).
    // A pass holds 64 packets' records of this band and plane -- anything from a few hundred to a few thousand
    // slots -- and an item has only ~2-10 passes per wave, so with fixed assignments the workgroup waited for its
    // unluckiest wave.  A draw takes two units (a whole pass: the per-pass set-up is amortised over 64 packets)
    // until few are left, then one (guided: the waves finish within half a pass of each other).  (The sums are
    // integer: which wave votes a pass does not change a bit.)
    const int half = pass >> 1;
    const int n_units = (p_end - p_begin + half - 1) / half;
    // single units once fewer than kGuided per wave remain (variant 100, experiments builds only: never, i.e. whole
    // passes to the end as in round 2, for same-box A/B runs)
    const int kGuided = (variant >= 100 && variant <= 108) ? variant - 100 : 4;
    const int n_waves_wg = (int)(blockDim.x >> 6);
    auto draw = [&](int& take) {
        int v = 0, t = 2;
        if (lane == 0) {
            const int seen = __hip_atomic_load(pass_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = (kGuided == 0 || n_units - seen > kGuided * n_waves_wg) ? 2 : 1;
            v = atomicAdd(pass_counter, t);
        }
        take = __builtin_amdgcn_readfirstlane(t);
        return __builtin_amdgcn_readfirstlane(v);
    };
    // lane l = packet l of the stretch that starts at unit u.  (Tried: interleaved passes -- lane l = packet
    // p_begin + l * npass + j, so that the 3-4 packets a batch mixes are far apart in time and a scene point's
    // votes do not meet in one wave instruction: -1 % ... +3 %, not adopted.  Tried: the ds_bpermute look-ups
    // issued a further iteration ahead, waited for with lgkmcnt(6): no change, 5.01 vs 5.00 ms.)
    auto packet_of = [&](int u) { return p_begin + u * half + lane; };
    const int u0 = 2 * ((p_first - p_begin) / pass);  // this wave's first stretch: a whole pass
    uint32_t cu_next = 0;
    // BandPlan::cuts_inline (ic.rsT): no cut table.  The runs of a stretch come in two dependent steps -- the packets'
    // {flags, d_a, by_a}, then two entries of the transposed row table -- so stretches are drawn TWO ahead: while stretch
    // u is voted, step 2 of stretch u' (the next) and step 1 of stretch u'' (the one after) are in flight.
    const bool inl = ic.rsT != nullptr;
    uint4 c1_next = make_uint4(0u, 0u, 0u, 0u);   // step 1 of the stretch after this one
    uint32_t e_lo = 0, e_hi = 0, e_flags = kCoefSkip;  // step 2 of this stretch (arrives during the previous one)
    int u_after = n_units, t_after = 2;
    bool valid_next = false;
    auto lanes_of = [&](int u, int t) { return u < n_units && lane < t * half && packet_of(u) < p_end; };
    auto step2 = [&](uint4 c1, int p, bool valid) {
        e_flags = kCoefSkip;
        e_lo = e_hi = 0;
        if (valid) {
            uint32_t at_lo, at_hi;
            inline_cut_entries(ic, c1, 2u * (uint32_t)p, &at_lo, &at_hi);
            e_flags = c1.y;
            e_lo = inline_cut_load(ic, at_lo);
            e_hi = inline_cut_load(ic, at_hi);
        }
    };
    int take = 2, tn = 2;
    if (inl) {
        const bool v0 = lanes_of(u0, 2);
        uint4 c1 = make_uint4(0u, 0u, 0u, 0u);
        if (v0) c1 = coef4[2 * (size_t)packet_of(u0) + 1];
        u_after = draw(t_after);                     // the second stretch
        valid_next = lanes_of(u_after, t_after);
        if (valid_next) c1_next = coef4[2 * (size_t)packet_of(u_after) + 1];
        step2(c1, packet_of(u0), v0);                // (waits for the first stretch's step 1: once per item and wave)
    } else if (u0 < n_units && lane < pass && packet_of(u0) < p_end) {
        cu_next = cutz[packet_of(u0)];
    }
    for (int u = u0, un; u < n_units; u = un, take = tn) {
        const int p = packet_of(u);
        uint32_t cu;
        (void)take;  // (lanes beyond the stretch loaded no cut word: length 0)
        if (inl) {
            cu = inline_cut_word(e_flags, e_lo, e_hi);   // this stretch's entries (requested one stretch ago)
            un = u_after;
            tn = t_after;
            const uint4 c1 = c1_next;                    // the next stretch's step 1 (requested one stretch ago) ...
            const bool vn = valid_next;
            u_after = draw(t_after);                     // ... the stretch after it is drawn now, its step 1 requested ...
            valid_next = lanes_of(u_after, t_after);
            c1_next = make_uint4(0u, 0u, 0u, 0u);
            if (valid_next) c1_next = coef4[2 * (size_t)packet_of(u_after) + 1];
            step2(c1, packet_of(un), vn);                // ... and the next stretch's entries
        } else {
            cu = cu_next;
            // the next stretch is drawn now and its cut words travel while this one is voted
            un = draw(tn);
            cu_next = 0;
            if (un < n_units && lane < tn * half && packet_of(un) < p_end) cu_next = cutz[packet_of(un)];
        }
        const int lo = (int)(cu & 0xffffu), hi = (int)(cu >> 16);
        const int len = max(hi - lo, 0);
        const int incl = wave_incl_scan(len, lane);
        const int T = __builtin_amdgcn_readlane(incl, 63);
        if (T == 0) continue;
        const int D = (p << 10) + lo - (incl - len);
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(len > 0);
        const int c = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ne >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)ne, 0u));
        const int dst = len > 0 ? c : 63 - (lane - c);  // non-empty runs to lanes 0, 1, ...; the rest behind
        // table entry of compacted run `dst`: {12 * D (byte offsets of 12-byte records, mod 2^32), byte
        // offset of the packet's coefficients}
        const unsigned long long entry =
            (unsigned long long)(uint32_t)(D * 12) | ((unsigned long long)(uint32_t)(min(p, p_end - 1) << 5) << 32);
        int Cbase = 0;
        for (int rbase = 0; rbase < T; rbase += 4096) {
            scratch[lane] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int ts = incl - 1 - rbase;  // last slot of this lane's run, relative to the range
            if (len > 0 && ts >= 0 && ts < 4096) atomicOr(&scratch[ts >> 6], 1ull << (ts & 63));
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const unsigned long long w = scratch[lane];
            // the tail words now live in registers: the same words carry the run table
            // {D, coefficient byte offset} of the compacted runs, one ds_read_b64 per batch instead
            // of two ds_bpermute
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            scratch[dst] = entry;  // (a permutation of the lanes: every entry 0..63 is written)
            // slots beyond the pass count all R <= 64 runs as ended: entry 64 must hold a valid
            // coefficient offset too (their record is the multiplicity-0 dummy)
            if (lane == 0) scratch[64] = entry;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int pc = __builtin_popcountll(w);
            const int inc = wave_incl_scan(pc, lane);
            const int nb = min(64, (T - rbase + 63) >> 6);
            if (variant == 3)
                vfill_range_asm(sxy, coef4, nb, rbase, T, (uint32_t)w, (uint32_t)(w >> 32), inc - pc + Cbase, scratch,
                                band_bytes, lane, nx, Li, Ui, row_base, dummy_eo);
            else
                vfill_range_asm3(sxy, coef4, nb, rbase, T, (uint32_t)w, (uint32_t)(w >> 32), inc - pc + Cbase, scratch,
                                 band_bytes, lane, nx, Li, Ui, row_base, dummy_eo);
            Cbase += __builtin_amdgcn_readlane(inc, 63);
        }
    }
}

// ---- the same hand-scheduled loop over GROUPS of S packets sorted together (lane mapping 4).
// A run is then S times longer (wide grids: ~19 records per packet and band at 1024 x 1024, which
// makes the per-packet stream scalar-bound at 4.4 run pieces per batch).  A record's coefficients
// now depend on the record (its packet index within the group sits in the high half of m), so
// the pipeline is three batches deep: records of batch i+2 and coefficients of batch i+1 are in
// flight while batch i is voted; three register slots, loop unrolled three times.
//   s47,s48,s49 lanes of the batches in slots 0,1,2      v52 record index per lane   v53 cut words
//   slot 0: v[20:22] record  v23 group's coefficient base  v[24:27] a,bx,by,d  v28 r
//   slot 1: v[30:32]         v33                           v[40:43]            v29
//   slot 2: v[44:46]         v47                           v[48:51]            v34
//   temporaries v36-v39, v56-v63 (v57 = multiplicity)
#define DSI_ASM_FILL_G(NOUT, L)                                                                    \
    "s_mov_b32 s44, 0\n"                                                                            \
    "Ltop" L "%=:\n\t"                                                                              \
    "s_cmp_ge_i32 s40, s41\n\t"                                                                     \
    "s_cbranch_scc1 Lnext" L "%=\n\t"                                                               \
    "s_sub_i32 s45, 64, s44\n\t"                                                                    \
    "s_sub_i32 s46, s41, s40\n\t"                                                                   \
    "s_min_i32 s45, s45, s46\n\t"                                                                   \
    "s_sub_i32 s46, s40, s44\n\t"                                                                   \
    "s_lshl_b64 exec, -1, s44\n\t"                                                                  \
    "v_add_u32 v52, s46, %15\n\t"                                                                   \
    "s_mov_b64 exec, -1\n\t"                                                                        \
    "s_add_i32 s44, s44, s45\n\t"                                                                   \
    "s_add_i32 s40, s40, s45\n\t"                                                                   \
    "s_cmp_lt_u32 s44, 64\n\t"                                                                      \
    "s_cbranch_scc0 Ldone" L "%=\n"                                                                 \
    "Lnext" L "%=:\n\t"                                                                             \
    "s_add_i32 s42, s42, 1\n\t"                                                                     \
    "s_cmp_ge_i32 s42, %3\n\t"                                                                      \
    "s_cbranch_scc1 Leos" L "%=\n\t"                                                                \
    "s_and_b32 s45, s42, %4\n\t"                                                                    \
    "s_cmp_eq_u32 s45, 0\n\t"                                                                       \
    "s_cbranch_scc1 Lreload" L "%=\n\t"                                                             \
    "v_readlane_b32 s46, v53, s45\n\t"                                                              \
    "s_add_i32 s43, s43, %16\n"           /* next group: S * 1024 records further */                \
    "Lhave" L "%=:\n\t"                                                                             \
    "s_and_b32 s45, s46, 0xffff\n\t"                                                                \
    "s_lshr_b32 s46, s46, 16\n\t"                                                                   \
    "s_add_i32 s40, s43, s45\n\t"                                                                   \
    "s_add_i32 s41, s43, s46\n\t"                                                                   \
    "s_branch Ltop" L "%=\n"                                                                        \
    "Lreload" L "%=:\n\t"                                                                           \
    "s_lshr_b32 s45, s42, %7\n\t"                                                                   \
    "s_mul_i32 s45, s45, %6\n\t"                                                                    \
    "s_add_i32 s45, s45, %5\n\t"          /* first group of the pass */                             \
    "s_mul_i32 s43, s45, %16\n\t"                                                                   \
    "v_add_u32 v58, s45, %15\n\t"                                                                   \
    "v_min_i32 v58, %8, v58\n\t"                                                                    \
    "v_lshlrev_b32 v58, 2, v58\n\t"                                                                 \
    "global_load_dword v53, v58, %2\n\t"                                                            \
    "s_waitcnt vmcnt(0)\n\t"                                                                        \
    "v_readfirstlane_b32 s46, v53\n\t"                                                              \
    "s_branch Lhave" L "%=\n"                                                                       \
    "Leos" L "%=:\n\t"                                                                              \
    "s_cmp_eq_u32 s44, 0\n\t"                                                                       \
    "s_cbranch_scc1 Ldone" L "%=\n\t"                                                               \
    "s_lshl_b64 exec, -1, s44\n\t"                                                                  \
    "v_mov_b32 v52, %14\n\t"                                                                        \
    "s_mov_b64 exec, -1\n"                                                                          \
    "Ldone" L "%=:\n\t"                                                                             \
    "s_mov_b32 " NOUT ", s44\n\t"

// request the records of the batch just filled; keep the byte offset of its group's coefficients
#define DSI_ASM_GREC(REC, GB)                                                                      \
    "v_mul_lo_u32 v56, v52, 12\n\t"                                                                 \
    "v_lshrrev_b32 " GB ", 10, v52\n\t"   /* packet of the record, if groups were single packets */ \
    "v_and_b32 " GB ", %17, " GB "\n\t"   /* first packet of the group */                           \
    "v_lshlrev_b32 " GB ", 5, " GB "\n\t" /* 32 bytes per coefficient set */                        \
    "global_load_dwordx3 " REC ", v56, %0\n\t"

// the records have arrived: request each lane's coefficients (packet index = high half of m)
#define DSI_ASM_GCOEF(M, GB, CA, CR)                                                               \
    "v_lshrrev_b32 v56, 16, " M "\n\t"                                                              \
    "v_lshl_add_u32 v56, v56, 5, " GB "\n\t"                                                        \
    "global_load_dwordx4 " CA ", v56, %1\n\t"                                                       \
    "global_load_dword " CR ", v56, %1 offset:16\n\t"

#define DSI_ASM_GITER(NJ, RECJ2, GBJ2, NJ2, MJ1, GBJ1, CAJ1, CRJ1, EX, EY, EM, KA, KBX, KBY, KD, KR, L) \
    DSI_ASM_FILL_G(NJ2, L)                                                                         \
    DSI_ASM_GREC(RECJ2, GBJ2)                                                                      \
    "s_waitcnt vmcnt(3)\n\t"              /* records of the next batch are here */                  \
    DSI_ASM_GCOEF(MJ1, GBJ1, CAJ1, CRJ1)                                                           \
    "s_waitcnt vmcnt(3)\n\t"              /* coefficients of this batch are here */                 \
    "s_cmp_eq_u32 " NJ ", 0\n\t"                                                                    \
    "s_cbranch_scc1 Lend%=\n\t"                                                                     \
    "v_and_b32 v57, 0xffff, " EM "\n\t"   /* multiplicity */                                        \
    DSI_ASM_VOTE(EX, EY, "v57", KA, KBX, KBY, KD, KR)

__device__ __forceinline__ void group_stream_asm(const EvRec* sxy, const uint4* coef4,
                                                 const uint32_t* cutz, char* band_bytes,
                                                 int g_first, int g_end, int lg_pass, int stride,
                                                 int lane, int nx, int Li, int Ui, int row_base,
                                                 uint32_t dummy_eo, int S)
{
    const int pass = 1 << lg_pass;
    int n_my = 0;
    if (g_first < g_end) {
        const int passes = (g_end - g_first + stride - 1) / stride;
        const int last = g_first + (passes - 1) * stride;
        n_my = (passes - 1) * pass + min(pass, g_end - last);
    }
    if (Ui - 1 < Li) n_my = 0;  // no acceptable row (the unsigned range test needs Ui-1-Li >= 0)
    const int s_n_my = __builtin_amdgcn_readfirstlane(n_my);
    const int s_gmask = __builtin_amdgcn_readfirstlane(pass - 1);
    const int s_g_first = __builtin_amdgcn_readfirstlane(g_first);
    const int s_stride = __builtin_amdgcn_readfirstlane(stride);
    const int s_lg = __builtin_amdgcn_readfirstlane(lg_pass);
    const int s_g_last = __builtin_amdgcn_readfirstlane(g_end - 1);
    const int s_nx8 = __builtin_amdgcn_readfirstlane(nx * 8);
    const int lds_base = (int)(uintptr_t)band_bytes;
    const int s_cbase = __builtin_amdgcn_readfirstlane(lds_base - row_base * nx * 8);
    const int s_nxm2 = __builtin_amdgcn_readfirstlane(nx - 2);
    const int s_Li = __builtin_amdgcn_readfirstlane(Li);
    // rows the band accepts: 0 <= yi - Li <= Ui - 1 - Li  (a band with no acceptable row gets no stream)
    const int s_Uim1 = __builtin_amdgcn_readfirstlane(Ui - 1 - Li);
    const uint32_t s_dummy = __builtin_amdgcn_readfirstlane(dummy_eo);
    const int s_gs = __builtin_amdgcn_readfirstlane(S * kPacket);
    const int s_pkmask = __builtin_amdgcn_readfirstlane(~(S - 1));
    asm volatile(
        "s_mov_b32 s42, -1\n\t"
        "s_mov_b32 s40, 0\n\t"
        "s_mov_b32 s41, 0\n\t"
        "s_mov_b32 s43, 0\n\t"
        "v_mov_b32 v52, 0\n\t"
        "v_mov_b32 v53, 0\n\t"
        DSI_ASM_FILL_G("s47", "p")
        DSI_ASM_GREC("v[20:22]", "v23")
        DSI_ASM_FILL_G("s48", "q")
        DSI_ASM_GREC("v[30:32]", "v33")
        "s_waitcnt vmcnt(1)\n\t"
        DSI_ASM_GCOEF("v22", "v23", "v[24:27]", "v28")
        "Lloop%=:\n\t"
        DSI_ASM_GITER("s47", "v[44:46]", "v47", "s49", "v32", "v33", "v[40:43]", "v29",
                      "v20", "v21", "v22", "v24", "v25", "v26", "v27", "v28", "a")
        DSI_ASM_GITER("s48", "v[20:22]", "v23", "s47", "v46", "v47", "v[48:51]", "v34",
                      "v30", "v31", "v32", "v40", "v41", "v42", "v43", "v29", "b")
        DSI_ASM_GITER("s49", "v[30:32]", "v33", "s48", "v22", "v23", "v[24:27]", "v28",
                      "v44", "v45", "v46", "v48", "v49", "v50", "v51", "v34", "c")
        "s_branch Lloop%=\n"
        "Lend%=:\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        :
        : "s"(sxy), "s"(coef4), "s"(cutz), "s"(s_n_my), "s"(s_gmask), "s"(s_g_first), "s"(s_stride),
          "s"(s_lg), "s"(s_g_last), "s"(s_nx8), "s"(s_cbase), "s"(s_nxm2), "s"(s_Li), "s"(s_Uim1),
          "s"(s_dummy), "v"(lane), "s"(s_gs), "s"(s_pkmask)
        : "memory", "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32",
          "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
          "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
          "v59", "v60", "v61", "v62", "v63");
}

// The votes of ONE work item (band j, plane z, packets [p_begin, p_end) of one camera) into the band in
// LDS: every wave of the workgroup streams its share of the packets.  LDS row 0 of the band is grid
// row `row_base`; the item accepts events with floor(Y) in [Li, Ui - 1].
// MAPPING = the lane mapping (1 packed / hand-scheduled, 3 packed / compiled, 5 vector fill /
// hand-scheduled, 6 vector fill / compiled, 7 packed / hand-scheduled with dealt passes).  TWO_SETS: the vector fill keeps two instead of three batches of
// gathers in flight (32 instead of 40 named registers; the fused kernel needs the difference).  DEAL: the hand-scheduled
// packed stream draws its passes from *s_pass instead of taking every kWaves-th one.
template <int BLOCK, int MAPPING, bool TWO_SETS = false, bool DEAL = false>
__device__ __forceinline__ void stream_item(const EvRec* __restrict__ sxy, const PlaneCoef* __restrict__ coef,
                                            const uint32_t* __restrict__ cuts, const uint32_t* __restrict__ slow_any,
                                            int np, const Geom& g, const BandPlan& bp, int j, int z, int p_begin,
                                            int p_end, char* __restrict__ band_bytes, int Li, int Ui, int row_base,
                                            int* __restrict__ s_pass, uint32_t first_cuts = 0)
{
    const int nx = g.nx;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    // packets a wave takes per pass:
    constexpr int kWaves = BLOCK / kWave;
    // as many as possible (fewer pass boundaries), but every wave should get >= 3 passes so that the
    // waves of the workgroup finish together (measured: 2 % at 346x260, 6 % at 512x512 against one
    // pass per wave; 512x512x200, 489 packets: passes of 16 / 8 / 4 / 2 packets 0.224 / 0.216 / 0.219 /
    // 0.232 ms)
    int lg_group = 6;
    while (lg_group > 2 && (p_end - p_begin) < ((kWaves * 3) << lg_group)) --lg_group;
    if (bp.pass_lg > 0) lg_group = bp.pass_lg;
    const int group = 1 << lg_group;
    const uint4* __restrict__ coef4 = reinterpret_cast<const uint4*>(coef) + 2 * (size_t)z * np;
    // (bp.cuts_inline: `cuts` is the transposed row table, there is no table row of this item)
    const uint32_t* __restrict__ cutz = bp.cuts_inline ? cuts : cuts + ((size_t)j * g.nz + z) * np;
    // record np * 1024 is a dummy with multiplicity 0 (k_sort_packets) for the lanes a short last
    // batch does not reach; "its" coefficients are whatever follows the plane's table
    const uint32_t dummy_eo = (uint32_t)np * (uint32_t)kPacket;
    if constexpr (MAPPING == 5 || MAPPING == 6) {
        // vector fill (wide grids): passes of up to 64 packets, smaller when the chunk has few
        // packets so that every wave gets >= 2 passes; 64 words of LDS per wave behind the band.
        // 5 = hand-scheduled batches, 6 = all compiled (A/B tests; also the IEEE-divide planes of 5)
        // (the per-pass set-up costs a few LDS round trips: as few, as large passes as keep every
        //  wave busy)
        int lg_pass = 6;
        while (lg_pass > 3 && (p_end - p_begin) < (kWaves << lg_pass)) --lg_pass;
        if (bp.pass_lg > 0) lg_pass = bp.pass_lg;
        const int pass = 1 << lg_pass;
        unsigned long long* scratch =
            reinterpret_cast<unsigned long long*>(band_bytes + bp.scratch_offset) + wave * kVfillScratchWords;
        const InlineCuts ic = inline_cuts_of(cuts, g, bp, j);  // (ic.rsT != nullptr: `cuts` is the transposed row table)
        if (slow_any[z] != 0)
            vfill_stream<true>(sxy, coef4, cutz, band_bytes, scratch, p_begin + wave * pass, p_end, lg_pass,
                               kWaves * pass, lane, nx, Li, Ui, row_base, dummy_eo, ic);
        else if constexpr (MAPPING == 6)
            vfill_stream<false>(sxy, coef4, cutz, band_bytes, scratch, p_begin + wave * pass, p_end, lg_pass,
                                kWaves * pass, lane, nx, Li, Ui, row_base, dummy_eo, ic);
        else
            vfill_stream_asm(sxy, coef4, cutz, band_bytes, scratch, p_begin + wave * pass, p_end, lg_pass,
                             kWaves * pass, lane, nx, Li, Ui, row_base, dummy_eo, TWO_SETS ? 3 : bp.experiment, p_begin, s_pass, ic);
    } else {
        // MAPPING 3 is the compiled stream on the fast path too (A/B testing)
        if (slow_any[z] != 0)
            packed_stream<true, MAPPING == 8>(sxy, coef4, cutz, band_bytes, p_begin + wave * group, p_end, lg_group,
                                              kWaves * group, lane, nx, Li, Ui, row_base, dummy_eo, paired_row_words(nx) * 8);
        else if constexpr (MAPPING == 3)
            packed_stream<false>(sxy, coef4, cutz, band_bytes, p_begin + wave * group, p_end, lg_group,
                                 kWaves * group, lane, nx, Li, Ui, row_base, dummy_eo);
        else if constexpr (MAPPING == 8) {  // mapping 7's stream on paired 32-bit cells (opt-in, NOT the exact sums)
            int lg = (p_end - p_begin) >= kWaves * 16 ? 3 : 2;
            while (lg < 5 && (p_end - p_begin) >= ((kWaves * 16) << (lg + 1))) ++lg;
            if (bp.pass_lg > 0) lg = bp.pass_lg;
            first_cuts = p_end > p_begin ? cutz[min(p_begin + (wave << lg) + lane, p_end - 1)] : 0u;
            packed_stream_asm_dealt<true>(sxy, coef4, cutz, band_bytes, p_begin, p_end, lg, wave, kWaves, lane, nx, Li, Ui, row_base,
                                          dummy_eo, s_pass, first_cuts, paired_row_words(nx) * 8);
        }
        else if constexpr (DEAL || MAPPING == 7) {
            // (*s_pass = 2 * kWaves, set by the item's set-up.)  Packets per pass: enough passes per wave for the dealing
            // to balance the waves (>= ~16), few enough switches (each costs a wait for the wave's outstanding LDS
            // operations): 4 at a 50 ms window's 489 packets (measured 4 / 8 / 16: 446 / 455 / 471 us for the fused kernel),
            // 16 at 4,900 (346x260x100: 1.190 / 1.182 / 1.179 ms for 4 / 8 / 16), 16-32 at 9,800 (512x512x200, 10 M
            // events: 2.76 / 2.62 / 2.55 ms for 4 / 8 / 16)
            // Re-measured at the end of round 3 (passes of 4 / 8 / 16 packets, kernel time): 976 packets 346x260x100
            // 146.5 / 143.6 / 143.4 us; 3,906 packets 497 / 491 / 492 us, at 512x512x200 1.138 / 1.095 / 1.088 ms; a window's
            // 489 packets in the fused kernel 431.9 / 430.6 / 458.5 us (2 packets: 450.9) -- 8 is never worse than 4.
            int lg = (p_end - p_begin) >= kWaves * 16 ? 3 : 2;
            while (lg < 5 && (p_end - p_begin) >= ((kWaves * 16) << (lg + 1))) ++lg;
            if (bp.pass_lg > 0) lg = bp.pass_lg;
            if constexpr (!DEAL)  // nobody loaded the wave's first cut words ahead of time (the fused kernel does)
                first_cuts = p_end > p_begin ? cutz[min(p_begin + (wave << lg) + lane, p_end - 1)] : 0u;
            packed_stream_asm_dealt(sxy, coef4, cutz, band_bytes, p_begin, p_end, lg, wave, kWaves,
                                    lane, nx, Li, Ui, row_base, dummy_eo, s_pass, first_cuts);
        }
        else
            packed_stream_asm(sxy, coef4, cutz, band_bytes, p_begin + wave * group, p_end, lg_group,
                              kWaves * group, lane, nx, Li, Ui, row_base, dummy_eo);
    }
}

// One kernel per mapping, so that each carries only its own
// streams (all of them in one kernel needed 70 VGPRs -- 7 waves per SIMD, i.e. ONE 1024-thread
// workgroup per CU instead of two -- and spilled scalars).  8 waves per SIMD = at most 64 VGPRs.
template <int BLOCK, int MAPPING>
__device__ __forceinline__ void vote_bands_packed_body(const EvRec* __restrict__ sxy,
                                                       const PlaneCoef* __restrict__ coef,
                                                       const uint32_t* __restrict__ cuts,
                                                       const uint32_t* __restrict__ slow_any,
                                                       int np, const Geom& g, const BandPlan& bp,
                                                       void* __restrict__ out,
                                                       acc_t* __restrict__ seam,
                                                       uint32_t* __restrict__ work_counters)
{
    extern __shared__ acc_t band[];
    __shared__ int s_item;
    __shared__ int s_pass;  // mapping 5: the next pass of the item to hand out (vfill_stream_asm)
    const int pairs = bp.chunks * bp.bands;
    const int total = item_count(pairs, g.nz, bp.experiment == 200);
    const int nx = g.nx;
    // PERSISTENT workgroups (work_counters != nullptr): the grid is only as large as the chip holds
    // at once and every workgroup pulls work items until none is left -- a 1024-thread workgroup
    // with up to 160 KB of LDS costs ~5 us to launch and a CU cannot overlap that with the previous
    // workgroup (measured at 512x512x200: 7.7 us of fixed cost per work item, 38 % of a 500 k-event
    // launch).  Items are dealt per XCD class (block % 8, the dispatch rule the block -> (pair, plane)
    // mapping relies on): class x takes items x, x + 8, x + 16, ... in order, through one atomic
    // counter per class, so the load stays balanced like the hardware's own dispatch.
    const int cls = blockIdx.x & 7;
    __shared__ int s_next;  // persistent: the item AFTER the current one, drawn while the current one is voted
    constexpr bool kPaired = MAPPING == 8;
    const int row_words = kPaired ? paired_row_words(nx) : nx;  // 8-byte words per band row
    if (work_counters) {
        const int all_cells = (bp.band_rows + 1) * row_words;
        for (int i = threadIdx.x; i < all_cells; i += BLOCK) band[i] = 0;
        if (threadIdx.x == 0) {
            s_item = (int)atomicAdd(&work_counters[cls], 1u) * 8 + cls;
            s_pass = 2 * (BLOCK / kWave);  // (vector fill: units of half a pass, two pre-assigned per wave)
        }
        __syncthreads();
    }
    for (;;) {
    const int b = work_counters ? s_item : (int)blockIdx.x;
    if (b >= total) break;
    int q, z;
    if (!item_of(b, pairs, g.nz, q, z, bp.experiment == 200)) {  // (a leftover slot beyond the last plane)
        if (!work_counters) break;
        __syncthreads();  // every thread has read s_item before thread 0 draws again
        if (threadIdx.x == 0) s_item = (int)atomicAdd(&work_counters[cls], 1u) * 8 + cls;
        __syncthreads();
        continue;
    }
    // Round 4: the NEXT item is drawn now, at the start of this one.  Only thread 0's wave waits for the atomic's round
    // trip (~1 us); the other 15 waves are already voting, and because the waves of a workgroup draw their passes from a
    // counter, a wave that enters the stream late simply takes fewer passes.  Drawn after the flush (rounds 2-3), the
    // round trip stood between two items with all 16 waves idle: 61 items per CU at 1024x1024x256.
    if (work_counters && threadIdx.x == 0) s_next = (int)atomicAdd(&work_counters[cls], 1u) * 8 + cls;
    const int c = q / bp.bands, j = q % bp.bands;
    const int r0 = j * bp.band_rows;
    const int r1 = min(g.ny, r0 + bp.band_rows);
    const int cells = (r1 - r0 + 1) * row_words;  // owned rows + the carry row
    if (!work_counters) {
        for (int i = threadIdx.x; i < cells; i += BLOCK) band[i] = 0;
        if (threadIdx.x == 0) s_pass = 2 * (BLOCK / kWave);
        __syncthreads();
    }

    const int p_begin = (int)(((long long)np * c) / bp.chunks);
    const int p_end = (int)(((long long)np * (c + 1)) / bp.chunks);
    if (bp.experiment != 1)  // (1: timing experiment, the item's fixed cost only: zero, barriers, flush)
        stream_item<BLOCK, MAPPING>(sxy, coef, cuts, slow_any, np, g, bp, j, z, p_begin, p_end,
                                    reinterpret_cast<char*>(band), r0, min(r1, g.ny - 1), r0, &s_pass);
    __syncthreads();

    const size_t vol = partial_stride((size_t)g.nx * g.ny * g.nz);
    const size_t off = (size_t)c * vol + ((size_t)z * g.ny + r0) * nx;
    if constexpr (kPaired) {
        // (the overflow word sits behind the per-plane flags and the 8 work counters: zeroed by the sort kernel)
        flush_band_paired<BLOCK>(band, nx, r1 - r0, out, off, bp.raw_out, seam_rows(seam, c, z, j, g, bp), j, bp.bands,
                                 const_cast<uint32_t*>(slow_any) + g.nz + 8);
        if (!work_counters) break;
    } else {
    if (!work_counters) {
        flush_band<BLOCK>(band, nx, (r1 - r0) * nx, out, off, bp.raw_out, seam_rows(seam, c, z, j, g, bp), j, bp.bands);
        break;
    }
    if (bp.experiment != 2)  // (2: timing experiment without the flush)
        flush_band_and_clear<BLOCK>(band, nx, (r1 - r0) * nx, out, off, bp.raw_out, seam_rows(seam, c, z, j, g, bp), j, bp.bands);
    }
    if (threadIdx.x == 0) {  // (every thread read s_item before the stream's barrier; s_pass is idle between the barriers)
        s_item = s_next;
        s_pass = 2 * (BLOCK / kWave);
    }
    __syncthreads();  // the band is clear and the next item known before anybody goes on
    }
}

template <int BLOCK, int MAPPING>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_vote_bands_packed(
    const EvRec* __restrict__ sxy, const PlaneCoef* __restrict__ coef, const uint32_t* __restrict__ cuts,
    const uint32_t* __restrict__ slow_any, int np, Geom g, BandPlan bp, void* __restrict__ out,
    acc_t* __restrict__ seam, uint32_t* __restrict__ work_counters)
{
    vote_bands_packed_body<BLOCK, MAPPING>(sxy, coef, cuts, slow_any, np, g, bp, out, seam, work_counters);
}

// The vector-fill mappings run where ONE workgroup fills a CU (wide grids): 4 waves per SIMD may use up
// to 128 VGPRs, which pays for a third register set -- three batches of gathers in flight.  (Run with a
// band small enough for two workgroups per CU, this kernel still places only one.)
template <int BLOCK, int MAPPING>
__global__ __launch_bounds__(BLOCK) void k_vote_bands_vfill(
    const EvRec* __restrict__ sxy, const PlaneCoef* __restrict__ coef, const uint32_t* __restrict__ cuts,
    const uint32_t* __restrict__ slow_any, int np, Geom g, BandPlan bp, void* __restrict__ out,
    acc_t* __restrict__ seam, uint32_t* __restrict__ work_counters)
{
    vote_bands_packed_body<BLOCK, MAPPING>(sxy, coef, cuts, slow_any, np, g, bp, out, seam, work_counters);
}

// the 2-ary camera-fusion ops of Grid3D (cartesian3dgrid.h:111-192), used by the fused kernel below and by
// the Grid3D kernels further down
// Grid3D::harmonicMeanTwoGrids(grid2, n) (cartesian3dgrid.h:130-139); fn = n, fn1 = n - 1
__device__ __forceinline__ float harmonic_mean_n(float a, float g, float fn, float fn1)
{
    const float av = a / fn1;
    const float prod = av * g, sum = av + g;
    return fn * prod / (sum + 0.1f);
}

template <int OP>
__device__ __forceinline__ float fuse_op(float a, float g)
{
    if (OP == 1) return (g < a) ? g : a;  // std::min, cartesian3dgrid.h:115
    if (OP == 2) {                        // :119-127, eps = 1e-1f in the denominator
        const float prod = a * g, sum = a + g;
        return 2.f * prod / (sum + 0.1f);
    }
    if (OP == 3) return __builtin_sqrtf(a * g);  // :154
    if (OP == 4) return 0.5f * (a + g);  // :162, double 0.5 * float sum: exact halving
    if (OP == 5) {                       // :145-146, pow() and 0.5 in double
        const double ad = (double)a, gd = (double)g;
        const float ms = (float)(0.5 * (ad * ad + gd * gd));
        return __builtin_sqrtf(ms);  // == (float)sqrt((double)ms): 53 >= 2*24+2
    }
    return (a < g) ? g : a;  // std::max, :188
}

// (3d) FUSED: vote -> camera fusion -> arg-max over Z without the DSIs ever leaving the CU.
//
// What "evaluateDSI x n; fused = op(dsi_0, dsi_1); collapseMaxZSlice(fused)" (process1.cpp:76-166 + :222
// -> mapper_emvs_stereo.cpp:368) computes, for callers that keep only the depth map (the 50 ms window
// loop of main.cpp:177-302): at 512 x 512 x 200 a window writes 2 x 210 MB of camera DSIs only for the
// arg-max to read them straight back.  Here a persistent workgroup owns a contiguous range of
// (band, plane) pairs; for each pair it votes camera 0's events into the band in LDS, reads the
// owned rows back as fp32 into REGISTERS (one value per thread and 1024-voxel stretch; <= 20 of them),
// clears the band, votes camera 1, applies the 2-ary op per voxel (fuse_op, the function k_fuse2_into
// and k_collapse_max_z_fused use: same bits) and keeps a running (maximum, first index) per pixel in
// registers.  When its range leaves a band (and at the end) the running maxima go to one 64-bit key
// per pixel -- confidence bits << 8 | 255 - plane, the key of the plane-sharded arg-max -- with a
// global atomic MAX, which is collapseMaxZSlice's first-maximum-wins over the plane ranges of all
// workgroups; k_unpack_argmax turns the keys into confidence / index / depth.
//
// A band cannot wait for the band above to hand over its carry row here, so the band keeps a HALO row
// on either side of its owned rows [r0, r1) (LDS row 0 = grid row r0 - 1) and processes the events
// with floor(Y) in [r0 - 1, r1 - 1]: the events of the two seam rows are voted by both neighbours
// (1 / band_rows more votes), the halo rows collect the halves that belong to the neighbours and are
// dropped.  The owned rows then hold the exact 64-bit sums, which is what k_seam_rows reconstructs
// for the unfused path: both paths round the same integers once, so their depth maps are bit-equal.
template <int CELLS>
struct FusedBest {
    float best[CELLS];
    uint32_t idx4[(CELLS + 3) / 4];  // plane indices, four per register
};

// Read the band's owned rows back (one value per thread and 1024-cell stretch), zero the band, and either keep
// the fp32 values (camera 0 of 2) or fuse them with camera 0's and update the running arg-max.  Written without
// control flow per cell -- out-of-range lanes read the last owned cell and "zero" a halo cell that is zeroed
// anyway -- so that the CELLS / 2 LDS reads of a half are in flight together: with a branch per cell every
// read was waited for on its own, and the read-back took 3 us per phase (tools/fused_trace.py).
// MODE: what a phase does with the band it has just voted
enum { FUSED_LAST4 = 6,  // camera 3 of 4: sqrt(sqrt(c0 c1) * sqrt(c2 c3)) -- the GM tree's root (cartesian3dgrid.h:150-156), arg-max
       FUSED_READ1 = 5,  // camera 1 of 2, round 6: read back, clear and convert only -- values kept in vb; the fusion op and the
                         // running arg-max (the VALU-bound two thirds of that read-back) are DEFERRED to fused_deferred_argmax,
                         // which every wave runs at the end of ITS OWN next voting stream, while the other waves still vote
       FUSED_KEEP = 0,   // camera 0 of several: keep the values
       FUSED_MID = 1,    // camera 1 of 3: the two-camera op, result kept (process1.cpp:126-158)
       FUSED_LAST2 = 2,  // camera 1 of 2: the two-camera op, then the running arg-max
       FUSED_LAST1 = 3,  // the only camera: arg-max of its own values
       FUSED_LAST3 = 4 };// camera 2 of 3: the third-camera op (process1.cpp:169-191: 1 min, 2 HM with n = 3, 6 max), arg-max
// the deferred half of camera 1's read-back (two cameras): process1.cpp:126-158 on the values of both cameras, then the
// running first maximum (cartesian3dgrid.cpp:132-134) -- registers only, no LDS, no barrier
template <int CELLS, int OP>
__device__ __forceinline__ void fused_deferred_argmax(const float* __restrict__ va, const float* __restrict__ vb,
                                                      FusedBest<CELLS>& fb, int z)
{
#pragma unroll
    for (int kk = 0; kk < CELLS; ++kk) {
        const float f = fuse_op<OP>(0.f + va[kk], vb[kk]);
        const bool better = fb.best[kk] < f;
        fb.best[kk] = better ? f : fb.best[kk];
        const int sh = (kk & 3) * 8;
        const uint32_t with_z = (fb.idx4[kk >> 2] & ~(0xffu << sh)) | ((uint32_t)z << sh);
        fb.idx4[kk >> 2] = better ? with_z : fb.idx4[kk >> 2];
    }
}

template <int CELLS, int OP, int MODE>
__device__ __forceinline__ void fused_consume(acc_t* __restrict__ band, int nx, int n_own, int rows_lds,
                                              float* __restrict__ va, FusedBest<CELLS>& fb, int z,
                                              const float* __restrict__ vpair = nullptr /* FUSED_LAST4: cameras 2's values */)
{
    acc_t* own = band + nx;
    // (the thread index is re-read behind an opaque barrier so that the compiler recomputes the 20 cell
    //  addresses per call instead of keeping them -- 60 registers of loop invariants -- alive across
    //  the voting loops, which name 40 physical registers themselves)
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    // cells read back together.  With camera 1's values parked in a second register array (DEFER) the kernel has ~20
    // registers fewer for the reads in flight: quarters instead of halves there
    constexpr int HALF = (MODE == FUSED_READ1 || (MODE == FUSED_KEEP && CELLS % 4 == 0 && CELLS == 20)) ? CELLS / 4 : CELLS / 2;
    static_assert(CELLS % 2 == 0 && CELLS % HALF == 0, "whole parts");
#pragma unroll
    for (int h = 0; h < CELLS; h += HALF) {
        acc_t raw[HALF];
#pragma unroll
        for (int k = 0; k < HALF; ++k) raw[k] = own[min(t + (h + k) * 1024, n_own - 1)];
        uint32_t hi_or = 0;
#pragma unroll
        for (int k = 0; k < HALF; ++k) hi_or |= (uint32_t)(raw[k] >> 32);
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
            const int i = t + (h + k) * 1024;
            acc_t* cell = i < n_own ? own + i : band;  // band[0]: first halo row, cleared below
            *cell = 0;
        }
        // sums of 2^52 and more (2 M votes in one voxel) take the general conversion: decided per wave
        const bool big = __builtin_amdgcn_ballot_w64((hi_or >> 20) != 0u) != 0ull;
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
            const float v = big ? (float)((double)raw[k] * kFixInv)
                                : (float)(__longlong_as_double((long long)(raw[k] | 0x4140000000000000ull)) - 2097152.0);
            const int kk = h + k;
            if (MODE == FUSED_KEEP || MODE == FUSED_READ1) {
                va[kk] = v;
            } else if (MODE == FUSED_MID) {
                va[kk] = fuse_op<OP>(0.f + va[kk], v);
            } else {
                // process1.cpp:126-158: fused = 0; fused += dsi0; fused.<op>TwoGrids(dsi1)
                // process1.cpp:169-191: fused.minTwoGrids / harmonicMeanTwoGrids(dsi2, 3) / maxTwoGrids(dsi2)
                // (FUSED_LAST4: gm_tree<4> of k_fuse_gm_tree -- t0 = va = sqrt(c0 c1), t1 = sqrt(c2 c3), sqrt(t0 t1))
                const float f = MODE == FUSED_LAST2   ? fuse_op<OP>(0.f + va[kk], v)
                                : MODE == FUSED_LAST3 ? (OP == 2 ? harmonic_mean_n(va[kk], v, 3.f, 2.f) : fuse_op<OP>(va[kk], v))
                                : MODE == FUSED_LAST4 ? fuse_op<3>(va[kk], fuse_op<3>(vpair[kk], v))
                                                      : v;
                const bool better = fb.best[kk] < f;  // strict: the first maximum wins (cartesian3dgrid.cpp:132-134)
                fb.best[kk] = better ? f : fb.best[kk];
                const int sh = (kk & 3) * 8;
                const uint32_t with_z = (fb.idx4[kk >> 2] & ~(0xffu << sh)) | ((uint32_t)z << sh);
                fb.idx4[kk >> 2] = better ? with_z : fb.idx4[kk >> 2];
            }
        }
    }
    acc_t* last = band + (size_t)(rows_lds - 1) * nx;
    for (int i = threadIdx.x; i < nx; i += 1024) {
        band[i] = 0;
        last[i] = 0;
    }
}

// DEFER (round 6, the one-workgroup-per-CU kernel with the packed stream): with two cameras, camera 1's read-back only
// reads, clears and converts (FUSED_READ1); its fusion op and arg-max update -- ~60 % of that read-back's vector
// instructions, the IEEE division of the harmonic mean among them -- are run by every wave at the end of its NEXT voting
// stream, i.e. in the time it would otherwise wait at the phase barrier for the slowest wave (the waves of a phase end
// up to one pass = 2-4 us apart), while the others still vote.  Same values, same order per cell: same bits.
// FOUR (round 6): four cameras fused by the balanced tree of the reference's 2-ary geometric mean (DSI_ACC_GM_TREE,
// cartesian3dgrid.h:150-156 applied pairwise: BASELINE configs[4]) -- camera 0 kept, camera 1 folded into it, camera 2 kept in
// a second register array, camera 3 closes both pairs and the root; same bits as k_collapse_max_z_gm_tree<4> on the four DSIs.
template <int MAPPING, int CELLS, bool DEFER = false, bool FOUR = false, bool DEAL = true>
__device__ __forceinline__ void vote_fuse_argmax_body(const FusedCameras& cams, const Geom& g, const BandPlan& bp, int op,
                                                      const uint32_t* __restrict__ splits,
                                                      unsigned long long* __restrict__ keys,
                                                      unsigned long long* __restrict__ trace)
{
    constexpr int BLOCK = 1024;
    extern __shared__ acc_t band[];
    __shared__ int s_pass;  // the next pass of the item to hand out
    // (the vector fill pre-assigns two units = one pass per wave, the dealt packed stream two passes)
    constexpr int kPass0 = 2 * (BLOCK / kWave);
    const int nx = g.nx;
    // Workgroup b runs on XCD b % 8: each XCD gets one contiguous eighth of the (band-major) pair list,
    // so that a band's records stream through at most two XCDs' L2s, and splits it evenly over its
    // workgroups -- or by `splits` (gridDim.x + 1 pair indices in XCD-major workgroup order) when
    // k_fused_splits has balanced the partition by the records each pair holds.
    const int P = bp.bands * g.nz;
    int q_begin, q_end, q_step = 1;
    // DEALT (bp.interleave == 2, round 6): as "in turn", but only a workgroup's FIRST pair is fixed; every further pair is drawn
    // from its XCD's counter (one global atomic per pair, by thread 0, while the other waves finish camera 0's stream), and a
    // workgroup whose XCD has run dry draws from the next XCD's counter.  The pairs a workgroup meets still ascend within a
    // stretch (few band changes), the workgroups of an XCD still share one band's records at a time, and the kernel no
    // longer lasts as long as the workgroup whose FIXED share was the slowest (span / mean busy time 1.11 with fixed
    // shares: the dense middle bands are slow per record).  Which workgroup votes a pair changes no bit.
    // Counters: 8 words behind the keys, zero at launch (k_unpack_argmax re-zeroes them with the keys).
    // (DEAL = false: the two-workgroups-per-CU kernel, at 64 registers, takes its pairs in turn instead)
    const bool dealt = DEAL && !splits && bp.interleave == 2;
    __shared__ int s_next_q[2];
    __shared__ int s_deal_tries;
    if (splits) {
        // rank = position of this workgroup in XCD-major order: XCD x still covers one contiguous stretch
        const int rank = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        q_begin = (int)splits[rank];
        q_end = (int)splits[rank + 1];
    } else {
        const int x = blockIdx.x & 7, l = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int lo = (int)(((long long)P * x) / 8), hi = (int)(((long long)P * (x + 1)) / 8);
        if (dealt) {
            q_begin = lo + l;
            q_end = hi;
        } else if (bp.interleave) {
            // the XCD's workgroups take the pairs of its stretch in turn: 32 workgroups on 32 consecutive planes of ONE band at a
            // time, so that the band's records (all cameras') stay in that XCD's L2 while they are read once per plane --
            // with contiguous pieces 32 workgroups work on ~9 bands at once at 1024 x 1024 x 256 (four cameras x 375 KB each:
            // 13 MB against 4 MB of L2, 4.6 TB/s from the Infinity Cache).  A workgroup then meets a band change every
            // nz / 32 pairs (one atomicMax per owned pixel each: microseconds).
            q_begin = lo + l;
            q_end = hi;
            q_step = per;
        } else {
        q_begin = lo + (int)(((long long)(hi - lo) * l) / per);
        q_end = lo + (int)(((long long)(hi - lo) * (l + 1)) / per);
        }
    }
    if (q_begin >= q_end) return;
    {
        const int all_cells = (bp.band_rows + 2) * nx;
        for (int i = threadIdx.x; i < all_cells; i += BLOCK) band[i] = 0;
        if (threadIdx.x == 0) {
            s_pass = kPass0;
            s_deal_tries = 0;
        }
    }
    __syncthreads();
    if (dealt) q_end = P;  // (a drawn pair may lie in another XCD's stretch; "none" is beyond every pair)

    // The cameras' table is read where it lies, in the kernel-argument segment, with scalar loads at the (uniform)
    // camera index: indexing the by-value array dynamically made the compiler keep all three cameras' pointers in
    // registers next to the ~40 the voting loops name (or copy the table to scratch).  `cams` is the FIRST argument,
    // so cam[0] starts the segment.
    // (layout lock: dsi_kernels.h, next to FusedCameras)
    typedef const FusedCamera __attribute__((address_space(4))) * KernargCameras;
    const KernargCameras kcam = (KernargCameras)__builtin_amdgcn_kernarg_segment_ptr();
    auto camera = [&](int c) -> FusedCamera {
        FusedCamera o;
        o.sxy = kcam[c].sxy;
        o.coef = kcam[c].coef;
        o.cuts = kcam[c].cuts;
        o.slow_any = kcam[c].slow_any;
        o.np = kcam[c].np;
        return o;
    };
    float va[CELLS];
    static_assert(!(DEFER && FOUR), "one use of the second register array");
    float vb[(DEFER || FOUR) ? CELLS : 1];  // DEFER: camera 1's values of the pair whose arg-max update is pending; FOUR: camera 2's
    int pend_z = -1;              // its plane (wave-uniform); -1: nothing pending
    FusedBest<CELLS> fb;
    auto run_pending = [&]() {
        if (!DEFER || pend_z < 0) return;
        switch (op) {
        case 1: fused_deferred_argmax<CELLS, 1>(va, vb, fb, pend_z); break;
        case 2: fused_deferred_argmax<CELLS, 2>(va, vb, fb, pend_z); break;
        case 3: fused_deferred_argmax<CELLS, 3>(va, vb, fb, pend_z); break;
        case 4: fused_deferred_argmax<CELLS, 4>(va, vb, fb, pend_z); break;
        case 5: fused_deferred_argmax<CELLS, 5>(va, vb, fb, pend_z); break;
        default: fused_deferred_argmax<CELLS, 6>(va, vb, fb, pend_z); break;
        }
        pend_z = -1;
    };
    int cur_j = -1, r0 = 0, r1 = 0, n_own = 0;
    // the cut words of this wave's first pass of phase (pair q, camera c) -- see packed_stream_asm_dealt
    constexpr bool kPrefetchCuts = MAPPING == 1;
    auto first_cuts_of = [&](int q, int c) -> uint32_t {
        if (!kPrefetchCuts) return 0u;
        const FusedCamera cam = camera(c);
        if (cam.np <= 0) return 0u;
        const int j = q / g.nz, z = q - j * g.nz;
        int lg = cam.np >= (BLOCK / kWave) * 16 ? 3 : 2;  // (the rule of stream_item's dealt stream)
        while (lg < 5 && cam.np >= (((BLOCK / kWave) * 16) << (lg + 1))) ++lg;
        if (bp.pass_lg > 0) lg = bp.pass_lg;
        const int p = min((int)((threadIdx.x / kWave) << lg) + (int)(threadIdx.x & 63), cam.np - 1);
        return cam.cuts[((size_t)j * g.nz + z) * cam.np + p];
    };
    uint32_t cuts_next = first_cuts_of(q_begin, 0);
    auto emit = [&]() {
        // one key per owned pixel (the owned rows are contiguous in the image)
        unsigned long long* kp = keys + (size_t)r0 * nx;
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));  // see fused_consume
#pragma unroll
        for (int k = 0; k < CELLS; ++k) {
            const int i = t + k * 1024;
            if (i < n_own) {
                const uint32_t zi = (fb.idx4[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                const unsigned long long key = ((unsigned long long)__float_as_uint(fb.best[k]) << 8) | (255u - zi);
                atomicMax(kp + i, key);
            }
        }
    };
    int it = 0;  // pairs this workgroup has begun
    for (int q = q_begin; q < q_end; ++it) {
        const int j = q / g.nz, z = q - j * g.nz;
        int q_next = q + q_step;  // (DEALT: read from s_next_q behind the last phase's first barrier)
        if (j != cur_j) {
            run_pending();  // (the last pair of the band that ends here)
            if (cur_j >= 0) emit();
            cur_j = j;
            r0 = j * bp.band_rows;
            r1 = min(g.ny, r0 + bp.band_rows);
            n_own = (r1 - r0) * nx;
#pragma unroll
            for (int k = 0; k < CELLS; ++k) fb.best[k] = -1.f;  // below every DSI value (>= 0)
#pragma unroll
            for (int k = 0; k < (CELLS + 3) / 4; ++k) fb.idx4[k] = 0u;
        }
        const int rows_lds = r1 - r0 + 2;
        // events with floor(Y) in [r0 - 1, r1 - 1] (and in [0, ny - 2], cartesian3dgrid.h:255-259)
        const int Li = max(r0 - 1, 0), Ui = min(r1, g.ny - 1);
        // (the DEFER instantiation is launched for two cameras only: the other camera counts' read-backs are not in it)
        const int n_cams = DEFER ? 2 : (FOUR ? 4 : cams.n);
#pragma nounroll
        for (int c = 0; c < n_cams; ++c) {
            const FusedCamera cam = camera(c);
#ifdef DSI_TIMING_EXPERIMENTS
            // development aid (experiments flavour only, dsi_test_fused_trace_*): 100 MHz time stamps per (workgroup,
            // phase, wave): stream begins, stream ends, after the barrier + the read-back / clear, after the closing barrier
            int tr = -1;  // (wave-uniform: lives in a scalar register)
            if (trace) {
                const int phase = it * cams.n + c;
                if (phase < kFusedTracePhases)
                    tr = __builtin_amdgcn_readfirstlane((((int)blockIdx.x * kFusedTracePhases + phase) * (BLOCK / kWave) + (int)(threadIdx.x / kWave)) * 4);
            }
#define DSI_FUSED_STAMP(k) do { if (tr >= 0 && (threadIdx.x & 63) == 0) trace[tr + (k)] = wall_clock64(); } while (0)
#else
#define DSI_FUSED_STAMP(k) do { } while (0)
#endif
            DSI_FUSED_STAMP(0);
            const uint32_t cuts_now = cuts_next;
            stream_item<BLOCK, MAPPING, true, true>(cam.sxy, cam.coef, cam.cuts, cam.slow_any, cam.np, g, bp, j, z, 0, cam.np,
                                        reinterpret_cast<char*>(band), Li, Ui, r0 - 1, &s_pass, cuts_now);
            // the next phase's first cut words travel during this phase's barrier and read-back
            const bool last = c == n_cams - 1;
            if (c + 1 < n_cams)
                cuts_next = first_cuts_of(q, c + 1);
            else if (!dealt && q + q_step < q_end)
                cuts_next = first_cuts_of(q + q_step, 0);
            run_pending();  // the previous pair's fusion + arg-max update, while the other waves finish their passes
            if (dealt && c == 0 && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
                // the next pair of this workgroup: its XCD's counter first, then the other XCDs' in turn (the first `per`
                // pairs of every stretch are the workgroups' fixed first pairs).  Wave 0, scalar code but for the atomic.
                // (P < 2^24: at most 2^16 rows -- 16-bit row tables -- and 256 planes -- 8-bit plane indices in the keys)
                const int xcd = (int)(blockIdx.x & 7), per = (int)(gridDim.x >> 3);
                unsigned* const ctr = reinterpret_cast<unsigned*>(keys + (size_t)g.nx * g.ny);
                int tries = __builtin_amdgcn_readfirstlane(s_deal_tries);  // XCDs this workgroup has found dry
                int nq = 0x7fffffff;
                while (tries < 8) {
                    const int xx = (xcd + tries) & 7;
                    unsigned got = 0u;
                    if ((threadIdx.x & 63) == 0) got = atomicAdd(&ctr[xx], 1u);
                    const int cand = (P * xx) / 8 + per + (int)__builtin_amdgcn_readfirstlane(got);
                    if (cand < (P * (xx + 1)) / 8) {
                        nq = cand;
                        break;
                    }
                    ++tries;
                }
                if ((threadIdx.x & 63) == 0) {
                    s_deal_tries = tries;
                    s_next_q[it & 1] = nq;
                }
            }
            DSI_FUSED_STAMP(1);
            __syncthreads();
            if (threadIdx.x == 0) s_pass = kPass0;
            if (dealt && last) {
                // (written before the first barrier of this pair's camera-0 phase; slot it & 1 is next written two pairs on)
                q_next = __builtin_amdgcn_readfirstlane(s_next_q[it & 1]);
                if (q_next < P) cuts_next = first_cuts_of(q_next, 0);
            }
            if constexpr (FOUR) {
                if (c == 0)
                    fused_consume<CELLS, 3, FUSED_KEEP>(band, nx, n_own, rows_lds, va, fb, z);
                else if (c == 1)
                    fused_consume<CELLS, 3, FUSED_MID>(band, nx, n_own, rows_lds, va, fb, z);
                else if (c == 2)
                    fused_consume<CELLS, 3, FUSED_KEEP>(band, nx, n_own, rows_lds, vb, fb, z);
                else
                    fused_consume<CELLS, 3, FUSED_LAST4>(band, nx, n_own, rows_lds, va, fb, z, vb);
            } else if constexpr (DEFER) {
                if (!last) {
                    fused_consume<CELLS, 1, FUSED_KEEP>(band, nx, n_own, rows_lds, va, fb, z);
                } else {
                    fused_consume<CELLS, 1, FUSED_READ1>(band, nx, n_own, rows_lds, vb, fb, z);
                    pend_z = z;
                }
            } else if (!last) {
                if (c == 0) {
                    fused_consume<CELLS, 1, FUSED_KEEP>(band, nx, n_own, rows_lds, va, fb, z);
                } else {  // camera 1 of 3: only the ops whose third step exists get here (1, 2, 6)
                    switch (op) {
                    case 1: fused_consume<CELLS, 1, FUSED_MID>(band, nx, n_own, rows_lds, va, fb, z); break;
                    case 2: fused_consume<CELLS, 2, FUSED_MID>(band, nx, n_own, rows_lds, va, fb, z); break;
                    default: fused_consume<CELLS, 6, FUSED_MID>(band, nx, n_own, rows_lds, va, fb, z); break;
                    }
                }
            } else if (cams.n == 1) {
                fused_consume<CELLS, 1, FUSED_LAST1>(band, nx, n_own, rows_lds, va, fb, z);
            } else if (cams.n == 2) {
                switch (op) {
                case 1: fused_consume<CELLS, 1, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                case 2: fused_consume<CELLS, 2, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                case 3: fused_consume<CELLS, 3, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                case 4: fused_consume<CELLS, 4, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                case 5: fused_consume<CELLS, 5, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                default: fused_consume<CELLS, 6, FUSED_LAST2>(band, nx, n_own, rows_lds, va, fb, z); break;
                }
            } else {
                switch (op) {
                case 1: fused_consume<CELLS, 1, FUSED_LAST3>(band, nx, n_own, rows_lds, va, fb, z); break;
                case 2: fused_consume<CELLS, 2, FUSED_LAST3>(band, nx, n_own, rows_lds, va, fb, z); break;
                default: fused_consume<CELLS, 6, FUSED_LAST3>(band, nx, n_own, rows_lds, va, fb, z); break;
                }
            }
            DSI_FUSED_STAMP(2);
            __syncthreads();
            DSI_FUSED_STAMP(3);
        }
        q = q_next;
    }
    run_pending();
    emit();
#undef DSI_FUSED_STAMP
}

// one workgroup per CU: the band takes (almost) the whole LDS, up to 128 VGPRs
template <int MAPPING, int CELLS, bool DEFER = false, bool FOUR = false>
__global__ __launch_bounds__(1024) void k_vote_fuse_argmax(FusedCameras cams, Geom g, BandPlan bp, int op,
                                                           const uint32_t* __restrict__ splits,
                                                           unsigned long long* __restrict__ keys,
                                                           unsigned long long* __restrict__ trace)
{
    vote_fuse_argmax_body<MAPPING, CELLS, DEFER, FOUR>(cams, g, bp, op, splits, keys, trace);
}

// TWO workgroups per CU (round 4): bands of at most half the LDS, half the cells per thread, <= 64 VGPRs -- while one
// workgroup's 16 waves stand at a phase barrier or read a band back, the other's vote.  Same body, same bits.
// (`cams` stays the FIRST parameter: the body reads the camera table from the start of the kernel-argument segment.)
template <int MAPPING, int CELLS>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_vote_fuse_argmax_2cu(
    FusedCameras cams, Geom g, BandPlan bp, int op, const uint32_t* __restrict__ splits,
    unsigned long long* __restrict__ keys, unsigned long long* __restrict__ trace)
{
    vote_fuse_argmax_body<MAPPING, CELLS, false, false, false>(cams, g, bp, op, splits, keys, trace);
}

// Balanced partition of the (band-major) pair list for the fused kernel: pair q costs work0[q] + work1[q]
// records (counted by k_plane_coef from the runs the voting loop will walk) plus a fixed cost per pair (the
// set-up, barriers and read-back of its phases, in record units); the n_wg + 1 split points cut the list
// into stretches of equal cost.  One block; P <= 1024 * kSplitSlice pairs.
constexpr int kSplitSlice = 64;

__global__ __launch_bounds__(1024) void k_fused_splits(const uint32_t* __restrict__ work0, const uint32_t* __restrict__ work1,
                                                       int P, uint32_t fixed_per_pair, int n_wg,
                                                       unsigned long long* __restrict__ prefix /* [P] scratch */,
                                                       uint32_t* __restrict__ splits)
{
    __shared__ unsigned long long wave_tot[16];
    const int per = (P + 1023) / 1024;
    const int b0 = min(P, (int)threadIdx.x * per), b1 = min(P, b0 + per);
    unsigned long long local = 0;
    for (int i = b0; i < b1; ++i) local += (unsigned long long)work0[i] + (work1 ? work1[i] : 0u) + fixed_per_pair;
    unsigned long long incl = local;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long base = incl - local, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += wave_tot[w];
        total += wave_tot[w];
    }
    for (int i = b0; i < b1; ++i) {
        base += (unsigned long long)work0[i] + (work1 ? work1[i] : 0u) + fixed_per_pair;
        prefix[i] = base;  // cost of pairs 0 .. i
    }
    __syncthreads();
    for (int r = threadIdx.x; r <= n_wg; r += 1024) {
        // first pair index whose inclusive prefix reaches r / n_wg of the total: that many pairs lie before the cut
        const unsigned long long target = (total * (unsigned long long)r + (unsigned long long)n_wg - 1) / (unsigned long long)n_wg;
        int lo = 0, hi = P;  // smallest i in [0, P] with (i == P or prefix[i] >= target) -> pairs [0, i] ... cut after i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (prefix[mid] >= target) hi = mid; else lo = mid + 1;
        }
        // pairs before the r-th cut: those whose inclusive prefix is <= target (so that the last cut takes all)
        int cut = (r == 0) ? 0 : (r == n_wg ? P : min(P, lo + 1));
        splits[r] = (uint32_t)cut;
    }
}

// (3c) GROUPED mapping: S consecutive packets (a "group"; their poses are microseconds apart)
//     are sorted TOGETHER by z0 row, every event keeping its packet's index within the group.
//     The events a band needs from the whole group are then one long contiguous run (the
//     union of the S packets' row ranges), so a wave walks a few hundred events per run with
//     a plain loop -- almost no scalar bookkeeping per batch (the CU's one-per-clock scalar
//     unit is what limits the packed mapping) -- and fetches each lane's coefficients with a
//     gather that stays inside a 32*S-byte window of the plane-major table.
__global__ __launch_bounds__(256) void k_sort_groups(const float2* __restrict__ xy, int np, int S,
                                                     int ny, int nz, int pad, EvRec* __restrict__ sxy,
                                                     uint8_t* __restrict__ spk,
                                                     uint32_t* __restrict__ nvalid,
                                                     uint16_t* __restrict__ rowstart)
{
    extern __shared__ uint32_t hist[];  // nb + 1 counters -> exclusive offsets -> running cursors
    __shared__ uint32_t wave_tot[4];
    const int nb = ny + 2 * pad + 2;
    const int gidx = blockIdx.x;
    const int p0 = gidx * S;
    const int n_ev = min(S, np - p0) * kPacket;
    const float2* __restrict__ src = xy + (size_t)p0 * kPacket;
    if (gidx == 0) {
        for (int i = threadIdx.x; i < nz + 8; i += 256) nvalid[np + i] = 0;  // see k_sort_packets
        if (threadIdx.x == 0) {
            const EvRec none = {0.f, 0.f, 0u};
            sxy[(size_t)np * kPacket] = none;  // the multiplicity-0 record
        }
    }
    for (int i = threadIdx.x; i <= nb; i += 256) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_ev; i += 256) {
        const float2 e = src[i];
        if (finitef(e.x) && finitef(e.y)) atomicAdd(&hist[row_bin(e.y, ny, pad)], 1u);
    }
    __syncthreads();
    const int per = (nb + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    uint32_t local = 0;
    for (int i = b0; i < b1; ++i) local += hist[i];
    uint32_t incl = local;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - local;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
    const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    for (int i = b0; i < b1; ++i) {
        const uint32_t c = hist[i];
        hist[i] = base;
        base += c;
    }
    if (threadIdx.x == 0) hist[nb] = total;
    __syncthreads();
    uint16_t* rs = rowstart + (size_t)gidx * (nb + 1);
    for (int i = threadIdx.x; i <= nb; i += 256) rs[i] = (uint16_t)hist[i];  // S*1024 <= 32768
    __syncthreads();
    EvRec* __restrict__ dst = sxy + (size_t)p0 * kPacket;
    uint8_t* __restrict__ dpk = spk + (size_t)p0 * kPacket;
    for (int i = threadIdx.x; i < n_ev; i += 256) {
        const float2 e = src[i];
        if (finitef(e.x) && finitef(e.y)) {
            const uint32_t pos = atomicAdd(&hist[row_bin(e.y, ny, pad)], 1u);
            // multiplicity 1 in the low half, packet index within the group in the high half (the
            // hand-scheduled loop finds the record's coefficients through it)
            const EvRec r = {e.x, e.y, 1u | ((uint32_t)(i >> 10) << 16)};
            dst[pos] = r;
            dpk[pos] = (uint8_t)(i >> 10);  // packet index within the group
        }
    }
    if (threadIdx.x == 0) nvalid[gidx] = total;
}

// union of the row-bin ranges of a group's packets -> the run [lo, hi) of the group
__global__ __launch_bounds__(256) void k_group_cuts(const uint32_t* __restrict__ prow,
                                                    const uint16_t* __restrict__ rowstart, int np,
                                                    int ngroups, int S, int nz, int bands, int nb,
                                                    uint32_t* __restrict__ gcuts)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)ngroups * nz * bands;
    if (tid >= total) return;
    const int gidx = (int)(tid % ngroups);  // group fastest: coalesced writes
    const size_t jz = tid / ngroups;        // = band * nz + z
    const int p0 = gidx * S, p1 = min(np, p0 + S);
    uint32_t blo = 0xffffu, bhi = 0;
    const uint32_t* __restrict__ row = prow + jz * np;
    for (int p = p0; p < p1; ++p) {
        const uint32_t w = row[p];
        const uint32_t l = w & 0xffffu, h = w >> 16;
        if (l <= h && !(l == 0xffffu)) {
            blo = min(blo, l);
            bhi = max(bhi, h);
        }
    }
    uint32_t lo = 0, hi = 0;
    if (blo <= bhi && blo != 0xffffu) {
        const uint16_t* rs = rowstart + (size_t)gidx * (nb + 1);
        lo = rs[blo];
        hi = rs[bhi];
    }
    gcuts[jz * ngroups + gidx] = lo | (hi << 16);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_vote_groups(const EvRec* __restrict__ sxy,
                                                       const uint8_t* __restrict__ spk,
                                                       const PlaneCoef* __restrict__ coef,
                                                       const uint32_t* __restrict__ gcuts,
                                                       const uint32_t* __restrict__ slow_any, int np,
                                                       int ngroups, int S, Geom g, BandPlan bp,
                                                       void* __restrict__ out,
                                                       acc_t* __restrict__ seam)
{
    extern __shared__ acc_t band[];
    const int b = blockIdx.x;
    const int pairs = bp.chunks * bp.bands;
    int q, z;
    if (!item_of(b, pairs, g.nz, q, z, bp.experiment == 200)) return;
    const int c = q / bp.bands, j = q % bp.bands;
    const int r0 = j * bp.band_rows;
    const int r1 = min(g.ny, r0 + bp.band_rows);
    const int nx = g.nx;
    const int cells = (r1 - r0 + 1) * nx;  // owned rows + the carry row
    for (int i = threadIdx.x; i < cells; i += BLOCK) band[i] = 0;
    __syncthreads();

    const int g_begin = (int)(((long long)ngroups * c) / bp.chunks);
    const int g_end = (int)(((long long)ngroups * (c + 1)) / bp.chunks);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int Li = r0, Ui = min(r1, g.ny - 1);
    const int row_base = r0;
    const uint4* __restrict__ coef4 = reinterpret_cast<const uint4*>(coef) + 2 * (size_t)z * np;
    const uint32_t* __restrict__ cutz = gcuts + ((size_t)j * g.nz + z) * ngroups;

    if (bp.packed == 4 && slow_any[z] == 0) {
        // lane mapping 4: the hand-scheduled stream over groups (runs of all the wave's groups
        // packed back to back into the lanes); planes that need the IEEE divide take the loop below
        constexpr int kWaves = BLOCK / kWave;
        int lg_pass = 6;
        while (lg_pass > 0 && (g_end - g_begin) < ((kWaves * 4) << lg_pass)) --lg_pass;
        if (bp.pass_lg > 0) lg_pass = bp.pass_lg;
        const int pass = 1 << lg_pass;
        group_stream_asm(sxy, coef4, cutz, reinterpret_cast<char*>(band), g_begin + wave * pass, g_end,
                         lg_pass, kWaves * pass, lane, nx, Li, Ui, row_base,
                         (uint32_t)np * (uint32_t)kPacket, S);
    } else
    for (int gi = g_begin + wave; gi < g_end; gi += BLOCK / kWave) {
        const uint32_t cu = cutz[gi];
        const int lo = (int)(cu & 0xffffu), hi = (int)(cu >> 16);
        if (lo >= hi) continue;
        const size_t ev0 = (size_t)gi * S * kPacket;
        const EvRec* __restrict__ ev = sxy + ev0;
        const uint8_t* __restrict__ pk = spk + ev0;
        const uint4* __restrict__ cg = coef4 + 2 * (size_t)gi * S;  // the group's coefficient window
        // Two batches per trip with two register sets (A, B): the gathers of batch b+1 (event +
        // coefficients, which need that batch's packet indices) and the packet indices of
        // batch b+2 are in flight while batch b is voted.  All loads are straight-line with
        // indices clamped into the run, so the compiler counts outstanding loads exactly and no
        // register copy waits for a load.
        auto vote_one = [&](EvRec e, uint4 va, uint2 vb, bool act) {
            const float ka = __uint_as_float(va.x), kbx = __uint_as_float(va.y);
            const float kby = __uint_as_float(va.z), kd = __uint_as_float(va.w);
            const float kr = __uint_as_float(vb.x);
            const float nxv = e.x * ka + kbx;  // mapper_emvs_stereo.cpp:194-195
            const float nyv = e.y * ka + kby;
            float X, Y, nmax;
            if (__builtin_amdgcn_ballot_w64((vb.y & kCoefSlow) != 0) != 0) {  // rare, whole wave
                X = nxv / kd;
                Y = nyv / kd;
                nmax = __builtin_inff();
            } else {
                X = div_rc(nxv, kd, kr);
                Y = div_rc(nyv, kd, kr);
                nmax = 1e30f;
            }
            // cartesian3dgrid.h:255-259 restricted to this band's rows as one integer test
            // (see k_vote_bands_packed); dead packets of the group carry kCoefSkip
            const float xf = __builtin_floorf(X), yf = __builtin_floorf(Y);
            const int xi = (int)xf, yi = (int)yf;
            int sgn = xi | (nx - 2 - xi) | (yi - Li) | (Ui - 1 - yi);
            sgn |= (act && !(vb.y & kCoefSkip) && fabsf(nxv) < nmax && fabsf(nyv) < nmax) ? 0 : -1;
            if (sgn >= 0) {
                const int idx = __mul24(yi - row_base, nx) + xi;
                vote4(band, idx, nx, X - xf, Y - yf, e.m & 0xffffu);  // cartesian3dgrid.h:261-270
            }
        };
        const int last = hi - 1;
        int i0 = lo + lane, i1 = i0 + kWave;
        int k0 = pk[min(i0, last)], k1 = pk[min(i1, last)];
        EvRec eA = ev[min(i0, last)], eB;
        uint4 vaA = cg[2 * k0], vaB;
        uint2 vbA = *reinterpret_cast<const uint2*>(cg + 2 * k0 + 1), vbB;
        for (int base = lo; base < hi; base += 2 * kWave) {
            // set B <- batch b+1, packet indices of b+2; vote batch b (set A)
            eB = ev[min(i1, last)];
            vaB = cg[2 * k1];
            vbB = *reinterpret_cast<const uint2*>(cg + 2 * k1 + 1);
            k0 = pk[min(i0 + 2 * kWave, last)];
            vote_one(eA, vaA, vbA, i0 < hi);
            // set A <- batch b+2, packet indices of b+3; vote batch b+1 (set B)
            eA = ev[min(i0 + 2 * kWave, last)];
            vaA = cg[2 * k0];
            vbA = *reinterpret_cast<const uint2*>(cg + 2 * k0 + 1);
            k1 = pk[min(i1 + 2 * kWave, last)];
            vote_one(eB, vaB, vbB, i1 < hi);
            i0 += 2 * kWave;
            i1 += 2 * kWave;
        }
    }
    __syncthreads();
    const size_t vol = partial_stride((size_t)g.nx * g.ny * g.nz);
    const size_t off = (size_t)c * vol + ((size_t)z * g.ny + r0) * nx;
    flush_band<BLOCK>(band, nx, (r1 - r0) * nx, out, off, bp.raw_out, seam_rows(seam, c, z, j, g, bp), j, bp.bands);
}

// (4) seam rows: the first row of band j >= 1 of every plane of every chunk's volume is the exact
// fixed-point sum of the band's own votes (head) and of the band above's votes into the row below
// it (carry), rounded to fp32 ONCE: out[c][z][j * band_rows][x] = fl((head + carry) * 2^-31).
// Runs after the voting kernel and before the chunk volumes are summed.
// block x = one (plane, seam) row, block y = a 512-voxel stretch of it, block z = chunk; two voxels
// per thread (16-byte loads, 8-byte stores) when nx is even
template <bool RAW>
__global__ __launch_bounds__(256) void k_seam_rows(const acc_t* __restrict__ seam, Geom g, int bands, int band_rows,
                                                   void* __restrict__ out, size_t vol_stride)
{
    const int j = (int)(blockIdx.x % (unsigned)(bands - 1)) + 1;
    const int z = (int)(blockIdx.x / (unsigned)(bands - 1));
    const int c = (int)blockIdx.z;
    const size_t row = ((size_t)c * g.nz + z) * bands;
    const acc_t* head = seam + ((row + j) * 2) * g.nx;
    const acc_t* carry = seam + ((row + j - 1) * 2 + 1) * g.nx;
    const size_t off = (size_t)c * vol_stride + ((size_t)z * g.ny + (size_t)j * band_rows) * g.nx;
    float* dstf = reinterpret_cast<float*>(out) + off;
    acc_t* dstr = reinterpret_cast<acc_t*>(out) + off;
    if ((g.nx & 1) == 0) {
        const int x = ((int)blockIdx.y * 256 + (int)threadIdx.x) * 2;
        if (x >= g.nx) return;
        const ulonglong2 h = *reinterpret_cast<const ulonglong2*>(head + x);
        const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(carry + x);
        if (RAW) {
            *reinterpret_cast<ulonglong2*>(dstr + x) = make_ulonglong2(h.x + k.x, h.y + k.y);
        } else {
            float2 r;
            r.x = fix_to_float(h.x + k.x);  // one rounding to f32
            r.y = fix_to_float(h.y + k.y);
            *reinterpret_cast<float2*>(dstf + x) = r;
        }
    } else {
        for (int x = (int)blockIdx.y * 512 + (int)threadIdx.x; x < min(g.nx, ((int)blockIdx.y + 1) * 512); x += 256) {
            if (RAW)
                dstr[x] = head[x] + carry[x];
            else
                dstf[x] = fix_to_float(head[x] + carry[x]);
        }
    }
}

// (5) DSI = fl(sum over the chunks of their raw 64-bit partial volumes * 2^-31) [+ the grid's previous
// contents: fillVoxelGrid accumulates, mapper_emvs_stereo.cpp:151-205 has no reset]: the integer sum is
// exact and order-free, so the DSI does not depend on the number of chunks.  Two voxels per thread.
__global__ __launch_bounds__(256) void k_reduce_partials(const acc_t* __restrict__ partials,
                                                         int chunks, size_t n,
                                                         float* __restrict__ dsi, int accumulate,
                                                         const acc_t* __restrict__ seam, int nx, int ny, int nz, int bands,
                                                         int band_rows)
{
    // seam != nullptr: the seam rows (first row of every band but the first) are taken from the bands' head and
    // carry sums instead of the partial volumes (k_seam_rows folded in: one launch less per evaluateDSI);
    // requires an even nx (two voxels per thread never straddle a row)
    const size_t n2 = n / 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t vs = partial_stride(n);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        ulonglong2 acc = make_ulonglong2(0ull, 0ull);
        bool from_seam = false;
        if (seam) {
            const unsigned v = (unsigned)(2 * i);          // voxel index < 2^32 (checked by the launcher)
            const unsigned rowi = v / (unsigned)nx;        // z * ny + y
            const unsigned y = rowi % (unsigned)ny;
            const unsigned j = y / (unsigned)band_rows;
            if (j >= 1 && y == j * (unsigned)band_rows) {
                from_seam = true;
                const unsigned z = rowi / (unsigned)ny, x = v - rowi * (unsigned)nx;
                for (int c = 0; c < chunks; ++c) {
                    const size_t row = ((size_t)c * nz + z) * bands;
                    const ulonglong2 h = *reinterpret_cast<const ulonglong2*>(seam + ((row + j) * 2) * nx + x);
                    const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(seam + ((row + j - 1) * 2 + 1) * nx + x);
                    acc.x += h.x + k.x;
                    acc.y += h.y + k.y;
                }
            }
        }
        if (!from_seam)
            for (int c = 0; c < chunks; ++c) {
                const ulonglong2 v = reinterpret_cast<const ulonglong2*>(partials + (size_t)c * vs)[i];
                acc.x += v.x;
                acc.y += v.y;
            }
        float2 r = make_float2(fix_to_float(acc.x), fix_to_float(acc.y));
        float2* d2 = reinterpret_cast<float2*>(dsi) + i;  // (grids are 16-byte aligned, also wrapped ones)
        if (accumulate) {
            const float2 old = *d2;
            r.x = old.x + r.x;
            r.y = old.y + r.y;
        }
        *d2 = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        const size_t i = n - 1;
        acc_t acc = 0;
        for (int c = 0; c < chunks; ++c) acc += partials[(size_t)c * vs + i];
        const float r = fix_to_float(acc);
        dsi[i] = accumulate ? dsi[i] + r : r;
    }
}

// ------------------------------------------------------------ Grid3D ops ---
template <int OP>
__global__ __launch_bounds__(256) void k_fuse2(float* __restrict__ a, const float* __restrict__ g,
                                               size_t n)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 va = reinterpret_cast<float4*>(a)[i];
        const float4 vg = reinterpret_cast<const float4*>(g)[i];
        va.x = fuse_op<OP>(va.x, vg.x);
        va.y = fuse_op<OP>(va.y, vg.y);
        va.z = fuse_op<OP>(va.z, vg.z);
        va.w = fuse_op<OP>(va.w, vg.w);
        reinterpret_cast<float4*>(a)[i] = va;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        a[i] = fuse_op<OP>(a[i], g[i]);
    }
}

// dst = op(a, g): what "dst.resetGrid(); dst.addTwoGrids(a); dst.<op>TwoGrids(g)"
// (process1.cpp:126-158) leaves in dst, in one pass (0 + a == a exactly for a >= 0)
template <int OP>
__global__ __launch_bounds__(256) void k_fuse2_into(float* __restrict__ dst,
                                                    const float* __restrict__ a,
                                                    const float* __restrict__ g, size_t n)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 va = reinterpret_cast<const float4*>(a)[i];
        const float4 vg = reinterpret_cast<const float4*>(g)[i];
        float4 r;
        r.x = fuse_op<OP>(0.f + va.x, vg.x);
        r.y = fuse_op<OP>(0.f + va.y, vg.y);
        r.z = fuse_op<OP>(0.f + va.z, vg.z);
        r.w = fuse_op<OP>(0.f + va.w, vg.w);
        reinterpret_cast<float4*>(dst)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        dst[i] = fuse_op<OP>(0.f + a[i], g[i]);
    }
}

// generic element-wise driver for the remaining ops
enum { EW_HM_N = 0, EW_ADD = 1, EW_ADD_INV = 2, EW_FIN_AM = 3, EW_FIN_HM = 4,
       EW_ADD_LOG = 5, EW_ADD_SQ = 6, EW_MIN = 7, EW_MAX = 8, EW_FIN_GM = 9, EW_FIN_RMS = 10 };

// ---- n-ary geometric mean: exp(mean(log v)), 0 as soon as one factor is 0 (SURVEY 8d cfg 5).
// The reference only has the 2-ary sqrt(a*g) (cartesian3dgrid.h:150-156).  log and exp are
// spelled out in IEEE double operations (+, *, /, floor, bit moves, explicit fused multiply-adds;
// no libm, nothing fused implicitly: -ffp-contract=off) so that the CPU oracle, which repeats
// exactly this sequence, gets the same bits.
//
// log v = e ln2 + log c_i + log1p(r):  v = 2^e m, m in [1, 2); the top 7 mantissa bits pick
// c_i = 1 + (2 i + 1) / 256 (the centre of m's 1/128 interval), r = (m - c_i) / c_i with |r| <= 2^-8,
// log1p(r) by its degree-5 Taylor polynomial (next term r^6 / 6 < 6e-16).  {1 / c_i, log c_i} come
// from a 128-entry table in LDS that every block fills for itself with the division-based series
// below (full double accuracy; 128 threads x ~60 operations per block), so the per-voxel path has
// no division: ~25 instructions per log instead of ~60 (4-camera fusion at 1024 x 1024 x 256:
// 2.09 -> 1.37 ms, still bound by fp64 issue rather than by the 5.4 GB it streams).
// The absolute error of the log is ~1e-15; its fp32 rounding and the fp32 sum of logs
// dominate (the accumulator is a float grid).
__device__ __forceinline__ double det_log_series(double m)  // m in [1, 2); table set-up only
{
    double e = 0.0;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = 1.0;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;  // <= 0.0295
    double p = 0.047619047619047616;         // 1/21
    p = p * z + 0.05263157894736842;         // 1/19
    p = p * z + 0.058823529411764705;        // 1/17
    p = p * z + 0.06666666666666667;         // 1/15
    p = p * z + 0.07692307692307693;         // 1/13
    p = p * z + 0.09090909090909091;         // 1/11
    p = p * z + 0.1111111111111111;          // 1/9
    p = p * z + 0.14285714285714285;         // 1/7
    p = p * z + 0.2;                         // 1/5
    p = p * z + 0.3333333333333333;          // 1/3
    p = p * z + 1.0;
    const double t1 = e * 0.6931471805599453;
    const double t2 = 2.0 * s;
    return t1 + t2 * p;
}

constexpr int kLogTabSize = 128;

__device__ __forceinline__ void det_log_table_fill(double2* tab)  // whole block; ends with a barrier
{
    for (int i = threadIdx.x; i < kLogTabSize; i += blockDim.x) {
        const double c = 1.0 + (double)(2 * i + 1) * 0.00390625;  // exact
        tab[i] = make_double2(1.0 / c, det_log_series(c));
    }
    __syncthreads();
}

__device__ __forceinline__ float det_logf(float v, const double2* __restrict__ tab)
{
    // (selects instead of early returns: the special cases are rare and a wave would take both
    //  sides anyway)
    const bool positive = v > 0.f, regular = positive && finitef(v);
    const double x = (double)(regular ? v : 1.f);  // every positive float is a normal double
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int e = (int)((b >> 52) & 0x7ffull) - 1023;
    const unsigned i = (unsigned)(b >> 45) & 127u;
    const double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double c = __longlong_as_double((long long)(0x3ff0000000000000ull | ((unsigned long long)(2u * i + 1u) << 44)));
    const double2 t = tab[i];
    const double r = (m - c) * t.x;          // m - c is exact
    double p = __builtin_fma(r, 0.2, -0.25);  // fused multiply-adds, spelled out (IEEE 754 fma)
    p = __builtin_fma(p, r, 0.3333333333333333);
    p = __builtin_fma(p, r, -0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    const float res = (float)__builtin_fma((double)e, 0.6931471805599453, t.y + p);
    const float special = positive ? v /* +inf */ : (v == 0.f ? -__builtin_inff() : __builtin_nanf(""));
    return regular ? res : special;
}

// exp(y) rounded to float: y = k ln2 + r, |r| <= 0.347, degree-9 Taylor polynomial (next term
// r^10 / 10! < 7e-12 relative, far below the float rounding)
__device__ __forceinline__ float det_expf(double y)
{
    const bool regular = y >= -104.0 && y <= 89.0;  // false for NaN
    const float special = (y != y) ? __builtin_nanf("") : (y > 89.0 ? __builtin_inff() : 0.f);  // (-inf: a factor was 0)
    y = regular ? y : 0.0;
    const double kd = __builtin_floor(y * 1.4426950408889634 + 0.5);
    double r = __builtin_fma(kd, -0.6931471803691238, y);      // ln2 split as in fdlibm (hi part has 32 bits)
    r = __builtin_fma(kd, -1.9082149292705877e-10, r);
    double p = 2.7557319223985893e-06;           // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 0.0001984126984126984);
    p = __builtin_fma(p, r, 0.001388888888888889);
    p = __builtin_fma(p, r, 0.008333333333333333);
    p = __builtin_fma(p, r, 0.041666666666666664);
    p = __builtin_fma(p, r, 0.16666666666666666);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const long long k = (long long)kd;           // |k| <= 151: 2^k is a normal double
    const double sc = __longlong_as_double((k + 1023) << 52);
    return regular ? (float)(p * sc) : special;
}

template <int KIND>
__device__ __forceinline__ float ew_op(float a, float g, float fn, float fn1, const double2* __restrict__ log_tab = nullptr)
{
    if (KIND == EW_HM_N) return harmonic_mean_n(a, g, fn, fn1);  // cartesian3dgrid.h:130-139
    if (KIND == EW_ADD) return a + g;                       // :68
    if (KIND == EW_ADD_INV) return a + 1.0f / (0.01f + g);  // :76, eps = 1e-2f
    if (KIND == EW_FIN_AM) return a / fn;                   // :91
    if (KIND == EW_FIN_HM) return fn / a;                   // :84
    // n-ary extensions of the 2-ary camera-fusion ops (cartesian3dgrid.h:111-190) in
    // accumulate / finalize form; sum-, min- and max-reducible across GPUs
    if (KIND == EW_ADD_LOG) return a + det_logf(g, log_tab);  // GM: sum of log v (-inf once a v is 0)
    if (KIND == EW_ADD_SQ) return a + g * g;                // RMS: sum of v^2
    if (KIND == EW_MIN) return (g < a) ? g : a;             // std::min, :115
    if (KIND == EW_MAX) return (a < g) ? g : a;             // std::max, :188
    if (KIND == EW_FIN_GM) return det_expf((double)a * (1.0 / (double)fn));  // (the reciprocal is loop-invariant)
    // EW_FIN_RMS: mean square in double rounded to float, then sqrt like rmsTwoGrids (:145-146)
    const float ms = (float)((double)a / (double)fn);
    return __builtin_sqrtf(ms);
}

template <int KIND>
__global__ __launch_bounds__(256) void k_elementwise(float* __restrict__ a,
                                                     const float* __restrict__ g, size_t n,
                                                     float fn, float fn1)
{
    constexpr bool has_g = (KIND == EW_HM_N || KIND == EW_ADD || KIND == EW_ADD_INV ||
                            KIND == EW_ADD_LOG || KIND == EW_ADD_SQ || KIND == EW_MIN || KIND == EW_MAX);
    __shared__ double2 log_tab[KIND == EW_ADD_LOG ? kLogTabSize : 1];
    if (KIND == EW_ADD_LOG) det_log_table_fill(log_tab);
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 va = reinterpret_cast<float4*>(a)[i];
        float4 vg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_g) vg = reinterpret_cast<const float4*>(g)[i];
        va.x = ew_op<KIND>(va.x, vg.x, fn, fn1, log_tab);
        va.y = ew_op<KIND>(va.y, vg.y, fn, fn1, log_tab);
        va.z = ew_op<KIND>(va.z, vg.z, fn, fn1, log_tab);
        va.w = ew_op<KIND>(va.w, vg.w, fn, fn1, log_tab);
        reinterpret_cast<float4*>(a)[i] = va;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        a[i] = ew_op<KIND>(a[i], has_g ? g[i] : 0.f, fn, fn1, log_tab);
    }
}

// n-ary fusion in ONE pass: dst = finalize(accumulate(... accumulate(identity, src[0]) ..., src[n-1]))
// with exactly the operations, in exactly the order, of dsi_grid_accumulate_begin / n x
// dsi_grid_accumulate / dsi_grid_finalize (so the bits are the same), reading every source once:
// (n + 1) * 4 B per voxel instead of (3 n + 2) * 4 B.
constexpr int kMaxFuseSources = 8;
struct FuseSources {
    const float* p[kMaxFuseSources];
};

template <int ACC, int FIN>
__global__ __launch_bounds__(256) void k_fuse_n(float* __restrict__ dst, FuseSources src, int n_src,
                                                size_t n, float identity, float fn)
{
    __shared__ double2 log_tab[ACC == EW_ADD_LOG ? kLogTabSize : 1];
    if (ACC == EW_ADD_LOG) det_log_table_fill(log_tab);
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 acc = make_float4(identity, identity, identity, identity);
        float4 v[kMaxFuseSources];  // all loads of a voxel group in flight before the arithmetic
#pragma unroll
        for (int c = 0; c < kMaxFuseSources; ++c)
            if (c < n_src) v[c] = reinterpret_cast<const float4*>(src.p[c])[i];
#pragma unroll
        for (int c = 0; c < kMaxFuseSources; ++c)
            if (c < n_src) {
                acc.x = ew_op<ACC>(acc.x, v[c].x, 0.f, 0.f, log_tab);
                acc.y = ew_op<ACC>(acc.y, v[c].y, 0.f, 0.f, log_tab);
                acc.z = ew_op<ACC>(acc.z, v[c].z, 0.f, 0.f, log_tab);
                acc.w = ew_op<ACC>(acc.w, v[c].w, 0.f, 0.f, log_tab);
            }
        if (FIN >= 0) {
            acc.x = ew_op<(FIN >= 0 ? FIN : 0)>(acc.x, 0.f, fn, 0.f);
            acc.y = ew_op<(FIN >= 0 ? FIN : 0)>(acc.y, 0.f, fn, 0.f);
            acc.z = ew_op<(FIN >= 0 ? FIN : 0)>(acc.z, 0.f, fn, 0.f);
            acc.w = ew_op<(FIN >= 0 ? FIN : 0)>(acc.w, 0.f, fn, 0.f);
        }
        reinterpret_cast<float4*>(dst)[i] = acc;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float acc = identity;
        for (int c = 0; c < n_src; ++c) acc = ew_op<ACC>(acc, src.p[c][i], 0.f, 0.f, log_tab);
        if (FIN >= 0) acc = ew_op<(FIN >= 0 ? FIN : 0)>(acc, 0.f, fn, 0.f);
        dst[i] = acc;
    }
}

// ---- n-ary geometric mean as the balanced TREE of the reference's own 2-ary op (DSI_ACC_GM_TREE):
//   n = 2: sqrt(a b)                     = Grid3D::geometricMeanTwoGrids, cartesian3dgrid.h:150-156, bit for bit
//   n = 4: sqrt(sqrt(a b) sqrt(c d))     n = 8: one level more
// i.e. what a user of the reference gets by calling geometricMeanTwoGrids on pairs of grids and then on the
// results.  ~2 fp32 operations per source and voxel: a pure HBM stream, where exp(mean(log v)) of
// DSI_ACC_LOG_SUM (any n, sum-reducible across GPUs) costs ~100 fp64 operations per voxel.
template <int N>
__device__ __forceinline__ float gm_tree(const float* v)
{
    float t[N];
#pragma unroll
    for (int c = 0; c < N; ++c) t[c] = v[c];
#pragma unroll
    for (int w = N; w > 1; w >>= 1)
#pragma unroll
        for (int c = 0; c < w / 2; ++c) t[c] = fuse_op<3>(t[2 * c], t[2 * c + 1]);
    return t[0];
}

template <int N>
__global__ __launch_bounds__(256) void k_fuse_gm_tree(float* __restrict__ dst, FuseSources src, size_t n)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v[N];
#pragma unroll
        for (int c = 0; c < N; ++c) v[c] = reinterpret_cast<const float4*>(src.p[c])[i];
        float x[N], y[N], z[N], w[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            x[c] = v[c].x;
            y[c] = v[c].y;
            z[c] = v[c].z;
            w[c] = v[c].w;
        }
        reinterpret_cast<float4*>(dst)[i] = make_float4(gm_tree<N>(x), gm_tree<N>(y), gm_tree<N>(z), gm_tree<N>(w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float x[N];
#pragma unroll
        for (int c = 0; c < N; ++c) x[c] = src.p[c][i];
        dst[i] = gm_tree<N>(x);
    }
}

// collapseMaxZSlice of the tree geometric mean without materialising it (same bits as k_fuse_gm_tree + k_collapse_max_z)
template <int N>
__global__ __launch_bounds__(256) void k_collapse_max_z_gm_tree(FuseSources src, int npix, int nz,
                                                                float* __restrict__ conf, uint8_t* __restrict__ idx,
                                                                const float* __restrict__ planes, float* __restrict__ depth)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float best = 0.f;
    int best_k = 0;
    for (int k = 0; k < nz; k += 2) {
        float v[2][N];  // two planes' loads in flight before the arithmetic
        const int k1 = min(k + 1, nz - 1);
#pragma unroll
        for (int c = 0; c < N; ++c) {
            v[0][c] = src.p[c][(size_t)k * npix + p];
            v[1][c] = src.p[c][(size_t)k1 * npix + p];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (k + u >= nz) break;
            const float f = gm_tree<N>(v[u]);
            if (k + u == 0 || best < f) {  // std::max_element: the first maximum wins
                best = f;
                best_k = k + u;
            }
        }
    }
    conf[p] = best;
    idx[p] = (uint8_t)best_k;
    if (depth) depth[p] = planes[best_k];
}

// cartesian3dgrid.cpp:115-137: thread = pixel, planes walked in order, strict '<'
// keeps the first maximum (std::max_element).  Lanes of a wave read consecutive x
// of one plane row: coalesced 256-B segments.
__global__ __launch_bounds__(256) void k_collapse_max_z(const float* __restrict__ dsi, int npix,
                                                        int nz, float* __restrict__ conf,
                                                        uint8_t* __restrict__ idx,
                                                        const float* __restrict__ planes,
                                                        float* __restrict__ depth)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float* col = dsi + p;
    float best = col[0];
    int best_k = 0;
    int k = 1;
    for (; k + 8 <= nz; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(k + u) * npix];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (best < v[u]) {
                best = v[u];
                best_k = k + u;
            }
    }
    for (; k < nz; ++k) {
        const float v = col[(size_t)k * npix];
        if (best < v) {
            best = v;
            best_k = k;
        }
    }
    conf[p] = best;
    idx[p] = (uint8_t)best_k;
    if (depth) depth[p] = planes[best_k];  // mapper_emvs_stereo.cpp:302-313
}

// collapseMaxZSlice of op(a, b) without materialising the fused volume: what
// "fused.resetGrid(); fused.addTwoGrids(a); fused.<op>TwoGrids(b); fused.collapseMaxZSlice()"
// (process1.cpp:126-158 + mapper_emvs_stereo.cpp:368) yields, bit for bit (the fused value is
// computed by the same fuse_op as k_fuse2_into), with 8 instead of 12 + 4 bytes per voxel of
// traffic.  For streams of windows that only need the depth map (main.cpp:177).
template <int OP>
__global__ __launch_bounds__(256) void k_collapse_max_z_fused(const float* __restrict__ a,
                                                              const float* __restrict__ b, int npix,
                                                              int nz, float* __restrict__ conf,
                                                              uint8_t* __restrict__ idx,
                                                              const float* __restrict__ planes,
                                                              float* __restrict__ depth)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float* ca = a + p;
    const float* cb = b + p;
    float best = fuse_op<OP>(0.f + ca[0], cb[0]);
    int best_k = 0;
    int k = 1;
    for (; k + 4 <= nz; k += 4) {
        float va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = ca[(size_t)(k + u) * npix];
            vb[u] = cb[(size_t)(k + u) * npix];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v = fuse_op<OP>(0.f + va[u], vb[u]);
            if (best < v) {
                best = v;
                best_k = k + u;
            }
        }
    }
    for (; k < nz; ++k) {
        const float v = fuse_op<OP>(0.f + ca[(size_t)k * npix], cb[(size_t)k * npix]);
        if (best < v) {
            best = v;
            best_k = k;
        }
    }
    conf[p] = best;
    idx[p] = (uint8_t)best_k;
    if (depth) depth[p] = planes[best_k];
}

// collapseMaxZSlice of the n-ary fusion of up to 8 volumes (dsi_grid_fuse_n's begin / accumulate x n /
// finalize per voxel, the same operations in the same order, hence the same bits) without
// materialising the fused volume: n * 4 B per voxel read, nothing written but the maps.
template <int ACC, int FIN>
__global__ __launch_bounds__(256) void k_collapse_max_z_fused_n(FuseSources src, int n_src, int npix, int nz,
                                                                float identity, float fn,
                                                                float* __restrict__ conf,
                                                                uint8_t* __restrict__ idx,
                                                                const float* __restrict__ planes,
                                                                float* __restrict__ depth)
{
    __shared__ double2 log_tab[ACC == EW_ADD_LOG ? kLogTabSize : 1];
    if (ACC == EW_ADD_LOG) det_log_table_fill(log_tab);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float best = 0.f;
    int best_k = 0;
    for (int k = 0; k < nz; k += 2) {
        // two planes' loads in flight before the arithmetic
        float v[2][kMaxFuseSources];
        const int k1 = min(k + 1, nz - 1);
#pragma unroll
        for (int c = 0; c < kMaxFuseSources; ++c)
            if (c < n_src) {
                v[0][c] = src.p[c][(size_t)k * npix + p];
                v[1][c] = src.p[c][(size_t)k1 * npix + p];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (k + u >= nz) break;
            float acc = identity;
#pragma unroll
            for (int c = 0; c < kMaxFuseSources; ++c)
                if (c < n_src) acc = ew_op<ACC>(acc, v[u][c], 0.f, 0.f, log_tab);
            if (FIN >= 0) acc = ew_op<(FIN >= 0 ? FIN : 0)>(acc, 0.f, fn, 0.f);
            if (k + u == 0 || best < acc) {  // std::max_element: the first maximum wins
                best = acc;
                best_k = k + u;
            }
        }
    }
    conf[p] = best;
    idx[p] = (uint8_t)best_k;
    if (depth) depth[p] = planes[best_k];
}

// cartesian3dgrid.cpp:164-174: sum of squares in double (order differs from the
// sequential loop; relative difference ~1e-16 * log n)
__global__ __launch_bounds__(256) void k_mean_square(const float* __restrict__ dsi, size_t n,
                                                     double* __restrict__ accum)
{
    __shared__ double part[4];
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = (double)dsi[i];
        acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (part[0] + part[1]) + (part[2] + part[3]);
        unsafeAtomicAdd(accum, t);
    }
}

// --------------------------------------------------- post-arg-max filters ---
// Device side of MapperEMVS::getDepthMapFromDSI after the arg-max
// (mapper_emvs_stereo.cpp:390-436; the Telea inpainting of the dense map stays out).
// All of it is integer / exactly representable float work on a W x H image.
__device__ __forceinline__ uint32_t float_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotone float -> uint
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// :393 conf(0,0) = max_confidence, then min / max for cv::normalize(NORM_MINMAX)
__global__ __launch_bounds__(256) void k_conf_minmax(float* __restrict__ conf, int n,
                                                     float max_confidence,
                                                     uint32_t* __restrict__ mm /* [min,max] keys */)
{
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v = conf[i];
        if (i == 0) {
            v = max_confidence;
            conf[0] = v;
        }
        const uint32_t k = float_key(v);
        lo = min(lo, k);
        hi = max(hi, k);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_down((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_down((int)hi, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

__device__ __forceinline__ uint8_t saturate_u8(float v)
{
    const float r = __builtin_rintf(v);  // cvRound: half to even
    return (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}

// :394-397 normalize to [0,255] (scale/shift in double, applied in float), (0,0) = 0, to u8
__global__ __launch_bounds__(256) void k_conf8(const float* __restrict__ conf, int n,
                                               const uint32_t* __restrict__ mm,
                                               uint8_t* __restrict__ conf8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double smin = (double)key_float(mm[0]), smax = (double)key_float(mm[1]);
    const double range = smax - smin;
    const double scale = 255.0 * (range > 2.220446049250313e-16 ? 1. / range : 0.);
    const double shift = 0.0 - smin * scale;
    const float a = (float)scale, b = (float)shift;
    float v = conf[i] * a + b;
    if (i == 0) v = 0.f;
    conf8[i] = saturate_u8(v);
}

struct GaussK {
    float w[64];
};

// :403-409 cv::adaptiveThreshold(ADAPTIVE_THRESH_GAUSSIAN_C, THRESH_BINARY, ksize, delta = -C)
__global__ __launch_bounds__(256) void k_adaptive_mask(const uint8_t* __restrict__ conf8, int nx,
                                                       int ny, int ksize, GaussK kern, int idelta,
                                                       uint8_t* __restrict__ mask)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const int r = ksize / 2;
    float acc = 0.f;
    for (int ty = -r; ty <= r; ++ty) {  // column pass over row-pass results, BORDER_REPLICATE
        const int yy = min(max(y + ty, 0), ny - 1);
        float row = 0.f;
        for (int tx = -r; tx <= r; ++tx)
            row += kern.w[tx + r] * (float)conf8[(size_t)yy * nx + min(max(x + tx, 0), nx - 1)];
        acc += kern.w[ty + r] * row;
    }
    const int mean = saturate_u8(acc);
    mask[(size_t)y * nx + x] = ((int)conf8[(size_t)y * nx + x] - mean > -idelta) ? 1 : 0;
}

// :420-423 huangMedianFilter (median_filtering.cpp:33-158): for every pixel the median
// (compute_median_histogram: smallest v whose cumulative count reaches (num+1)/2) of the
// masked in-image values of its window; 0 when the window holds none
__global__ __launch_bounds__(256) void k_masked_median(const uint8_t* __restrict__ idx,
                                                       const uint8_t* __restrict__ mask, int nx,
                                                       int ny, int p, uint8_t* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nx || y >= ny) return;
    const int y0 = max(y - p, 0), y1 = min(y + p, ny - 1), x0 = max(x - p, 0), x1 = min(x + p, nx - 1);
    int num = 0;
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) num += mask[(size_t)yy * nx + xx] > 0 ? 1 : 0;
    const int middle = (num + 1) / 2;
    int best = 256;
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
            if (!(mask[(size_t)yy * nx + xx] > 0)) continue;
            const int v = idx[(size_t)yy * nx + xx];
            if (v >= best) continue;
            int c = 0;  // values <= v in the window
            for (int y2 = y0; y2 <= y1; ++y2)
                for (int x2 = x0; x2 <= x1; ++x2)
                    c += (mask[(size_t)y2 * nx + x2] > 0 && idx[(size_t)y2 * nx + x2] <= v) ? 1 : 0;
            if (c >= middle) best = v;
        }
    out[(size_t)y * nx + x] = (uint8_t)(num == 0 ? 0 : best);
}

// :426-427 removeMaskBoundary (:316-329) and :435 convertDepthIndicesToValues
__global__ __launch_bounds__(256) void k_finish_depth(uint8_t* __restrict__ mask,
                                                      const uint8_t* __restrict__ idxf, int nx,
                                                      int ny, int border,
                                                      const float* __restrict__ planes,
                                                      float* __restrict__ depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx * ny) return;
    const int x = i % nx, y = i / nx;
    if (x <= border || x >= nx - border || y <= border || y >= ny - border) mask[i] = 0;
    depth[i] = planes[idxf[i]];
}

__global__ void k_div_probe(const float* __restrict__ n, const float* __restrict__ d,
                            size_t count, float* __restrict__ q, float* __restrict__ ref)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float r = 1.f / d[i];
    q[i] = div_rc(n[i], d[i], r);
    ref[i] = n[i] / d[i];
}

// ---------------------------------------------------------------- exact tie resolver --
// The banded kernels sum a voxel's votes exactly (64-bit fixed point, one rounding); the reference adds them in
// fp32, in event order, rounding after every vote (cartesian3dgrid.h:261-270 inside mapper_emvs_stereo.cpp:197-201).
// Both are within ~1e-5 of each other, so the arg-max over Z can differ only where a column's two best planes are
// closer than that.  For those columns the resolver re-sums the contending voxels in the REFERENCE's order:
//   k_tie_columns      per pixel: does the column have >= 2 planes within rel_gap of its maximum (one streaming sweep)
//   k_tie_contenders   wave = such a column, lane = plane: the contending voxels
//   k_tie_hits_binned  every (packet, contending voxel): the events the plane transfer takes into the voxel's 2 x 2
//                      neighbourhood, voted with the reference's coordinates (IEEE divide), accept test and weights;
//                      a vote is recorded as (voxel, event order, weight)
//   (partition by voxel, per-voxel order in LDS, one-by-one fp32 sums: k_tie_partition, k_tie_sort_runs, k_tie_add_runs below)
//   k_tie_pick         thread = column: camera fusion, first maximum (cartesian3dgrid.cpp:132-134), patch
template <int OP>
__device__ __forceinline__ float tie_value(const float* __restrict__ a, const float* __restrict__ b, size_t i)
{
    if (OP == 0) return a[i];
    return fuse_op<OP>(0.f + a[i], b[i]);  // as k_collapse_max_z_fused / k_fuse2_into compute the fused voxel
}

// (1) thread = pixel, ONE streaming sweep over the column(s): the largest and second largest (fused) value; a column has
// >= 2 planes within rel_gap of its maximum iff the second largest is one of them.  The near-tie columns' pixels and maxima go
// to cols[] (x = pixel, y = float bits of the maximum), slots reserved per workgroup; counters[1] = their number (it keeps
// counting beyond cols_cap: the host retries with more room).
template <int OP>
__global__ __launch_bounds__(256) void k_tie_columns(const float* __restrict__ a, const float* __restrict__ b, int npix, int nz,
                                                     float rel_gap, unsigned* __restrict__ counters, uint4* __restrict__ cols,
                                                     uint32_t cols_cap)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    float best = 0.f, second = -__builtin_inff();
    if (p < npix) {
        best = tie_value<OP>(a, b, p);
        int k = 1;
        for (; k + 8 <= nz; k += 8) {  // eight planes' loads in flight before the arithmetic (as k_collapse_max_z streams)
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = a[(size_t)(k + u) * npix + p];
                vb[u] = OP == 0 ? 0.f : b[(size_t)(k + u) * npix + p];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = OP == 0 ? va[u] : fuse_op<OP>(0.f + va[u], vb[u]);
                second = best < v ? best : (second < v ? v : second);
                best = best < v ? v : best;
            }
        }
        for (; k < nz; ++k) {
            const float v = tie_value<OP>(a, b, (size_t)k * npix + p);
            second = best < v ? best : (second < v ? v : second);
            best = best < v ? v : best;
        }
    }
    // (an empty column is exactly zero in either summation order)
    const bool tie = p < npix && best > 0.f && second >= best - rel_gap * best;
    __shared__ unsigned s_n, s_base;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    const unsigned mine = tie ? atomicAdd(&s_n, 1u) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(&counters[1], s_n);
    __syncthreads();
    if (tie && s_base + mine < cols_cap) cols[s_base + mine] = make_uint4((unsigned)p, __float_as_uint(best), 0u, 0u);
}

// (2) WAVE = near-tie column, lane = plane: the contenders (value >= the column's threshold) by ballot -- two load
// instructions per 64 planes instead of a thread walking its column three times behind dependent loads.  cand[] <- the
// column's contending voxels (z * npix + p), one contiguous run, planes ascending; cols[j] <- (first entry of the run,
// pixel, contenders, 0); counters[0] = voxels (keeps counting beyond cap: nothing is written there, the host retries)
template <int OP>
__global__ __launch_bounds__(1024) void k_tie_contenders(const float* __restrict__ a, const float* __restrict__ b, int npix, int nz,
                                                         float rel_gap, unsigned* __restrict__ counters, uint32_t* __restrict__ cand,
                                                         uint32_t cap, uint4* __restrict__ cols, uint32_t cols_cap)
{
    // (a workgroup's 16 columns reserve their runs with ONE global atomic: thousands on one address serialise at ~13 ns apiece)
    __shared__ unsigned s_n, s_base;
    const unsigned n_cols = min(counters[1], cols_cap);
    const int lane = threadIdx.x & 63;
    for (unsigned j0 = blockIdx.x * 16u; j0 < n_cols; j0 += gridDim.x * 16u) {
        const unsigned j = j0 + (threadIdx.x >> 6);
        const bool live = j < n_cols;  // (wave-uniform)
        if (threadIdx.x == 0) s_n = 0u;
        __syncthreads();
        unsigned long long mask[4] = {0ull, 0ull, 0ull, 0ull};  // dimZ <= 256 (main.cpp:156)
        unsigned cnt = 0u, p = 0u, off = 0u;
        if (live) {
            const uint4 col = cols[j];
            p = col.x;
            const float best = __uint_as_float(col.y), thr = best - rel_gap * best;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int z = i * 64 + lane;
                if (i * 64 < nz) {
                    const bool in = z < nz && tie_value<OP>(a, b, (size_t)z * npix + p) >= thr;
                    mask[i] = __ballot(in);
                    cnt += (unsigned)__popcll(mask[i]);
                }
            }
            if (lane == 0) off = atomicAdd(&s_n, cnt);  // (LDS)
            off = (unsigned)__builtin_amdgcn_readfirstlane((int)off);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_n) s_base = atomicAdd(&counters[0], s_n);
        __syncthreads();
        if (live) {
            const unsigned base = s_base + off;
            if (lane == 0) cols[j] = make_uint4(base, p, cnt, 0u);
            if ((unsigned long long)base + cnt <= cap) {
                unsigned before = 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int z = i * 64 + lane;
                    if ((mask[i] >> lane) & 1ull) {
                        const unsigned at = before + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(mask[i] >> 32),
                                                                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask[i], 0u));
                        cand[base + at] = (uint32_t)((size_t)z * npix + p);
                    }
                    before += (unsigned)__popcll(mask[i]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_tie_patch(const uint32_t* __restrict__ pix, const uint8_t* __restrict__ new_idx,
                                                   const float* __restrict__ new_conf, int n,
                                                   const float* __restrict__ planes, float* __restrict__ conf,
                                                   uint8_t* __restrict__ idx, float* __restrict__ depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pix[i];
    conf[p] = new_conf[i];
    idx[p] = new_idx[i];
    depth[p] = planes[new_idx[i]];  // mapper_emvs_stereo.cpp:302-313
}

// ---- the resolver's event pass, INVERTED: (packet, contending voxel) pairs instead of (event, contending plane) pairs ----
// Walking every event over every contending plane (1 G event-planes per configs[1] camera, two IEEE divides each: round
// 4's k_tie_hits, 4.9 ms) finds the ~0.7 % of them that reach a contending voxel.  This kernel turns the loop around: a block
// takes a packet, bins its 1024 z0 locations by 2-D tile in LDS (counting sort), and every thread asks, for one
// contending voxel (x, y, z) at a time, WHICH z0 locations the transfer of mapper_emvs_stereo.cpp:177-195 can take into
// the 2 x 2 integer locations whose bilinear vote touches that voxel: X = (x0 a + bx) / d is affine in x0, so the
// pre-image of X in [x - 1, x + 1) is an interval of x0 (widened by a rigorous bound on the fp32 rounding of both
// directions), i.e. a few tiles of the packet.  The records of those tiles are then voted EXACTLY -- the reference's
// operations in the reference's order, IEEE divide, accept test of cartesian3dgrid.h:255-259, weights of :261-270 --
// so the tile test only has to be a superset.  Work: voxels x packets cheap rejections (5.5 k x 9,765 at configs[1] =
// 54 M, against 1 G event-planes) + the few records near a voxel.
// A (packet, plane) whose transfer is degenerate (a = 0: every event lands on one location; non-finite or extreme
// coefficients) scans all records of the packet; d = 0 gives no finite X: no votes.
// Output: (key, weight) records, key = rank << pos_bits | position of the vote in the reference's loop over events
// (packet * 1024 + slot); rank = rank_base + index of the voxel in `desc`.  Slots: a block owns SEGMENTS of kTieSeg
// records; a hit takes the next position of its block's virtual cursor (one LDS atomic), the hit that opens a segment
// reserves it with ONE global atomic (7.5 M hits -> ~8 k global atomics; one per hit or per wave instruction
// serialises at the memory side: ~50 ns each, 331 ms at configs[1] in round 4) and publishes its index in LDS, where the other hits of that segment
// wait for it.  A block pads its last segment with sentinel keys (rank = sentinel_rank, beyond every voxel's), so that
// the array [0, segments * kTieSeg) can be sorted as it is.
constexpr int kTieSegShift = 10;
constexpr int kTieSeg = 1 << kTieSegShift;
constexpr int kTieMaxSegs = 512;  // per block: 512 k hits
constexpr unsigned kTieSegEmpty = 0xffffffffu;
constexpr int kTieMaxTiles = 16384;  // tile starts are 16-bit halves of LDS words: 32 KB

struct alignas(16) TiePlane {  // per (packet, plane), in LDS
    float a, bx, by, d;  // mapper_emvs_stereo.cpp:177-182
    float ia, qx, qy;    // z0 location of integer X: x0 = X * ia + qx (ia = d / a, qx = -bx / a), same for y
    float hw;            // half width of the pre-image of [X - 1, X + 1) in z0 pixels, rounding included; < 0: no votes;
                         // +inf (with ia = qx = qy = 0): scan the whole packet
    float c_lo, c_hi;    // tile of the box's edges: floor(fma(centre, 1 / tile, c_lo | c_hi)), c = (margin -+ hw) / tile
    float pad_[2];
};

struct TieBinGeom {
    int shift;         // log2 of the tile side
    int margin;        // the tiles cover [-margin, n + margin) in x and y; locations outside clamp to the border tiles
    int tx_n, ty_n;    // tiles per row / column
};

__host__ __device__ inline TieBinGeom tie_bin_geom(int nx, int ny)
{
    TieBinGeom b;
    b.margin = 32;
    for (b.shift = 2;; ++b.shift) {
        b.tx_n = (nx + 2 * b.margin + (1 << b.shift) - 1) >> b.shift;
        b.ty_n = (ny + 2 * b.margin + (1 << b.shift) - 1) >> b.shift;
        if (b.tx_n * b.ty_n <= kTieMaxTiles) break;
    }
    return b;
}

__device__ __forceinline__ int tie_tile_coord(float v, int margin, float inv_tile, int n_tiles)
{
    // monotone in v (add, multiply by a positive constant, floor, clamp): lo <= v <= hi implies tile(lo) <= tile(v) <= tile(hi)
    const float t = __builtin_floorf((v + (float)margin) * inv_tile);
    return (int)fminf(fmaxf(t, 0.f), (float)(n_tiles - 1));
}

struct TieEvents {  // the raw events of a camera's batch + what stage A needs per packet
    const uint16_t* x;
    const uint16_t* y;
    const uint32_t* first;  // first event of packet k (nullptr: k * 1024)
    const float* H;         // [np][9] (k_packet_geometry)
    const float2* lut;
    int sensor_w, sensor_h;
};

struct TieOut {
    unsigned* s_vpos;          // LDS: the block's virtual cursor
    unsigned* s_seg;           // LDS: segment table [kTieMaxSegs]
    unsigned* seg_counter;     // global: segments handed out
    unsigned* flags;           // global: bit 0 = a block ran out of table entries, bit 1 = out of segments (cap)
    unsigned cap_segs;
    unsigned long long* keys;
    float* wts;
};

__device__ __forceinline__ void tie_emit(const TieOut& o, unsigned long long key, float w)
{
    const unsigned vp = atomicAdd(o.s_vpos, 1u);  // (LDS)
    const unsigned j = vp >> kTieSegShift, off = vp & (unsigned)(kTieSeg - 1);
    if (j >= (unsigned)kTieMaxSegs) {  // (one global atomic per event of this kind, not per hit: they serialise)
        if (vp == (unsigned)kTieMaxSegs * (unsigned)kTieSeg) atomicOr(o.flags, 1u);
        return;
    }
    // step 1, all lanes of the wave that hit: the lane that opens a segment reserves and publishes it ...
    if (off == 0u) {
        const unsigned b = atomicAdd(o.seg_counter, 1u);
        if (b >= o.cap_segs) atomicOr(o.flags, 2u);
        __hip_atomic_store(&o.s_seg[j], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ... step 2: everybody reads it.  A lane can only wait here for a lane of ANOTHER wave (one of its own wave has
    // finished step 1 before any lane starts step 2), and that lane depends on nobody.
    unsigned b;
    while ((b = __hip_atomic_load(&o.s_seg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == kTieSegEmpty)
        __builtin_amdgcn_s_sleep(1);
    if (b >= o.cap_segs) return;  // (flagged by the lane that opened the segment; the host repeats the pass with more room)
    const size_t at = ((size_t)b << kTieSegShift) + off;
    o.keys[at] = key;
    o.wts[at] = w;
}

constexpr int kTieQueue = 128;  // queued (voxel, record) pairs per wave

// The exact vote of record r of packet k on voxel c: the reference's coordinates (mapper_emvs_stereo.cpp:194-195), accept test
// (cartesian3dgrid.h:255-259, see vote_global) and the bilinear weight of the corner the voxel is (:261-270)
__device__ __forceinline__ void tie_vote_pair(const TieOut& out, const uint2* __restrict__ desc, const TiePlane* __restrict__ tp,
                                              const float* __restrict__ rec_x, const float* __restrict__ rec_y,
                                              const unsigned* __restrict__ rec_slot, int c, unsigned r, int k, unsigned rank_base,
                                              unsigned pos_bits, float xmax, float ymax)
{
    const uint2 dsc = desc[c];
    const int vx = (int)(dsc.x & 0xffffu), vy = (int)(dsc.x >> 16), vz = (int)dsc.y;
    const float a = tp[vz].a, bx = tp[vz].bx, by = tp[vz].by, d = tp[vz].d;
    const float x0 = rec_x[r], y0 = rec_y[r];
    const float X = (x0 * a + bx) / d;
    const float Y = (y0 * a + by) / d;
    if (!(X >= 0.f && Y >= 0.f && X < xmax && Y < ymax)) return;
    const int xi = (int)X, yi = (int)Y;
    const unsigned dx = (unsigned)(vx - xi), dy = (unsigned)(vy - yi);
    if (dx > 1u || dy > 1u) return;
    const float fx = X - (float)xi, fy = Y - (float)yi, fx1 = 1.f - fx, fy1 = 1.f - fy;
    const float w = (dx ? fx : fx1) * (dy ? fy : fy1);  // fx1*fy1, fx*fy1, fx1*fy, fx*fy
    tie_emit(out, ((unsigned long long)(rank_base + (unsigned)c) << pos_bits) | ((unsigned long long)k * kPacket + rec_slot[r]), w);
}

// dynamic LDS: rec_x[1024] rec_y[1024] (f32) | rec_slot[1024] (u32) | tile starts, tiles + 1 of them as the 16-bit halves
// of (tiles + 2) / 2 words | TiePlane[nz] | seg[kTieMaxSegs] (u32)
// (a multiple of four words, so that the TiePlane table behind them is 16-byte aligned: one ds_read_b128 per look-up)
__host__ __device__ inline int tie_tile_words(int tiles) { return ((tiles + 2) / 2 + 3) & ~3; }
__host__ __device__ inline size_t tie_hits_lds_bytes(int tiles, int nz)
{
    return (size_t)kPacket * 12 + (size_t)tie_tile_words(tiles) * 4 + (size_t)nz * sizeof(TiePlane) + (size_t)kTieMaxSegs * 4;
}
__device__ __forceinline__ unsigned tie_tile_start(const unsigned* __restrict__ tw, int t)
{
    // half (t & 1) of word t >> 1 = 16-bit element t (little endian): ONE ds_read_u16 instead of a word read + shift + mask --
    // four of these per (voxel, packet) pair, 54 M pairs per camera
    return reinterpret_cast<const uint16_t*>(tw)[t];
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_tie_hits_binned(TieEvents ev, const float* __restrict__ centers,
                                                         const float* __restrict__ planes, Geom g, int np,
                                                         const uint2* __restrict__ desc, int nsv, unsigned rank_base,
                                                         unsigned pos_bits, unsigned sentinel_rank, TieBinGeom bg,
                                                         unsigned* __restrict__ seg_counter, unsigned cap_segs,
                                                         unsigned* __restrict__ flags,
                                                         unsigned long long* __restrict__ total_hits,
                                                         unsigned long long* __restrict__ keys, float* __restrict__ wts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    __shared__ unsigned s_vpos;
    __shared__ unsigned s_wave_tot[BLOCK / 64];
    constexpr int PER = kPacket / BLOCK;  // events of a packet per thread
    __shared__ unsigned s_q[BLOCK / 64][kTieQueue];  // per wave: (voxel - v_beg) << 10 | record, pairs that passed the box test
    __shared__ unsigned s_qn[BLOCK / 64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < BLOCK / 64) s_qn[threadIdx.x] = 0u;
    float* rec_x = reinterpret_cast<float*>(s_raw);
    float* rec_y = rec_x + kPacket;
    unsigned* rec_slot = reinterpret_cast<unsigned*>(rec_y + kPacket);
    const int tiles = bg.tx_n * bg.ty_n, twords = tie_tile_words(tiles);
    unsigned* tstart = rec_slot + kPacket;  // tile t: half (t & 1) of word t >> 1; entry `tiles` = the packet's records
    TiePlane* tp = reinterpret_cast<TiePlane*>(__builtin_assume_aligned(tstart + twords, 16));  // (kPacket * 12 + 4 * twords: 16 | both)
    unsigned* s_seg = reinterpret_cast<unsigned*>(tp + g.nz);
    const int tid = threadIdx.x;
    const float inv_tile = 1.f / (float)(1 << bg.shift);
    const float xmax = (float)(g.nx - 1), ymax = (float)(g.ny - 1);
    TieOut out{&s_vpos, s_seg, seg_counter, flags, cap_segs, keys, wts};
    if (tid == 0) s_vpos = 0u;
    for (int i = tid; i < kTieMaxSegs; i += BLOCK) s_seg[i] = kTieSegEmpty;

    // this block's share of the voxels (blockIdx.y): few packets (a 50 ms window: ~490) do not fill the chip by themselves
    const int v_per = (nsv + (int)gridDim.y - 1) / (int)gridDim.y;
    const int v_beg = (int)blockIdx.y * v_per, v_end = min(nsv, v_beg + v_per);
    for (int k = blockIdx.x; k < np; k += gridDim.x) {
        __syncthreads();  // the previous packet's voxel loop has left the tables
        for (int i = tid; i < twords; i += BLOCK) tstart[i] = 0u;
        {
            const float cx_ = centers[3 * k], cy_ = centers[3 * k + 1], cz_ = centers[3 * k + 2];
            for (int z = tid; z < g.nz; z += BLOCK) {
                TiePlane P;
                plane_coefficients(cx_, cy_, cz_, planes[z], g, P.a, P.bx, P.by, P.d);
                P.ia = P.d / P.a;
                P.qx = -P.bx / P.a;
                P.qy = -P.by / P.a;
                // |X~ - X| <= 3u (nx + 2 |bx / d| + 1), the centre x * ia + qx is off by <= 6u (nx + |bx / d|) |ia|, u = 2^-24:
                // 2^-16 (...) is more than 16 times their sum
                const float beta = fabsf(P.bx / P.d) + fabsf(P.by / P.d);
                const float delta = 1.52587890625e-5f * ((float)(g.nx + g.ny + 2) + beta);
                P.hw = fabsf(P.ia) * (1.f + delta) * 1.001f + 1e-6f;
                const bool finite = __builtin_isfinite(P.a) && __builtin_isfinite(P.bx) && __builtin_isfinite(P.by) &&
                                    __builtin_isfinite(P.d);
                if (finite && P.d == 0.f) {
                    P.hw = -1.f;  // n / 0 is +-inf or NaN: no event is accepted on this plane
                } else if (!finite || !__builtin_isfinite(P.hw) || !__builtin_isfinite(P.qx) || !__builtin_isfinite(P.qy) ||
                           !__builtin_isfinite(beta) || !(fabsf(P.ia) >= 1e-6f && fabsf(P.ia) <= 1e6f) || !(delta < 0.5f)) {
                    P.ia = P.qx = P.qy = 0.f;
                    P.hw = __builtin_inff();
                }
                // (widened by 1e-3 tile: the fused multiply-add, these constants' own roundings and the two roundings of a
                //  record's tile (tie_tile_coord) are each below 3e-5 tile wherever a tile index is not clamped, so a record
                //  with edge <= x has tile(edge) <= tile(x) whatever |ia| is; hw = +inf: -inf / +inf -- the clamps take the row)
                P.c_lo = ((float)bg.margin - P.hw) * inv_tile - 1e-3f;
                P.c_hi = ((float)bg.margin + P.hw) * inv_tile + 1e-3f;
                P.pad_[0] = P.pad_[1] = 0.f;
                tp[z] = P;
            }
        }
        // the packet's z0 locations, counted per tile; a non-finite location votes on no plane
        float2 e[PER];
        int tile[PER];
        unsigned off[PER];
        {
            // stage A of the packet's events here (mapper_emvs_stereo.cpp:129-142, the function k_warp_z0 applies: same bits)
            // instead of a z0 array written and read back per call (80 MB and 28 us per camera at configs[1])
            const size_t first = ev.first ? (size_t)ev.first[k] : (size_t)k * kPacket;
            float hh[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) hh[i] = ev.H[9 * (size_t)k + i];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const size_t at = first + (size_t)(tid + BLOCK * i);
                e[i] = warp_event_z0(ev.x[at], ev.y[at], hh, ev.lut, ev.sensor_w, ev.sensor_h);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool ok = __builtin_isfinite(e[i].x) && __builtin_isfinite(e[i].y);
            tile[i] = ok ? tie_tile_coord(e[i].y, bg.margin, inv_tile, bg.ty_n) * bg.tx_n +
                               tie_tile_coord(e[i].x, bg.margin, inv_tile, bg.tx_n)
                         : -1;
            // (LDS; a packet holds 1024 events: a half word cannot carry into its neighbour)
            off[i] = ok ? (atomicAdd(&tstart[tile[i] >> 1], 1u << ((tile[i] & 1) << 4)) >> ((tile[i] & 1) << 4)) & 0xffffu : 0u;
        }
        __syncthreads();
        // exclusive scan of the tile counts -> tile starts (thread t owns a contiguous stretch of tiles)
        {
            const int per = (twords + BLOCK - 1) / BLOCK;
            const int t0 = tid * per, t1 = min(twords, t0 + per);
            unsigned sum = 0u;
            for (int t = t0; t < t1; ++t) {
                const unsigned c = tstart[t];
                sum += (c & 0xffffu) + (c >> 16);
            }
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_up(incl, o, 64);
                if ((tid & 63) >= o) incl += v;
            }
            if ((tid & 63) == 63) s_wave_tot[tid >> 6] = incl;
            __syncthreads();
            unsigned base = incl - sum;
            for (int w = 0; w < (tid >> 6); ++w) base += s_wave_tot[w];
            for (int t = t0; t < t1; ++t) {
                const unsigned c = tstart[t], lo_n = c & 0xffffu;
                tstart[t] = base | ((base + lo_n) << 16);
                base += lo_n + (c >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (tile[i] >= 0) {
                const unsigned r = tie_tile_start(tstart, tile[i]) + off[i];
                rec_x[r] = e[i].x;
                rec_y[r] = e[i].y;
                rec_slot[r] = (unsigned)(tid + BLOCK * i);
            }
        __syncthreads();
        // Every contending voxel of this block against this packet.  A thread scans the records of its voxel's tiles
        // against the exact pre-image box; the ~10 % of (voxel, record) pairs that pass are QUEUED (per wave, in LDS) and
        // voted 64 at a time with all lanes busy -- voted where they are found, the two IEEE divisions, the weights and
        // the output slot would run for the few lanes that have a hit while the rest of the wave waits.
        // (a voxel's descriptor is loaded one turn ahead: read where it is used, its two words -- the plane first, for the
        //  early exit -- were two dependent L2 round trips per turn, in a loop whose waves are parked 69 % of the time)
        uint2 dsc_next = make_uint2(0u, 0u);
        if (v_beg + tid < v_end) dsc_next = desc[v_beg + tid];
        for (int c0 = v_beg; c0 < v_end; c0 += BLOCK) {
            const int c = c0 + tid;
            const uint2 dsc = dsc_next;
            if (c + BLOCK < v_end) dsc_next = desc[c + BLOCK];
            if (c < v_end) {
                const int vx = (int)(dsc.x & 0xffffu), vy = (int)(dsc.x >> 16), vz = (int)dsc.y;
                const float4 look = *reinterpret_cast<const float4*>(&tp[vz].ia);  // ia, qx, qy, hw
                const float hw = look.w;
                if (hw >= 0.f) {
                    const float ia = look.x, qx = look.y, qy = look.z;
                    // the box's tiles from the centre with ONE fused multiply-add per edge and per-plane constants (this is the
                    // superset filter, not the reference's arithmetic; its roundings are covered by the constants' widening,
                    // see where they are made): 18 vector instructions per (voxel, packet) instead of the 32 of
                    // centre -+ hw, (edge + margin) / tile, floor, two clamps (same box: 305 -> 296 us per camera)
                    const float cx0 = __builtin_fmaf((float)vx, ia, qx), cy0 = __builtin_fmaf((float)vy, ia, qy);
                    const float2 cc = *reinterpret_cast<const float2*>(&tp[vz].c_lo);
                    const float fx_n = (float)(bg.tx_n - 1), fy_n = (float)(bg.ty_n - 1);
                    const int txlo = (int)__builtin_amdgcn_fmed3f(__builtin_floorf(__builtin_fmaf(cx0, inv_tile, cc.x)), 0.f, fx_n);
                    const int txhi = (int)__builtin_amdgcn_fmed3f(__builtin_floorf(__builtin_fmaf(cx0, inv_tile, cc.y)), 0.f, fx_n);
                    const int tylo = (int)__builtin_amdgcn_fmed3f(__builtin_floorf(__builtin_fmaf(cy0, inv_tile, cc.x)), 0.f, fy_n);
                    const int tyhi = (int)__builtin_amdgcn_fmed3f(__builtin_floorf(__builtin_fmaf(cy0, inv_tile, cc.y)), 0.f, fy_n);
                    // (measured and dropped: both tile rows' starts read together and one loop over the records of both --
                    //  313 against 304 us)
                    for (int ty = tylo; ty <= tyhi; ++ty) {
                        const unsigned rb = tie_tile_start(tstart, ty * bg.tx_n + txlo), re = tie_tile_start(tstart, ty * bg.tx_n + txhi + 1);
                        // (the loop variable is the record's LDS address itself: no index -> address step per record)
                        for (const float* px = rec_x + rb; px < rec_x + re; ++px) {
                            const float x0 = px[0], y0 = px[kPacket];  // (rec_y = rec_x + kPacket)
                            // the pre-image test proper (the tiles are coarser), as max(|x0 - cx0|, |y0 - cy0|) <= hw: two
                            // subtractions, one maximum with |.| operands and ONE comparison, where four comparisons against
                            // xlo / xhi / ylo / yhi took four compares + three scalar mask operations -- in the loop whose trip count
                            // is the longest lane's, i.e. where most of this kernel's instructions are.  The subtraction's own
                            // rounding (<= 2^-24 hw) moves the cut by 1e-7 hw; hw carries 16 times the bound it needs (above).
                            // Never NaN: records are finite, ia and q finite, hw finite or +inf.
                            if (!(fmaxf(fabsf(x0 - cx0), fabsf(y0 - cy0)) <= hw)) continue;
                            const unsigned pos = atomicAdd(&s_qn[wv], 1u);  // (LDS)
                            const unsigned r = (unsigned)(px - rec_x);
                            if (pos < (unsigned)kTieQueue)
                                s_q[wv][pos] = ((unsigned)(c - v_beg) << 10) | r;
                            else  // a full queue (a burst of events on one pixel): vote in place
                                tie_vote_pair(out, desc, tp, rec_x, rec_y, rec_slot, c, r, k, rank_base, pos_bits, xmax, ymax);
                        }
                    }
                }
            }
            // (the wave is whole again) full batches of the queue
            unsigned n = min(*(volatile unsigned*)&s_qn[wv], (unsigned)kTieQueue);
            while (n >= 64u) {
                n -= 64u;
                const unsigned e = *(volatile unsigned*)&s_q[wv][n + (unsigned)lane];
                tie_vote_pair(out, desc, tp, rec_x, rec_y, rec_slot, v_beg + (int)(e >> 10), e & 1023u, k, rank_base, pos_bits, xmax, ymax);
            }
            if (lane == 0) *(volatile unsigned*)&s_qn[wv] = n;
        }
        {  // the rest of the queue: its entries point into this packet's records
            const unsigned n = min(*(volatile unsigned*)&s_qn[wv], (unsigned)kTieQueue);
            if ((unsigned)lane < n) {
                const unsigned e = *(volatile unsigned*)&s_q[wv][lane];
                tie_vote_pair(out, desc, tp, rec_x, rec_y, rec_slot, v_beg + (int)(e >> 10), e & 1023u, k, rank_base, pos_bits, xmax, ymax);
            }
            if (lane == 0) *(volatile unsigned*)&s_qn[wv] = 0u;
        }
    }
    __syncthreads();
    // pad the last segment with sentinels; report the block's hits
    const unsigned total = s_vpos;
    if (total != 0u && total <= (unsigned)kTieMaxSegs * (unsigned)kTieSeg) {
        const unsigned j = (total - 1u) >> kTieSegShift;
        const unsigned b = s_seg[j];
        const unsigned first = total - (j << kTieSegShift);  // records of the last segment in use: 1 .. kTieSeg
        if (b != kTieSegEmpty && b < cap_segs)
            for (unsigned o = first + (unsigned)tid; o < (unsigned)kTieSeg; o += (unsigned)BLOCK) {
                keys[((size_t)b << kTieSegShift) + o] = (unsigned long long)sentinel_rank << pos_bits;
                wts[((size_t)b << kTieSegShift) + o] = 0.f;
            }
    }
    if (tid == 0 && total != 0u) atomicAdd(total_hits, (unsigned long long)total);
}

// desc[c] = (x | y << 16, z) of voxel vox[c] = z * npix + y * nx + x
// plane_bits (optional, 8 zeroed words): bit z <- plane z holds one of the voxels (statistics; collected per workgroup in LDS)
__global__ __launch_bounds__(256) void k_tie_desc(const uint32_t* __restrict__ vox, int n, int nx, int npix, uint2* __restrict__ desc,
                                                  unsigned* __restrict__ plane_bits)
{
    __shared__ unsigned s_bits[8];
    if (threadIdx.x < 8) s_bits[threadIdx.x] = 0u;
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) {
        const uint32_t v = vox[c];
        const uint32_t z = v / (uint32_t)npix, p = v - z * (uint32_t)npix, y = p / (uint32_t)nx, x = p - y * (uint32_t)nx;
        desc[c] = make_uint2(x | (y << 16), z);
        if (plane_bits && z < 256u && !((s_bits[z >> 5] >> (z & 31u)) & 1u)) atomicOr(&s_bits[z >> 5], 1u << (z & 31u));
    }
    __syncthreads();
    if (plane_bits && threadIdx.x < 8 && s_bits[threadIdx.x] && (s_bits[threadIdx.x] & ~plane_bits[threadIdx.x]))
        atomicOr(&plane_bits[threadIdx.x], s_bits[threadIdx.x]);
}

// ---- round 6: the recorded votes PARTITIONED by (camera, voxel) and ordered per voxel in LDS, instead of one device-wide
// radix sort of 38-bit keys (rocPRIM: 5 passes over 15.7 M pairs, 0.74 ms at configs[1] -- the last library call of the engine).
// Nothing needs the votes of DIFFERENT voxels in any order, and a voxel has a few thousand of them:
//   k_tie_partition<false>   votes per rank r = camera * nsv + voxel (a workgroup counts its stretch in LDS first: one global
//                            atomic per (workgroup, rank), not per vote -- same-address atomics serialise at the memory side)
//   k_tie_scan               exclusive prefix -> where each rank's run begins
//   k_tie_partition<true>    every vote into its rank's run, as ONE 64-bit word (event position << 32 | weight bits), in any order
//   k_tie_sort_runs          workgroup = rank: the run put in event order in LDS (a bucket sort by event position: tie_block_sort),
//                            its weights written back in that order.  Runs beyond 4,096 votes (configs[4]: ~30 k in a voxel)
//                            go through the same LDS in windows of event positions.
//   k_tie_add_runs           wave = rank: the weights added one by one in fp32 -- resetGrid (mapper_emvs_stereo.cpp:145),
//                            "grid[i] += w" per vote in event order (cartesian3dgrid.h:261-270)
constexpr int kTiePartLdsRanks = 30720;  // 120 KB of counters
constexpr unsigned long long kTiePartStretches = 256ull;  // stretches of the recorded votes per partition launch (at most)
constexpr int kTieRunLds = 4096;         // words of a wave's sorting buffer (32 KB)

template <bool SCATTER>
__global__ __launch_bounds__(1024) void k_tie_partition(const unsigned long long* __restrict__ keys, const float* __restrict__ wts,
                                                        unsigned long long n_rec, unsigned long long per_block, unsigned pos_bits,
                                                        unsigned n_ranks, int use_lds, uint32_t* __restrict__ counts_or_cursor,
                                                        const uint32_t* __restrict__ starts, unsigned long long* __restrict__ runs)
{
    extern __shared__ uint32_t s_rank[];  // use_lds: n_ranks counters, then (SCATTER) the next free slot of each rank's run
    const unsigned long long b0 = (unsigned long long)blockIdx.x * per_block, b1 = min(n_rec, b0 + per_block);
    if (b0 >= b1) return;
    const unsigned long long pos_mask = (1ull << pos_bits) - 1ull;
    if (!use_lds) {  // too many ranks for the LDS: one global atomic per vote (they spread over that many addresses)
        for (unsigned long long i = b0 + threadIdx.x; i < b1; i += 1024) {
            const unsigned long long key = keys[i];
            const unsigned r = (unsigned)(key >> pos_bits);
            if (r >= n_ranks) continue;  // (the sentinel tails of the blocks' last segments)
            const uint32_t at = atomicAdd(&counts_or_cursor[r], 1u);
            if (SCATTER) runs[(size_t)starts[r] + at] = ((key & pos_mask) << 32) | (unsigned long long)__float_as_uint(wts[i]);
        }
        return;
    }
    if (SCATTER) {
        // where this stretch's votes of rank r go: the run's start + the votes of r in the stretches before this one
        // (k_tie_colscan turned the table of counts into these offsets)
        const uint32_t* const before = counts_or_cursor + (size_t)blockIdx.x * n_ranks;
        for (unsigned r = threadIdx.x; r < n_ranks; r += 1024) s_rank[r] = starts[r] + before[r];
    } else {
        for (unsigned r = threadIdx.x; r < n_ranks; r += 1024) s_rank[r] = 0u;
        __syncthreads();
        for (unsigned long long i = b0 + threadIdx.x; i < b1; i += 1024) {
            const unsigned r = (unsigned)(keys[i] >> pos_bits);
            if (r < n_ranks) atomicAdd(&s_rank[r], 1u);  // (LDS)
        }
    }
    __syncthreads();
    if (!SCATTER) {  // this stretch's row of the table [stretch][rank] (launch_tie_partition_sums)
        uint32_t* const row = counts_or_cursor + (size_t)blockIdx.x * n_ranks;
        for (unsigned r = threadIdx.x; r < n_ranks; r += 1024) row[r] = s_rank[r];
        return;
    }
    // Four records per thread and turn, their loads in flight together.  What this launch costs is its 8-byte stores: a
    // stretch holds ~5 votes per rank, so nearly every store is its own partial-line write into a 125 MB array (0.24 ms at
    // configs[1] = 65 G stores/s; the counting launch reads the same keys in 0.04 ms).  Measured: one record per turn 0.287,
    // four 0.277, without the returning atomics on the runs' cursors (the table) 0.240 ms; 512 / 1,024 stretches 0.287 / 0.265.
    constexpr int U = 4;
    for (unsigned long long i0 = b0 + threadIdx.x; i0 < b1; i0 += 1024ull * U) {
        unsigned long long key[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long i = i0 + 1024ull * u;
            const bool in = i < b1;
            key[u] = in ? keys[i] : ~0ull;  // (rank bits all ones: >= n_ranks)
            w[u] = in ? wts[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned r = (unsigned)(key[u] >> pos_bits);
            if (r >= n_ranks) continue;
            const uint32_t slot = atomicAdd(&s_rank[r], 1u);  // (LDS)
            runs[slot] = ((key[u] & pos_mask) << 32) | (unsigned long long)__float_as_uint(w[u]);
        }
    }
}

// starts[r] = votes of the ranks below r (starts[n_ranks] = all of them); cursor[r] = 0.  One workgroup.
__global__ __launch_bounds__(1024) void k_tie_scan(const uint32_t* __restrict__ counts, unsigned n_ranks, uint32_t* __restrict__ starts,
                                                   uint32_t* __restrict__ cursor)
{
    __shared__ uint32_t wave_tot[16];
    const unsigned per = (n_ranks + 1023u) / 1024u;
    const unsigned r0 = min(n_ranks, threadIdx.x * per), r1 = min(n_ranks, r0 + per);
    uint32_t local = 0;
    for (unsigned r = r0; r < r1; ++r) local += counts[r];
    uint32_t incl = local;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - local, total = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += wave_tot[w];
        total += wave_tot[w];
    }
    for (unsigned r = r0; r < r1; ++r) {
        starts[r] = base;
        base += counts[r];
        if (cursor) cursor[r] = 0u;
    }
    if (threadIdx.x == 0) starts[n_ranks] = total;
}

// table[b][r] = votes of rank r in stretch b  ->  votes of rank r in the stretches BEFORE b; counts[r] = all of them.
// Thread = rank (consecutive threads read consecutive words of a row); eight rows' loads in flight.
__global__ __launch_bounds__(64) void k_tie_colscan(uint32_t* __restrict__ table, unsigned n_ranks, unsigned n_stretches,
                                                    uint32_t* __restrict__ counts)
{
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranks) return;
    uint32_t running = 0u;
    unsigned b = 0;
    for (; b + 8 <= n_stretches; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = table[(size_t)(b + u) * n_ranks + r];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            table[(size_t)(b + u) * n_ranks + r] = running;
            running += c[u];
        }
    }
    for (; b < n_stretches; ++b) {
        const uint32_t c = table[(size_t)b * n_ranks + r];
        table[(size_t)b * n_ranks + r] = running;
        running += c;
    }
    counts[r] = running;
}

// The n <= kTieRunLds words of a run (any order; event position << 32 | weight bits) -> ascending in tmp[0 .. n), by the whole
// workgroup (T threads).  The words are dealt to nb >= n buckets of equal position width (one LDS atomic each: count and
// rank in the bucket), the bucket starts are a prefix sum, every word goes to its bucket's stretch, and inside a bucket every
// word finds its place by counting the smaller ones.  ~6 passes over the run instead of a 66-stage bitonic network (which cost
// 2 ms at configs[1] with a wave per run).  A burst -- all events of ONE packet on a voxel -- is one bucket of 1,024 words:
// 1,024 reads per word, spread over the workgroup: slow, correct, rare.
template <int CAP, int T>
__device__ __forceinline__ void tie_block_sort(const unsigned long long* __restrict__ src, int n,
                                               unsigned long long* __restrict__ tmp, uint32_t* __restrict__ bcnt, uint32_t* __restrict__ wave_tot)
{
    constexpr int PER = CAP / T, W = T / 64;  // (wave_tot: 2 * W words)
    const int tid = (int)threadIdx.x;
    int nb = 64;
    while (nb < n) nb <<= 1;
    for (int i = tid; i < nb; i += T) bcnt[i] = 0u;
    unsigned long long key[PER];
    uint32_t rank[PER], bucket[PER];
    // the run's own span of event positions (a voxel is voted while the camera looks past it: its votes sit in a stretch of
    // the stream, and buckets over the WHOLE stream would put them into a few): min and max over the workgroup
    uint32_t pmin = 0xffffffffu, pmax = 0u;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * T;
        if (i < n) {
            key[k] = src[i];
            const uint32_t p = (uint32_t)(key[k] >> 32);
            pmin = min(pmin, p);
            pmax = max(pmax, p);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        pmin = min(pmin, (uint32_t)__shfl_xor((int)pmin, off, 64));
        pmax = max(pmax, (uint32_t)__shfl_xor((int)pmax, off, 64));
    }
    if ((tid & 63) == 0) {
        wave_tot[tid >> 6] = pmin;
        wave_tot[W + (tid >> 6)] = pmax;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < W; ++w) {
        pmin = min(pmin, wave_tot[w]);
        pmax = max(pmax, wave_tot[W + w]);
    }  // (wave_tot is written again by the prefix sum below, behind the next barrier)
    // bucket = floor((p - pmin) * scale), scale a little below nb / (span + 1): monotone in p, < nb
    const float scale = (float)nb / ((float)(pmax - pmin) + 1.f) * 0.99999f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * T;
        if (i < n) {
            const uint32_t p = (uint32_t)(key[k] >> 32);
            bucket[k] = min((uint32_t)((float)(p - pmin) * scale), (uint32_t)(nb - 1));
            rank[k] = atomicAdd(&bcnt[bucket[k]], 1u);  // (LDS)
        }
    }
    __syncthreads();
    // exclusive prefix of the bucket counts, in place (thread t owns a contiguous stretch of nb / 256 buckets)
    {
        const int per = max(1, nb / T), b0 = tid * per, b1 = min(nb, b0 + per);
        uint32_t local = 0;
        for (int b = b0; b < b1; ++b) local += bcnt[b];
        uint32_t incl = local;
        const int lane = tid & 63;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - local;
        for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
        // the stretch's buckets with several words are ordered below by this thread: remember (start, count) in place as
        // start | count << 16 (n <= 4096: 13 bits each)
        for (int b = b0; b < b1; ++b) {
            const uint32_t c = bcnt[b];
            bcnt[b] = base | (c << 16);
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * T;
        if (i < n) tmp[(bcnt[bucket[k]] & 0xffffu) + rank[k]] = key[k];
    }
    __syncthreads();
    // order inside the buckets: every word counts the smaller words of its bucket (a voxel is voted while the camera looks
    // past it, so its votes CLUSTER in time: buckets of tens of words are the rule, and a serial sort of a bucket by one
    // thread made this kernel 0.6 ms; counted by their own threads, a bucket of c words costs c reads per word, in parallel)
    uint32_t final_at[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * T;
        if (i < n) {
            const uint32_t e = bcnt[bucket[k]], c = e >> 16, s0 = e & 0xffffu;
            uint32_t smaller = 0;
            if (c > 1u)
                for (uint32_t q = 0; q < c; ++q) smaller += tmp[s0 + q] < key[k] ? 1u : 0u;  // (positions are unique within a voxel)
            final_at[k] = s0 + smaller;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * T;
        if (i < n) tmp[final_at[k]] = key[k];
    }
    __syncthreads();
}

// workgroup = rank: its run in event order -> sorted_w[starts[r] .. + counts[r]) (the weights only, 4 bytes per vote); all T
// threads work all the time -- the one-by-one additions, which only one wave can do, are k_tie_add_runs' (with them in here three
// waves of four held 48 KB of LDS idle for ~5 us per run: 0.42 ms at configs[1])
// CAP: the runs of (CAP / 2, CAP] votes (CAP = 1024: of up to 1,024; CAP = kTieRunLds: also the longer ones, in windows) -- a
// workgroup of the wrong class leaves at once; three launches, so that the many short runs do not each hold 48 KB of LDS
// T threads: 256 for the shortest class, 512 for the others -- the longest's 48 KB of LDS allow three workgroups per CU whatever
// their size, so twice the threads sort a run in about two thirds of the time (same-box: k_tie_sort_runs<4096> 152 -> 99 us with
// 512 threads, 106 with 1,024; <2048> 30.8 -> 28.5 us)
template <int CAP, int T = (CAP > 1024 ? 512 : 256)>
__global__ __launch_bounds__(T) void k_tie_sort_runs(const unsigned long long* __restrict__ runs, const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ counts, unsigned pos_bits, int n_ranks,
                                                       float* __restrict__ sorted_w)
{
    __shared__ unsigned long long tmp[CAP];
    __shared__ uint32_t bcnt[CAP];
    __shared__ uint32_t wave_tot[2 * (T / 64)];
    __shared__ uint32_t s_n;
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x;
    if (r >= n_ranks) return;
    const uint32_t first = starts[r], cnt = counts[r];
    if (cnt == 0u || cnt <= (uint32_t)(CAP == 1024 ? 0 : CAP / 2) || (CAP != kTieRunLds && cnt > (uint32_t)CAP)) return;
    if (cnt <= (uint32_t)CAP) {
        tie_block_sort<CAP, T>(runs + first, (int)cnt, tmp, bcnt, wave_tot);
        for (uint32_t i = (uint32_t)tid; i < cnt; i += T) sorted_w[(size_t)first + i] = __uint_as_float((uint32_t)tmp[i]);
        return;
    }
    // Longer runs (configs[4]: ~30 k votes in a voxel): windows [lo, lo + 2^span_bits) of event positions; the run is scanned
    // once per window, the window's votes compacted into the upper half of tmp (the window is aimed at CAP / 4 votes), sorted
    // into the lower half and appended.  Positions are unique within a voxel: a window of one position holds one vote at most.
    unsigned span_bits = pos_bits;
    while (span_bits > 0 && ((unsigned long long)cnt >> (pos_bits - span_bits)) > (unsigned long long)(CAP / 4)) --span_bits;
    unsigned long long lo = 0;
    const unsigned long long pos_range = 1ull << pos_bits;
    unsigned long long* stage = tmp + CAP / 2;
    uint32_t done = 0;
    while (lo < pos_range) {
        const unsigned long long hi = lo + (1ull << span_bits);
        if (tid == 0) s_n = 0u;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < cnt; i0 += T) {
            const uint32_t i = i0 + (uint32_t)tid;
            const unsigned long long key = i < cnt ? runs[(size_t)first + i] : ~0ull;
            const unsigned long long p = key >> 32;
            if (i < cnt && p >= lo && p < hi) {
                const uint32_t at = atomicAdd(&s_n, 1u);  // (LDS; order is irrelevant here)
                if (at < (uint32_t)(CAP / 2)) stage[at] = key;
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n > (uint32_t)(CAP / 2)) {  // denser than expected here: a narrower window
            --span_bits;
            __syncthreads();
            continue;
        }
        if (n) {
            tie_block_sort<CAP, T>(stage, (int)n, tmp, bcnt, wave_tot);
            for (uint32_t i = (uint32_t)tid; i < n; i += T) sorted_w[(size_t)first + done + i] = __uint_as_float((uint32_t)tmp[i]);
            done += n;
        }
        __syncthreads();
        lo = hi;
        if (n < (uint32_t)(CAP / 16) && span_bits < pos_bits && (lo & ((1ull << (span_bits + 1)) - 1ull)) == 0ull) ++span_bits;
    }
}

// ROW of 16 lanes = (camera, contending voxel): its weights, in event order by now, added one by one in fp32 -- resetGrid
// (mapper_emvs_stereo.cpp:145), then "grid[i] += w" per vote (cartesian3dgrid.h:261-270).  A row loads 4 x 16 consecutive weights
// (the next 64 in flight meanwhile) and its lane 0 adds them in order, lane j's weight reaching it through the
// add's own DPP operand (row_shl:j): ONE vector instruction per vote step, and the four rows of a wave walk four runs with it.
// (Round 6, first version: a wave per run, v_readlane + v_add per vote -- 2.3 instructions per vote, 34 M per call at
// configs[1], 0.12 ms with eight such waves sharing a SIMD.)  A row whose run has ended, and the lanes behind a run's last
// weight, add +0: x + 0 = x for every x this sum can hold (weights >= 0, denormals kept), so padding changes no bit.
// diff[r] (optional) = |engine value - reference-order value| / max(1, |reference-order value|)
#define DSI_TIE_ADD16(SUM, W)                                                             \
    asm volatile("s_nop 4\n\t" /* (W / EXEC may have been written by the vector unit just before: DPP read hazards) */ \
                 "v_add_f32 %0, %1, %0\n\t"                                                \
                 "v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n\t"       \
                 "v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n\t"      \
                 "v_add_f32_dpp %0, %1, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n\t"      \
                 "v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n\t"      \
                 "v_add_f32_dpp %0, %1, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n\t"      \
                 "v_add_f32_dpp %0, %1, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n\t"      \
                 "v_add_f32_dpp %0, %1, %0 row_shl:15 row_mask:0xf bank_mask:0xf"            \
                 : "+v"(SUM)                                                               \
                 : "v"(W))

__global__ __launch_bounds__(256) void k_tie_add_runs(const float* __restrict__ sorted_w, const uint32_t* __restrict__ starts,
                                                      const uint32_t* __restrict__ counts, const uint32_t* __restrict__ vox, int nsv,
                                                      int n_cams, const float* __restrict__ grid0, const float* __restrict__ grid1,
                                                      float* __restrict__ exact, uint32_t* __restrict__ count, float* __restrict__ diff)
{
    const int n_ranks = nsv * n_cams;
    const int r = blockIdx.x * 16 + (int)(threadIdx.x >> 4);  // this row's run
    const unsigned j = threadIdx.x & 15u;                      // lane within the row
    const bool have_run = r < n_ranks;
    const unsigned long long first = have_run ? starts[r] : 0ull, last = have_run ? first + counts[r] : 0ull;
    float sum = 0.f;  // (only lane 0 of the row holds the run's sum)
    // 64 weights per row and turn: lane j holds weights base + 16 c + j, c = 0..3.  The loads are UNCONDITIONAL (a lane past its
    // run's end reads weight 0 of the array and keeps +0 instead): with a branch around them the compiler has to wait for
    // every load in flight before the additions, the next turn's included
    auto fetch = [&](unsigned long long at, float (&w)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned long long i = at + 16u * (unsigned)c + j;
            const float v = sorted_w[i < last ? i : 0ull];
            w[c] = i < last ? v : 0.f;
        }
    };
    if (__builtin_amdgcn_ballot_w64(first < last) == 0ull) {  // (no run with a vote in this wave: the array may be empty)
        if (j == 0 && have_run) {
            exact[r] = 0.f;
            count[r] = 0u;
            if (diff) {
                const int cam = r / nsv;
                diff[r] = fabsf((cam == 0 ? grid0 : grid1)[vox[r - cam * nsv]]);
            }
        }
        return;
    }
    float w[4];
    fetch(first, w);
    unsigned long long base = first;
    while (__builtin_amdgcn_ballot_w64(base < last) != 0ull) {  // (wave-uniform: until the longest of the four runs is through)
        float cur[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = w[c];
        fetch(base + 64, w);  // the next 64, in flight during these additions
        // one by one, in order (no reassociation; lanes and rows past their run's end add +0)
        DSI_TIE_ADD16(sum, cur[0]);
        DSI_TIE_ADD16(sum, cur[1]);
        DSI_TIE_ADD16(sum, cur[2]);
        DSI_TIE_ADD16(sum, cur[3]);
        base += 64;
    }
    if (j != 0 || !have_run) return;
    exact[r] = sum;
    count[r] = counts[r];
    if (diff) {
        const int cam = r / nsv;
        const float* grid = cam == 0 ? grid0 : grid1;
        const float have = grid[vox[r - cam * nsv]];
        diff[r] = fabsf(have - sum) / fmaxf(1.f, fabsf(sum));
    }
}
#undef DSI_TIE_ADD16

// thread = near-tie column: camera fusion of the reference-order values (fuse_op, what k_fuse2_into computes), first
// maximum over the column's contending planes (std::max_element, cartesian3dgrid.cpp:132-134), patch of the depth map
// (index -> depth: mapper_emvs_stereo.cpp:302-313).  cols[j] = (first entry in vox / exact, pixel, contenders); a column's
// contenders are contiguous there, planes ascending.  stats[2] += pixels whose plane changed
template <int OP>
__global__ __launch_bounds__(256) void k_tie_pick(const uint4* __restrict__ cols, int n_cols, const uint32_t* __restrict__ vox,
                                                  int nsv, int npix, const float* __restrict__ exact,
                                                  const uint32_t* __restrict__ count, const float* __restrict__ diff,
                                                  const float* __restrict__ planes, float* __restrict__ conf,
                                                  uint8_t* __restrict__ idx, float* __restrict__ depth,
                                                  unsigned* __restrict__ stats, float rel_gap)
{
    // stats[0] = max float bits of diff[], stats[1] = max votes of a voxel, stats[2] += changed pixels, stats[3] += columns whose
    // gap exceeds the worst-case reordering bound of their own most-voted contender (see below): one global atomic each per
    // WORKGROUP (thousands on one address serialise at ~50 ns apiece)
    __shared__ unsigned s_stats[4];
    if (threadIdx.x < 4) s_stats[threadIdx.x] = 0u;
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_cols) {
        const uint4 col = cols[j];
        float best = 0.f;
        int best_z = -1;
        unsigned dmax = 0u, cmax = 0u;
        for (unsigned i = col.x; i < col.x + col.z; ++i) {
            const float v = OP == 0 ? exact[i] : fuse_op<OP>(0.f + exact[i], exact[(size_t)nsv + i]);
            if (best_z < 0 || best < v) {
                best = v;
                best_z = (int)(vox[i] / (uint32_t)npix);
            }
            // (non-negative floats order like their bits; a NaN ranks above everything and fails the premise)
            dmax = max(dmax, __float_as_uint(diff[i]));
            cmax = max(cmax, count[i]);
            if (OP != 0) {
                dmax = max(dmax, __float_as_uint(diff[(size_t)nsv + i]));
                cmax = max(cmax, count[(size_t)nsv + i]);
            }
        }
        const uint32_t p = col.y;
        if (idx[p] != (uint8_t)best_z) atomicAdd(&s_stats[2], 1u);
        conf[p] = best;
        idx[p] = (uint8_t)best_z;
        depth[p] = planes[best_z];
        atomicMax(&s_stats[0], dmax);
        atomicMax(&s_stats[1], cmax);
        // The column's own worst case: a value summed in fp32 from n positive votes is within (n - 1) 2^-24 of the exact sum
        // (relative), the engine's value within 2^-24 + n 2^-31; the 2-ary op adds a few ulps.  With n = the most votes any of the
        // column's CONTENDERS has (either camera), a plane outside the gap could overtake the engine's best only if its error and
        // the best's together exceeded the gap: gap >= 2 (n 2^-24 + 2^-22) rules that out for planes with no more votes than n.
        // (Planes outside the gap with MORE votes than every contender are not covered: their counts are not known.  A
        // statistic beside the measured premise check, not a proof.)
        const float bound = 2.f * ((float)cmax * 5.9604644775390625e-8f + 2.384185791015625e-7f);
        if (rel_gap >= bound) atomicAdd(&s_stats[3], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_stats[0]) atomicMax(&stats[0], s_stats[0]);
        if (s_stats[1]) atomicMax(&stats[1], s_stats[1]);
        if (s_stats[2]) atomicAdd(&stats[2], s_stats[2]);
        if (s_stats[3]) atomicAdd(&stats[3], s_stats[3]);
    }
}

int grid_for(size_t work_items, int block, int max_blocks = 256 * 8)
{
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace

// 160 KB per CU, minus a little room for the kernels' static __shared__ variables
size_t max_dynamic_lds() { return 160 * 1024 - 128; }

// The dynamic-LDS limit of a kernel is a per-device function attribute: set it for the CURRENT
// device whenever this (kernel, device) pair has not been given at least `bytes` yet.  The table is
// per process; several host threads (one context each) may launch concurrently.
static hipError_t allow_dynamic_lds(const void* kern, size_t bytes)
{
    struct Entry {
        const void* kern;
        int device;
        size_t bytes;
    };
    static std::mutex mu;
    static std::vector<Entry> table;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& t : table)
        if (t.kern == kern && t.device == dev) {
            if (t.bytes >= bytes) return hipSuccess;
            if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes))
                return e;
            t.bytes = bytes;
            return hipSuccess;
        }
    if (hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)) return e;
    table.push_back({kern, dev, bytes});
    return hipSuccess;
}

// hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered per (kernel, device, block, dynamic LDS): a runtime query has
// no place on the path of every launch.  0 when the runtime cannot tell.
static int resident_blocks_cached(const void* kern, int block, size_t lds_bytes)
{
    struct Entry {
        const void* kern;
        int device, block;
        size_t bytes;
        int resident;
    };
    static std::mutex mu;
    static std::vector<Entry> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& t : table)
        if (t.kern == kern && t.device == dev && t.block == block && t.bytes == lds_bytes) return t.resident;
    int resident = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, block, lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        resident = 0;
    }
    table.push_back({kern, dev, block, lds_bytes, resident});
    return resident;
}

hipError_t launch_packet_geometry(hipStream_t s, const float* Rt, int np, const Geom& g,
                                  float* centers, float* H)
{
    if (np <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_packet_geometry, dim3((np + 63) / 64), dim3(64), 0, s, Rt, np, g,
                       centers, H);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_warp_z0(hipStream_t s, const uint16_t* ex, const uint16_t* ey,
                          const uint32_t* packet_first, int np, const float* H,
                          const float2* lut, int sensor_w, int sensor_h, float2* xy)
{
    if (np <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_warp_z0, dim3(np), dim3(256), 0, s, ex, ey, packet_first, np, H, lut,
                       sensor_w, sensor_h, xy);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_vote_global(hipStream_t s, const float2* xy, const float* centers, int np,
                              const float* planes, const Geom& g, float* dsi)
{
    if (np <= 0) return hipSuccess;
    const int zgroups = (g.nz + kVgPlanes - 1) / kVgPlanes;
    // gridDim.y is limited to 65535; x carries the packets
    hipLaunchKernelGGL(k_vote_global, dim3(np, zgroups), dim3(256), 0, s, xy, centers, planes, g,
                       dsi);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_sort_packets(hipStream_t s, const float2* xy, int np, int ny, int nz, int pad,
                               EvRec* sxy, uint32_t* nvalid, uint16_t* rowstart)
{
    if (np <= 0) return hipSuccess;
    const size_t lds = (size_t)(ny + 2 * pad + 3) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_sort_packets<false>, dim3(np), dim3(256), lds, s, xy, RawEvents{}, np, ny, nz, pad, sxy, nvalid,
                       rowstart);
    return hipExtGetLastError();
}

hipError_t launch_sort_packets_raw(hipStream_t s, const float* Rt, const uint16_t* ex, const uint16_t* ey,
                                   const uint32_t* packet_first, const float2* lut, int sensor_w, int sensor_h, const Geom& g,
                                   float* centers, int np, int pad, EvRec* sxy, uint32_t* nvalid,
                                   uint16_t* rowstart, int unit_multiplicity)
{
    if (np <= 0) return hipSuccess;
    const size_t lds = (size_t)(g.ny + 2 * pad + 3) * sizeof(uint32_t);
    RawEvents raw{Rt, ex, ey, packet_first, lut, sensor_w, sensor_h, g, centers, unit_multiplicity};
    hipLaunchKernelGGL(k_sort_packets<true>, dim3(np), dim3(256), lds, s, (const float2*)nullptr, raw, np, g.ny, g.nz, pad,
                       sxy, nvalid, rowstart);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_plane_coef(hipStream_t s, const float* centers, const float* planes,
                             const uint16_t* rowstart, uint32_t* nvalid, int np,
                             const Geom& g, const BandPlan& bp, PlaneCoef* coef, uint32_t* cuts)
{
    if (np <= 0) return hipSuccess;
    const unsigned tiles_p = (unsigned)((np + kCoefTilePackets - 1) / kCoefTilePackets);
    const size_t table_bytes = ((size_t)kCoefTilePackets * (size_t)(g.ny + 2 * bp.row_pad + 3) * sizeof(uint16_t) + 7) & ~(size_t)7;
    if (bp.cuts_inline) {
        // coefficients only (16 packets x 16 planes per block, no row tables staged), then the row tables transposed:
        // `cuts` is that table, u16 [ny + 2 row_pad + 3][bp.rs_stride]
        if (grouped(bp.packed) || bp.rs_stride < np || (bp.rs_stride & 63)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_plane_coef<false>, dim3((tiles_p * (unsigned)((g.nz + 15) / 16) + 63u) & ~63u), dim3(256), 0, s, centers,
                           planes, rowstart, nvalid, np, g, bp, coef, cuts);
        if (hipError_t e = hipExtGetLastError()) return e;
        const int nb1 = g.ny + 2 * bp.row_pad + 3;
        hipLaunchKernelGGL(k_transpose_rowstart, dim3((unsigned)(bp.rs_stride / 64), (unsigned)((nb1 + 63) / 64)), dim3(256), 0, s,
                           rowstart, np, nb1, bp.rs_stride, reinterpret_cast<uint16_t*>(cuts));
        return hipExtGetLastError();
    }
    if (!grouped(bp.packed) && table_bytes <= max_dynamic_lds()) {
        // 16 packets x 64 planes per block: the tables are loaded nz / 64 times
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_plane_coef<true>), table_bytes)) return e;
        hipLaunchKernelGGL(k_plane_coef<true>, dim3((tiles_p * (unsigned)((g.nz + 63) / 64) + 63u) & ~63u), dim3(1024), table_bytes, s,
                           centers, planes, rowstart, nvalid, np, g, bp, coef, cuts);
    } else {
        hipLaunchKernelGGL(k_plane_coef<false>, dim3((tiles_p * (unsigned)((g.nz + 15) / 16) + 63u) & ~63u), dim3(256), 0, s, centers,
                           planes, rowstart, nvalid, np, g, bp, coef, cuts);
    }
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

template <int BLOCK, int MAPPING>
static hipError_t launch_vote_bands_t(hipStream_t s, const EvRec* sxy, const PlaneCoef* coef,
                                      const uint32_t* cuts, const uint32_t* slow_any, int np,
                                      const Geom& g, const BandPlan& bp, void* out, unsigned long long* seam)
{
    constexpr bool PACKED = MAPPING != 0;
    constexpr bool VFILL = MAPPING == 5 || MAPPING == 6;
    const void* kern = VFILL ? reinterpret_cast<const void*>(&k_vote_bands_vfill<BLOCK, (VFILL ? MAPPING : 5)>)
                     : PACKED ? reinterpret_cast<const void*>(&k_vote_bands_packed<BLOCK, (PACKED && !VFILL ? MAPPING : 1)>)
                              : reinterpret_cast<const void*>(&k_vote_bands<BLOCK>);
    if (hipError_t e = allow_dynamic_lds(kern, bp.lds_bytes)) return e;
    unsigned blocks = (unsigned)item_count(bp.chunks * bp.bands, g.nz, bp.experiment == 200);
    uint32_t* counters = nullptr;
    if (PACKED && bp.persistent) {
        // as many workgroups as are resident at once: per CU, what the LDS and the 2048-thread limit allow
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned per_cu = std::max<unsigned>(1, std::min<unsigned>(2048u / BLOCK, (unsigned)(max_dynamic_lds() / std::max<size_t>(1, bp.lds_bytes))));
        unsigned resident = (unsigned)cus * per_cu;
        resident -= resident % 8;
        if (resident >= 8 && blocks > resident) {
            blocks = resident;
            counters = const_cast<uint32_t*>(slow_any) + g.nz;  // 8 counters behind the per-plane flags (zeroed by the sort kernel)
        }
    }
    if constexpr (VFILL)
        hipLaunchKernelGGL((k_vote_bands_vfill<BLOCK, (VFILL ? MAPPING : 5)>), dim3(blocks), dim3(BLOCK), bp.lds_bytes, s,
                           sxy, coef, cuts, slow_any, np, g, bp, out, seam, counters);
    else if constexpr (PACKED)
        hipLaunchKernelGGL((k_vote_bands_packed<BLOCK, (PACKED && !VFILL ? MAPPING : 1)>), dim3(blocks), dim3(BLOCK), bp.lds_bytes, s,
                           sxy, coef, cuts, slow_any, np, g, bp, out, seam, counters);
    else
        hipLaunchKernelGGL(k_vote_bands<BLOCK>, dim3(blocks), dim3(BLOCK), bp.lds_bytes, s, sxy, coef,
                           cuts, np, g, bp, out, seam);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

template <int BLOCK>
static hipError_t launch_vote_bands_b(hipStream_t s, const EvRec* sxy, const PlaneCoef* coef,
                                      const uint32_t* cuts, const uint32_t* slow_any, int np, const Geom& g,
                                      const BandPlan& bp, void* out, unsigned long long* seam)
{
    switch (bp.packed) {
    case 0: return launch_vote_bands_t<BLOCK, 0>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 1: return launch_vote_bands_t<BLOCK, 1>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 3: return launch_vote_bands_t<BLOCK, 3>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 5: return launch_vote_bands_t<BLOCK, 5>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 6: return launch_vote_bands_t<BLOCK, 6>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 7: return launch_vote_bands_t<BLOCK, 7>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 8:
        if constexpr (BLOCK == 1024) return launch_vote_bands_t<BLOCK, 8>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
        return hipErrorInvalidValue;  // (the paired cells exist for 1024-thread workgroups only)
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_vote_bands(hipStream_t s, const EvRec* sxy, const PlaneCoef* coef,
                             const uint32_t* cuts, const uint32_t* slow_any, int np, const Geom& g,
                             const BandPlan& bp, void* out, unsigned long long* seam)
{
    if (np <= 0) return hipSuccess;
    switch (bp.block_threads) {
    case 256: return launch_vote_bands_b<256>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 512: return launch_vote_bands_b<512>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    case 1024: return launch_vote_bands_b<1024>(s, sxy, coef, cuts, slow_any, np, g, bp, out, seam);
    default: return hipErrorInvalidValue;
    }
}

int fused_max_pairs() { return 1024 * kSplitSlice; }

hipError_t launch_prepare_cameras(hipStream_t s, const PrepCameraArgs* args, int n, const Geom& g, const BandPlan& bp)
{
    if (n < 1 || n > kFusedMaxCameras) return hipErrorInvalidValue;
    const int n_pairs = bp.bands * g.nz;
    PrepCameras cams{};
    cams.n = n;
    for (int c = 0; c < n; ++c) {
        const PrepCameraArgs& a = args[c];
        cams.cam[c].raw = RawEvents{a.Rt, a.ex, a.ey, a.packet_first, a.lut, a.sensor_w, a.sensor_h, a.g, a.centers, 0};
        cams.cam[c].np = a.np;
        cams.cam[c].sxy = a.sxy;
        cams.cam[c].nvalid = a.nvalid;
        cams.cam[c].rowstart = a.rowstart;
        cams.cam[c].planes = a.planes;
        cams.cam[c].coef = a.coef;
        cams.cam[c].cuts = a.cuts;
        cams.cam[c].pair_work = a.pair_work;
    }
    unsigned total_np = 0;
    for (int c = 0; c < cams.n; ++c) total_np += (unsigned)cams.cam[c].np;
    if (total_np == 0) return hipSuccess;
    {
        const size_t lds = (size_t)(g.ny + 2 * bp.row_pad + 3) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_sort_packets_multi, dim3(total_np), dim3(256), lds, s, cams, g.ny, g.nz, bp.row_pad, n_pairs);
        if (hipError_t e = hipExtGetLastError()) return e;
    }
    const size_t table_bytes = ((size_t)kCoefTilePackets * (size_t)(g.ny + 2 * bp.row_pad + 3) * sizeof(uint16_t) + 7) & ~(size_t)7;
    const bool staged = table_bytes <= max_dynamic_lds();
    const unsigned zdiv = staged ? 64u : 16u;
    PrepBlocks pb{};
    unsigned total_blocks = 0;
    for (int c = 0; c < cams.n; ++c) {
        const unsigned tiles_p = (unsigned)((cams.cam[c].np + kCoefTilePackets - 1) / kCoefTilePackets);
        pb.n[c] = (tiles_p * (((unsigned)g.nz + zdiv - 1) / zdiv) + 63u) & ~63u;
        total_blocks += pb.n[c];
    }
    if (staged) {
        if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_plane_coef_multi<true>), table_bytes)) return e;
        hipLaunchKernelGGL(k_plane_coef_multi<true>, dim3(total_blocks), dim3(1024), table_bytes, s, cams, bp, pb);
    } else {
        hipLaunchKernelGGL(k_plane_coef_multi<false>, dim3(total_blocks), dim3(256), 0, s, cams, bp, pb);
    }
    return hipExtGetLastError();
}

hipError_t launch_fused_splits(hipStream_t s, const uint32_t* work0, const uint32_t* work1, int n_pairs, uint32_t fixed_per_pair,
                               unsigned long long* prefix, uint32_t* splits)
{
    if (n_pairs > 1024 * kSplitSlice) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fused_splits, dim3(1), dim3(1024), 0, s, work0, work1, n_pairs, fixed_per_pair, fused_grid_blocks(), prefix,
                       splits);
    return hipExtGetLastError();
}

template <int MAPPING>
static hipError_t launch_vote_fuse_argmax_t(hipStream_t s, const FusedCameras& cams, const Geom& g, const BandPlan& bp,
                                            int op, const uint32_t* splits, unsigned blocks, unsigned long long* keys,
                                            unsigned long long* trace)
{
    constexpr int CELLS = fused_cells_per_thread(MAPPING);
    if ((size_t)(bp.band_rows + 2) * g.nx > (size_t)CELLS * 1024) return hipErrorInvalidValue;
    if (cams.n == 4) {
        // four cameras: the geometric-mean tree only (op 3), one workgroup per CU, a second register array per thread -- 16
        // cells per thread in every mapping (plan_fused cuts the bands accordingly)
        constexpr int CELLS4 = kFusedCellsFourCameras;
        if (op != 3 || (size_t)(bp.band_rows + 2) * g.nx > (size_t)CELLS4 * 1024) return hipErrorInvalidValue;
        const void* kern4 = reinterpret_cast<const void*>(&k_vote_fuse_argmax<MAPPING, CELLS4, false, true>);
        if (hipError_t e = allow_dynamic_lds(kern4, bp.lds_bytes)) return e;
        hipLaunchKernelGGL((k_vote_fuse_argmax<MAPPING, CELLS4, false, true>), dim3(blocks), dim3(1024), bp.lds_bytes, s, cams, g, bp, op,
                           splits, keys, trace);
        return hipExtGetLastError();
    }
    if constexpr (MAPPING == 1 || MAPPING == 3) {
        // a band of at most half the LDS and half the cells: two workgroups per CU, the variant with <= 64 VGPRs
        constexpr int HALF = kFusedCellsTwoPerCu;
        // (bp.experiment 300: experiments flavour, DSI_FUSED_2CU=0 -- the same small band with ONE workgroup per CU, for A/B)
        if (bp.experiment != 300 && bp.lds_bytes * 2 <= max_dynamic_lds() && (size_t)(bp.band_rows + 2) * g.nx <= (size_t)HALF * 1024) {
            const void* kern2 = reinterpret_cast<const void*>(&k_vote_fuse_argmax_2cu<MAPPING, HALF>);
            if (hipError_t e = allow_dynamic_lds(kern2, bp.lds_bytes)) return e;
            // ... only if the runtime agrees that two fit (static LDS and the allocation granularity also count: a band just
            // under half the LDS would otherwise run its 2 x grid of half-size assignments in two serialised waves)
            // (the answer depends on (kernel, device, LDS bytes) only: asked once, not on every window -- ADVICE r05)
            if (resident_blocks_cached(kern2, 1024, bp.lds_bytes) < 2) goto one_per_cu;
            // (the balanced partition, an experiments-flavour option, is laid out for one workgroup per CU: not used here)
            hipLaunchKernelGGL((k_vote_fuse_argmax_2cu<MAPPING, HALF>), dim3(2 * blocks), dim3(1024), bp.lds_bytes, s, cams, g, bp, op,
                               nullptr, keys, trace);
            return hipExtGetLastError();
        }
    }
one_per_cu:
    if constexpr (MAPPING == 1) {
        // two cameras, the packed stream: the instantiation that defers camera 1's fusion + arg-max update into the waits
        // of the next phase (vote_fuse_argmax_body, DEFER).  bp.experiment 301 (experiments flavour): off, for A/B runs
        if (cams.n == 2 && bp.experiment != 301) {
            const void* kern_d = reinterpret_cast<const void*>(&k_vote_fuse_argmax<MAPPING, CELLS, true>);
            if (hipError_t e = allow_dynamic_lds(kern_d, bp.lds_bytes)) return e;
            hipLaunchKernelGGL((k_vote_fuse_argmax<MAPPING, CELLS, true>), dim3(blocks), dim3(1024), bp.lds_bytes, s, cams, g, bp, op, splits,
                               keys, trace);
            return hipExtGetLastError();
        }
    }
    const void* kern = reinterpret_cast<const void*>(&k_vote_fuse_argmax<MAPPING, CELLS>);
    if (hipError_t e = allow_dynamic_lds(kern, bp.lds_bytes)) return e;
    hipLaunchKernelGGL((k_vote_fuse_argmax<MAPPING, CELLS>), dim3(blocks), dim3(1024), bp.lds_bytes, s, cams, g, bp, op, splits, keys, trace);
    return hipExtGetLastError();
}

size_t fused_max_cells(int mapping, int n_cameras)
{
    return (size_t)(n_cameras == 4 ? std::min(kFusedCellsFourCameras, fused_cells_per_thread(mapping)) : fused_cells_per_thread(mapping)) * 1024;
}

size_t fused_trace_words() { return (size_t)2 * fused_grid_blocks() * kFusedTracePhases * 16 * 4; }  // (two workgroups per CU at most)

int fused_grid_blocks()
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return std::max(8, cus - cus % 8);  // one 1024-thread workgroup per CU, whole groups of 8 (XCDs)
}

hipError_t launch_vote_fuse_argmax(hipStream_t s, const FusedCameras& cams, const Geom& g, const BandPlan& bp, int op,
                                   const uint32_t* splits, unsigned long long* keys, unsigned long long* trace)
{
    if (cams.n < 1 || cams.n > kFusedMaxCameras || bp.block_threads != 1024 || bp.chunks != 1 || !bp.halo) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)fused_grid_blocks();
    switch (bp.packed) {
    case 1: return launch_vote_fuse_argmax_t<1>(s, cams, g, bp, op, splits, blocks, keys, trace);
    case 3: return launch_vote_fuse_argmax_t<3>(s, cams, g, bp, op, splits, blocks, keys, trace);
    case 5: return launch_vote_fuse_argmax_t<5>(s, cams, g, bp, op, splits, blocks, keys, trace);
    case 6: return launch_vote_fuse_argmax_t<6>(s, cams, g, bp, op, splits, blocks, keys, trace);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sort_groups(hipStream_t s, const float2* xy, int np, int S, int ny, int nz, int pad,
                              EvRec* sxy, uint8_t* spk, uint32_t* nvalid, uint16_t* rowstart)
{
    if (np <= 0) return hipSuccess;
    const int ngroups = (np + S - 1) / S;
    const size_t lds = (size_t)(ny + 2 * pad + 3) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_sort_groups, dim3(ngroups), dim3(256), lds, s, xy, np, S, ny, nz, pad, sxy, spk,
                       nvalid, rowstart);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_group_cuts(hipStream_t s, const uint32_t* prow, const uint16_t* rowstart, int np,
                             int S, const Geom& g, const BandPlan& bp, uint32_t* gcuts)
{
    if (np <= 0) return hipSuccess;
    const int ngroups = (np + S - 1) / S;
    const size_t total = (size_t)ngroups * g.nz * bp.bands;
    hipLaunchKernelGGL(k_group_cuts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, prow,
                       rowstart, np, ngroups, S, g.nz, bp.bands, g.ny + 2 * bp.row_pad + 2, gcuts);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

template <int BLOCK>
static hipError_t launch_vote_groups_t(hipStream_t s, const EvRec* sxy, const uint8_t* spk,
                                       const PlaneCoef* coef, const uint32_t* gcuts,
                                       const uint32_t* slow_any, int np, int S, const Geom& g,
                                       const BandPlan& bp, void* out, unsigned long long* seam)
{
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_vote_groups<BLOCK>), bp.lds_bytes))
        return e;
    const int ngroups = (np + S - 1) / S;
    const unsigned blocks = (unsigned)item_count(bp.chunks * bp.bands, g.nz, bp.experiment == 200);
    hipLaunchKernelGGL(k_vote_groups<BLOCK>, dim3(blocks), dim3(BLOCK), bp.lds_bytes, s, sxy, spk,
                       coef, gcuts, slow_any, np, ngroups, S, g, bp, out, seam);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_vote_groups(hipStream_t s, const EvRec* sxy, const uint8_t* spk,
                              const PlaneCoef* coef, const uint32_t* gcuts, const uint32_t* slow_any,
                              int np, int S, const Geom& g, const BandPlan& bp, void* out, unsigned long long* seam)
{
    if (np <= 0) return hipSuccess;
    switch (bp.block_threads) {
    case 256: return launch_vote_groups_t<256>(s, sxy, spk, coef, gcuts, slow_any, np, S, g, bp, out, seam);
    case 512: return launch_vote_groups_t<512>(s, sxy, spk, coef, gcuts, slow_any, np, S, g, bp, out, seam);
    case 1024: return launch_vote_groups_t<1024>(s, sxy, spk, coef, gcuts, slow_any, np, S, g, bp, out, seam);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_seam_rows(hipStream_t s, const unsigned long long* seam, int chunks, const Geom& g,
                            const BandPlan& bp, void* out)
{
    if (bp.bands < 2) return hipSuccess;
    const size_t vol = partial_stride((size_t)g.nx * g.ny * g.nz);
    const dim3 grid((unsigned)g.nz * (unsigned)(bp.bands - 1), (unsigned)((g.nx + 511) / 512), (unsigned)chunks);
    if (bp.raw_out)
        hipLaunchKernelGGL(k_seam_rows<true>, grid, dim3(256), 0, s, seam, g, bp.bands, bp.band_rows, out, vol);
    else
        hipLaunchKernelGGL(k_seam_rows<false>, grid, dim3(256), 0, s, seam, g, bp.bands, bp.band_rows, out, vol);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_reduce_partials(hipStream_t s, const unsigned long long* partials, int chunks, size_t n,
                                  float* dsi, int accumulate, const unsigned long long* seam, const Geom* g, const BandPlan* bp)
{
    // the seam rows can be folded in when two voxels never straddle a row and 32-bit voxel indices suffice
    const bool fold = seam && g && bp && bp->bands >= 2 && (g->nx & 1) == 0 && n <= 0xffffffffull;
    if (seam && !fold && bp && bp->bands >= 2) return hipErrorInvalidValue;  // the caller must run launch_seam_rows instead
    hipLaunchKernelGGL(k_reduce_partials, dim3(grid_for(n / 2 + 1, 256)), dim3(256), 0, s,
                       partials, chunks, n, dsi, accumulate, fold ? seam : nullptr, g ? g->nx : 0, g ? g->ny : 0, g ? g->nz : 0,
                       bp ? bp->bands : 0, bp ? bp->band_rows : 0);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_fuse2(hipStream_t s, float* a, const float* g, size_t n, int op)
{
    const dim3 grid(grid_for(n / 4 + 1, 256)), block(256);
    switch (op) {
    case 1: hipLaunchKernelGGL(k_fuse2<1>, grid, block, 0, s, a, g, n); break;
    case 2: hipLaunchKernelGGL(k_fuse2<2>, grid, block, 0, s, a, g, n); break;
    case 3: hipLaunchKernelGGL(k_fuse2<3>, grid, block, 0, s, a, g, n); break;
    case 4: hipLaunchKernelGGL(k_fuse2<4>, grid, block, 0, s, a, g, n); break;
    case 5: hipLaunchKernelGGL(k_fuse2<5>, grid, block, 0, s, a, g, n); break;
    case 6: hipLaunchKernelGGL(k_fuse2<6>, grid, block, 0, s, a, g, n); break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_fuse2_into(hipStream_t s, float* dst, const float* a, const float* g, size_t n, int op)
{
    const dim3 grid(grid_for(n / 4 + 1, 256)), block(256);
    switch (op) {
    case 1: hipLaunchKernelGGL(k_fuse2_into<1>, grid, block, 0, s, dst, a, g, n); break;
    case 2: hipLaunchKernelGGL(k_fuse2_into<2>, grid, block, 0, s, dst, a, g, n); break;
    case 3: hipLaunchKernelGGL(k_fuse2_into<3>, grid, block, 0, s, dst, a, g, n); break;
    case 4: hipLaunchKernelGGL(k_fuse2_into<4>, grid, block, 0, s, dst, a, g, n); break;
    case 5: hipLaunchKernelGGL(k_fuse2_into<5>, grid, block, 0, s, dst, a, g, n); break;
    case 6: hipLaunchKernelGGL(k_fuse2_into<6>, grid, block, 0, s, dst, a, g, n); break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_fuse_hm_n(hipStream_t s, float* a, const float* g, size_t n, int n_maps)
{
    hipLaunchKernelGGL(k_elementwise<EW_HM_N>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, s, a,
                       g, n, (float)n_maps, (float)(n_maps - 1));
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_accumulate(hipStream_t s, float* acc, const float* g, size_t n, int mode)
{
    const dim3 grid(grid_for(n / 4 + 1, 256)), block(256);
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_elementwise<EW_ADD>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    case 1: hipLaunchKernelGGL(k_elementwise<EW_ADD_INV>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    case 2: hipLaunchKernelGGL(k_elementwise<EW_ADD_LOG>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    case 3: hipLaunchKernelGGL(k_elementwise<EW_ADD_SQ>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    case 4: hipLaunchKernelGGL(k_elementwise<EW_MIN>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    case 5: hipLaunchKernelGGL(k_elementwise<EW_MAX>, grid, block, 0, s, acc, g, n, 0.f, 0.f); break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_finalize(hipStream_t s, float* acc, size_t n, int mode, int n_maps)
{
    const dim3 grid(grid_for(n / 4 + 1, 256)), block(256);
    const float* none = nullptr;
    const float fn = (float)n_maps;
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_elementwise<EW_FIN_AM>, grid, block, 0, s, acc, none, n, fn, 0.f); break;
    case 1: hipLaunchKernelGGL(k_elementwise<EW_FIN_HM>, grid, block, 0, s, acc, none, n, fn, 0.f); break;
    case 2: hipLaunchKernelGGL(k_elementwise<EW_FIN_GM>, grid, block, 0, s, acc, none, n, fn, 0.f); break;
    case 3: hipLaunchKernelGGL(k_elementwise<EW_FIN_RMS>, grid, block, 0, s, acc, none, n, fn, 0.f); break;
    case 4:
    case 5: break;  // min / max need no finalisation
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

// plane-sharded arg-max: (confidence, local index) -> one 64-bit key per pixel whose MAX over the
// shards is the unsharded collapseMaxZSlice (cartesian3dgrid.cpp:115-137: larger value wins, the
// smaller plane index on ties).  DSI values are >= 0 (and -0.0 never occurs: sums of >= 0 terms),
// so their bit patterns order like the floats.
__global__ __launch_bounds__(256) void k_pack_argmax(const float* __restrict__ conf,
                                                     const uint8_t* __restrict__ idx, int n,
                                                     int plane_begin,
                                                     unsigned long long* __restrict__ keys, int combine)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long bits = __float_as_uint(conf[i]);
    const unsigned long long key = (bits << 8) | (unsigned long long)(255 - ((int)idx[i] + plane_begin));
    // combine: a second plane range of the same rank joins the keys of the first
    keys[i] = (combine && keys[i] > key) ? keys[i] : key;
}

__global__ __launch_bounds__(256) void k_unpack_argmax(unsigned long long* __restrict__ keys,
                                                       int n, const float* __restrict__ planes_full,
                                                       float* __restrict__ conf,
                                                       uint8_t* __restrict__ idx,
                                                       float* __restrict__ depth, int clear)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (clear: the buffer is the fused kernel's -- n keys + kFusedKeyTail words, the tail = its pair-dealing counters)
    if (clear && i < 2 * kFusedKeyTail) reinterpret_cast<unsigned*>(keys + n)[i] = 0u;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    if (clear) keys[i] = 0ull;  // ready for the next fused vote (saves a memset launch per window)
    const int gi = 255 - (int)(k & 255ull);
    conf[i] = __uint_as_float((uint32_t)(k >> 8));
    idx[i] = (uint8_t)gi;
    depth[i] = planes_full[gi];  // mapper_emvs_stereo.cpp:302-313 over the full depth vector
}

hipError_t launch_pack_argmax(hipStream_t s, const float* conf, const uint8_t* idx, int n, int plane_begin,
                              unsigned long long* keys, int combine)
{
    hipLaunchKernelGGL(k_pack_argmax, dim3((n + 255) / 256), dim3(256), 0, s, conf, idx, n, plane_begin, keys, combine);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_unpack_argmax(hipStream_t s, unsigned long long* keys, int n, const float* planes_full,
                                float* conf, uint8_t* idx, float* depth, int clear)
{
    hipLaunchKernelGGL(k_unpack_argmax, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, planes_full, conf, idx,
                       depth, clear);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_fuse_n(hipStream_t s, float* dst, const float* const* srcs, int n_src, size_t n, int mode)
{
    if (n_src < 1 || n_src > kMaxFuseSources) return hipErrorInvalidValue;
    FuseSources fs{};
    for (int c = 0; c < n_src; ++c) fs.p[c] = srcs[c];
    const dim3 grid(grid_for(n / 4 + 1, 256)), block(256);
    const float fn = (float)n_src;
    const float inf = __builtin_inff();
    switch (mode) {
    case 0: hipLaunchKernelGGL((k_fuse_n<EW_ADD, EW_FIN_AM>), grid, block, 0, s, dst, fs, n_src, n, 0.f, fn); break;
    case 1: hipLaunchKernelGGL((k_fuse_n<EW_ADD_INV, EW_FIN_HM>), grid, block, 0, s, dst, fs, n_src, n, 0.f, fn); break;
    case 2: hipLaunchKernelGGL((k_fuse_n<EW_ADD_LOG, EW_FIN_GM>), grid, block, 0, s, dst, fs, n_src, n, 0.f, fn); break;
    case 3: hipLaunchKernelGGL((k_fuse_n<EW_ADD_SQ, EW_FIN_RMS>), grid, block, 0, s, dst, fs, n_src, n, 0.f, fn); break;
    case 4: hipLaunchKernelGGL((k_fuse_n<EW_MIN, -1>), grid, block, 0, s, dst, fs, n_src, n, inf, fn); break;
    case 5: hipLaunchKernelGGL((k_fuse_n<EW_MAX, -1>), grid, block, 0, s, dst, fs, n_src, n, -inf, fn); break;
    case 6:  // the tree of 2-ary geometric means: n = 2, 4, 8
        if (n_src == 2) hipLaunchKernelGGL(k_fuse_gm_tree<2>, grid, block, 0, s, dst, fs, n);
        else if (n_src == 4) hipLaunchKernelGGL(k_fuse_gm_tree<4>, grid, block, 0, s, dst, fs, n);
        else if (n_src == 8) hipLaunchKernelGGL(k_fuse_gm_tree<8>, grid, block, 0, s, dst, fs, n);
        else return hipErrorInvalidValue;
        break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();
}

// identity element of an accumulate mode: 0 for the sums, +inf for min, -inf for max
__global__ __launch_bounds__(256) void k_fill(float* __restrict__ a, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

// depth map -> page-locked host memory by the CUs (the three arrays of a fetch in one launch): see launch_store_depth_map
__global__ __launch_bounds__(256) void k_store_depth_map(const float* __restrict__ depth, const float* __restrict__ conf,
                                                         const uint8_t* __restrict__ idx, size_t npix, float* __restrict__ depth_h,
                                                         float* __restrict__ conf_h, uint8_t* __restrict__ idx_h)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        if (depth_h) depth_h[i] = depth[i];
        if (conf_h) conf_h[i] = conf[i];
    }
    if (idx_h) {
        const size_t n4 = npix / 4;  // (hipMalloc'ed and hipHostMalloc'ed arrays: 4-byte aligned)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(idx);
        uint32_t* dst = reinterpret_cast<uint32_t*>(idx_h);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
        if (blockIdx.x == 0 && threadIdx.x < npix - 4 * n4) idx_h[4 * n4 + threadIdx.x] = idx[4 * n4 + threadIdx.x];
    }
}

// A device -> host copy queued on an SDMA engine BEHIND a kernel holds up every later copy of that engine -- the next
// window's uploads waited for the current window's kernel (profiles/r05_cpp_window_stream_timeline.txt).  A kernel that
// stores the maps into mapped page-locked memory keeps the copy engines free for the uploads.
hipError_t launch_store_depth_map(hipStream_t s, const float* depth, const float* conf, const uint8_t* idx, size_t npix,
                                  float* depth_host_dev, float* conf_host_dev, uint8_t* idx_host_dev)
{
    if (!npix) return hipSuccess;
    // (PCIe-bound: a few CUs do, the others are the next window's)
    hipLaunchKernelGGL(k_store_depth_map, dim3(grid_for(npix, 256, 64)), dim3(256), 0, s, depth, conf, idx, npix, depth_host_dev,
                       conf_host_dev, idx_host_dev);
    return hipExtGetLastError();
}

hipError_t launch_fill(hipStream_t s, float* a, size_t n, float v)
{
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, 256)), dim3(256), 0, s, a, n, v);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_collapse_max_z(hipStream_t s, const float* dsi, int nx, int ny, int nz,
                                 float* conf, uint8_t* idx, const float* planes, float* depth)
{
    const int npix = nx * ny;
    hipLaunchKernelGGL(k_collapse_max_z, dim3((npix + 255) / 256), dim3(256), 0, s, dsi, npix, nz,
                       conf, idx, planes, depth);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_collapse_max_z_fused(hipStream_t s, const float* a, const float* b, int nx, int ny, int nz, int op,
                                       float* conf, uint8_t* idx, const float* planes, float* depth)
{
    const int npix = nx * ny;
    const dim3 grid((npix + 255) / 256), block(256);
    switch (op) {
    case 1: hipLaunchKernelGGL(k_collapse_max_z_fused<1>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    case 2: hipLaunchKernelGGL(k_collapse_max_z_fused<2>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    case 3: hipLaunchKernelGGL(k_collapse_max_z_fused<3>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    case 4: hipLaunchKernelGGL(k_collapse_max_z_fused<4>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    case 5: hipLaunchKernelGGL(k_collapse_max_z_fused<5>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    case 6: hipLaunchKernelGGL(k_collapse_max_z_fused<6>, grid, block, 0, s, a, b, npix, nz, conf, idx, planes, depth); break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();
}

hipError_t launch_collapse_max_z_fused_n(hipStream_t s, const float* const* srcs, int n_src, int mode, int nx, int ny,
                                         int nz, float* conf, uint8_t* idx, const float* planes, float* depth)
{
    if (n_src < 1 || n_src > kMaxFuseSources) return hipErrorInvalidValue;
    FuseSources fs{};
    for (int c = 0; c < n_src; ++c) fs.p[c] = srcs[c];
    const int npix = nx * ny;
    const dim3 grid((npix + 255) / 256), block(256);
    const float fn = (float)n_src;
    const float inf = __builtin_inff();
    switch (mode) {
    case 0: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_ADD, EW_FIN_AM>), grid, block, 0, s, fs, n_src, npix, nz, 0.f, fn, conf, idx, planes, depth); break;
    case 1: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_ADD_INV, EW_FIN_HM>), grid, block, 0, s, fs, n_src, npix, nz, 0.f, fn, conf, idx, planes, depth); break;
    case 2: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_ADD_LOG, EW_FIN_GM>), grid, block, 0, s, fs, n_src, npix, nz, 0.f, fn, conf, idx, planes, depth); break;
    case 3: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_ADD_SQ, EW_FIN_RMS>), grid, block, 0, s, fs, n_src, npix, nz, 0.f, fn, conf, idx, planes, depth); break;
    case 4: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_MIN, -1>), grid, block, 0, s, fs, n_src, npix, nz, inf, fn, conf, idx, planes, depth); break;
    case 5: hipLaunchKernelGGL((k_collapse_max_z_fused_n<EW_MAX, -1>), grid, block, 0, s, fs, n_src, npix, nz, -inf, fn, conf, idx, planes, depth); break;
    case 6:
        if (n_src == 2) hipLaunchKernelGGL(k_collapse_max_z_gm_tree<2>, grid, block, 0, s, fs, npix, nz, conf, idx, planes, depth);
        else if (n_src == 4) hipLaunchKernelGGL(k_collapse_max_z_gm_tree<4>, grid, block, 0, s, fs, npix, nz, conf, idx, planes, depth);
        else if (n_src == 8) hipLaunchKernelGGL(k_collapse_max_z_gm_tree<8>, grid, block, 0, s, fs, npix, nz, conf, idx, planes, depth);
        else return hipErrorInvalidValue;
        break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();
}

hipError_t launch_mean_square(hipStream_t s, const float* dsi, size_t n, double* accum)
{
    hipLaunchKernelGGL(k_mean_square, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, dsi, n,
                       accum);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_depth_map_filters(hipStream_t s, float* conf, const uint8_t* idx, int nx, int ny,
                                    int ksize, double C, int median_size, double max_confidence,
                                    const float* planes, uint32_t* minmax_scratch, uint8_t* conf8,
                                    uint8_t* mask, uint8_t* idx_filtered, float* depth)
{
    const int n = nx * ny;
    const uint32_t init[2] = {0xffffffffu, 0u};
    hipError_t e = hipMemcpyAsync(minmax_scratch, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conf_minmax, dim3(grid_for(n, 256, 512)), dim3(256), 0, s, conf, n,
                       (float)max_confidence, minmax_scratch);
    hipLaunchKernelGGL(k_conf8, dim3((n + 255) / 256), dim3(256), 0, s, conf, n, minmax_scratch, conf8);
    GaussK kern{};
    if (ksize > 63) ksize = 63;
    // cv::getGaussianKernel, sigma <= 0: fixed tables up to 7, else sigma from the size
    static const float k1[] = {1.f};
    static const float k3[] = {0.25f, 0.5f, 0.25f};
    static const float k5[] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
    static const float k7[] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    const float* fixed = ksize == 1 ? k1 : ksize == 3 ? k3 : ksize == 5 ? k5 : ksize == 7 ? k7 : nullptr;
    if (fixed) {
        for (int i = 0; i < ksize; ++i) kern.w[i] = fixed[i];
    } else {
        const double sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
        const double scale2x = -0.5 / (sigma * sigma);
        double sum = 0, t[64];
        for (int i = 0; i < ksize; ++i) {
            const double x = i - (ksize - 1) * 0.5;
            t[i] = exp(scale2x * x * x);
            sum += t[i];
        }
        for (int i = 0; i < ksize; ++i) kern.w[i] = (float)(t[i] * (1. / sum));
    }
    const dim3 grid2((nx + 63) / 64, (ny + 3) / 4);
    hipLaunchKernelGGL(k_adaptive_mask, grid2, dim3(256), 0, s, conf8, nx, ny, ksize, kern,
                       (int)ceil(-C), mask);
    hipLaunchKernelGGL(k_masked_median, grid2, dim3(256), 0, s, idx, mask, nx, ny, median_size / 2,
                       idx_filtered);
    const int border = ksize / 2 > 1 ? ksize / 2 : 1;
    hipLaunchKernelGGL(k_finish_depth, dim3((n + 255) / 256), dim3(256), 0, s, mask, idx_filtered, nx,
                       ny, border, planes, depth);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

hipError_t launch_div_probe(hipStream_t s, const float* n, const float* d, size_t count, float* q,
                            float* ref)
{
    hipLaunchKernelGGL(k_div_probe, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, n, d,
                       count, q, ref);
    return hipExtGetLastError();  // status of THIS launch (hipGetLastError is sticky across calls)
}

// ---- the resolver's premise as a per-column PROOF (dsi_mapper_prove_near_ties) ----
// The resolver re-sums the planes within rel_gap of a column's maximum and trusts that no plane outside can win under the
// reference's arithmetic.  That is provable once the number n of votes of every voxel is known: the reference adds a voxel's n
// non-negative weights one by one in fp32, so its value R lies within gamma_(n-1) = (n-1) u / (1 - (n-1) u), u = 2^-24, of their
// real sum (the standard bound for recursive summation), and the engine's value E is that sum with every weight rounded to 2^-31
// and ONE rounding to fp32.  The engine keeps no counts; this pass makes them:
//   k_count_votes   H[z][yi][xi] += 1 for every vote the reference casts (transfer with the IEEE divide, accept test of
//                   cartesian3dgrid.h:255-259) -- the integer location (xi, yi) of the vote's 2 x 2 footprint; a voxel's n is
//                   the sum of the four H cells whose footprint contains it.  Global atomics on a u32 volume: a verification
//                   pass, not a product kernel (~1 G atomics per camera at configs[1]).
//   k_tie_prove     thread = column: the engine's first maximum, the resolver's threshold, and for every plane BELOW the
//                   threshold an upper bound of its reference value against a lower bound of the maximum's.
// block = (packet, group of kVgPlanes planes); thread t owns events t, t + 256, ... of the packet (as k_vote_global)
__global__ __launch_bounds__(256) void k_count_votes(TieEvents ev, const float* __restrict__ centers, const float* __restrict__ planes,
                                                     Geom g, uint32_t* __restrict__ H)
{
    const int k = blockIdx.x;
    const int zbeg = blockIdx.y * kVgPlanes, zend = min(g.nz, zbeg + kVgPlanes);
    const float cx_ = centers[3 * k], cy_ = centers[3 * k + 1], cz_ = centers[3 * k + 2];
    const size_t first = ev.first ? (size_t)ev.first[k] : (size_t)k * kPacket;
    float hh[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) hh[i] = ev.H[9 * (size_t)k + i];
    float2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t at = first + threadIdx.x + 256 * i;
        e[i] = warp_event_z0(ev.x[at], ev.y[at], hh, ev.lut, ev.sensor_w, ev.sensor_h);  // mapper_emvs_stereo.cpp:129-142
    }
    const float xmax = (float)(g.nx - 1), ymax = (float)(g.ny - 1);
    const size_t plane_sz = (size_t)g.nx * g.ny;
    for (int z = zbeg; z < zend; ++z) {
        float a, bx, by, d;
        plane_coefficients(cx_, cy_, cz_, planes[z], g, a, bx, by, d);
        uint32_t* plane = H + (size_t)z * plane_sz;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float X = (e[i].x * a + bx) / d;  // :194
            const float Y = (e[i].y * a + by) / d;  // :195
            if (X >= 0.f && Y >= 0.f && X < xmax && Y < ymax) atomicAdd(plane + (size_t)(int)Y * g.nx + (int)X, 1u);  // (see vote_global)
        }
    }
}

hipError_t launch_count_votes(hipStream_t s, const uint16_t* ex, const uint16_t* ey, const uint32_t* packet_first, const float* H9,
                              const float2* lut, int sensor_w, int sensor_h, const float* centers, const float* planes, const Geom& g,
                              int np, uint32_t* H)
{
    if (np <= 0) return hipSuccess;
    const TieEvents ev{ex, ey, packet_first, H9, lut, sensor_w, sensor_h};
    hipLaunchKernelGGL(k_count_votes, dim3(np, (g.nz + kVgPlanes - 1) / kVgPlanes), dim3(256), 0, s, ev, centers, planes, g, H);
    return hipExtGetLastError();
}

// [lo, hi] of the value the REFERENCE holds in a voxel whose engine value is E (exact sum of the weights truncated to the 2^-31
// grid, rounded once to fp32: the LDS-band mappings other than the paired one) and which received n votes.  n u >= 1/2: no bound
// (hi = +inf).
__device__ __forceinline__ void tie_reference_interval(float E, uint32_t n, double* lo, double* hi)
{
    if (n == 0u) {  // no vote: exactly zero in either arithmetic
        *lo = *hi = 0.0;
        return;
    }
    // u = 2^-24; q = n 2^-31: the voting kernels TRUNCATE a weight to the 2^-31 grid (v_cvt_u32_f32), so the sum of the real
    // weights lies between the engine's integer sum and that sum + n 2^-31 (taken on both sides here)
    const double u = 5.9604644775390625e-8, q = (double)n * 4.6566128730773926e-10;
    const double nu = (double)(n - 1u) * u;
    if (!(nu < 0.5)) {
        *lo = 0.0;
        *hi = __builtin_inf();
        return;
    }
    const double gamma = nu / (1.0 - nu);
    const double w_hi = (double)E * (1.0 + 2.0 * u) + q, w_lo = fmax(0.0, (double)E * (1.0 - 2.0 * u) - q);  // the weights' real sum
    *hi = w_hi * (1.0 + gamma);
    *lo = fmax(0.0, w_lo * (1.0 - gamma));
    // E > 0 means the truncated weights sum to at least 2^-31, so ONE weight is at least 2^-31 -- and a sequential fp32 sum of
    // non-negative terms is never below any of them (fl(s + w) >= max(s, w)): the reference's value is not zero either.  Without
    // this floor a column whose maximum is a handful of tiny weights could not even exclude its all-zero planes.
    if (E > 0.f) *lo = fmax(*lo, 1.1641532182693481e-10);  // 2^-33
}

// the fusion op as a real function (monotone non-decreasing in both arguments on [0, inf)); the fp32 chains of fuse_op stay
// within 8 u of it (at most five roundings)
template <int OP>
__device__ __forceinline__ double tie_fuse_real(double a, double g)
{
    if (OP == 0) return a;
    if (OP == 1) return fmin(a, g);
    if (OP == 2) return 2.0 * a * g / (a + g + (double)0.1f);
    if (OP == 3) return sqrt(a * g);
    if (OP == 4) return 0.5 * (a + g);
    if (OP == 5) return sqrt(0.5 * (a * a + g * g));
    return fmax(a, g);
}

__device__ __forceinline__ uint32_t tie_votes_of(const uint32_t* __restrict__ Hz, int x, int y, int nx)
{
    // the votes whose 2 x 2 footprint (xi .. xi + 1, yi .. yi + 1) contains (x, y): integer locations (x - 1 .. x, y - 1 .. y)
    uint32_t n = Hz[(size_t)y * nx + x];
    if (x > 0) n += Hz[(size_t)y * nx + x - 1];
    if (y > 0) n += Hz[(size_t)(y - 1) * nx + x];
    if (x > 0 && y > 0) n += Hz[(size_t)(y - 1) * nx + x - 1];
    return n;
}

// stats: [0] columns proven, [1] columns with a plane below the threshold that the bounds cannot exclude, [2] float bits of the
// largest (maximum - value) / maximum among those planes (the rel_gap that would have taken them all in), [3] most votes in a voxel,
// [4] entries of unproven[] (pixel, float bits of the gap the column needs)
template <int OP>
__global__ __launch_bounds__(256) void k_tie_prove(const float* __restrict__ a, const float* __restrict__ b,
                                                   const uint32_t* __restrict__ Ha, const uint32_t* __restrict__ Hb, int nx, int ny,
                                                   int nz, float rel_gap, unsigned* __restrict__ stats, uint2* __restrict__ unproven)
{
    __shared__ unsigned s_stats[4];
    if (threadIdx.x < 4) s_stats[threadIdx.x] = 0u;
    __syncthreads();
    const int npix = nx * ny;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npix) {
        const int y = p / nx, x = p - y * nx;
        // the engine's arg-max: first maximum of the fused exact values (k_collapse_max_z_fused / k_tie_columns)
        float best = tie_value<OP>(a, b, p);
        int zbest = 0;
        for (int z = 1; z < nz; ++z) {
            const float v = tie_value<OP>(a, b, (size_t)z * npix + p);
            if (best < v) {
                best = v;
                zbest = z;
            }
        }
        bool proven = true;
        float need = 0.f;
        unsigned most = 0u;
        if (best > 0.f) {  // (an empty column is exactly zero in either summation order)
            const float thr = best - rel_gap * best;  // k_tie_columns / k_tie_contenders
            const double c8 = 8.0 * 5.9604644775390625e-8;
            double lo0, hi0, lo1 = 0.0, hi1 = 0.0;
            {
                const size_t i = (size_t)zbest * npix + p;
                const uint32_t n0 = tie_votes_of(Ha + (size_t)zbest * npix, x, y, nx);
                tie_reference_interval(a[i], n0, &lo0, &hi0);
                most = n0;
                if (OP != 0) {
                    const uint32_t n1 = tie_votes_of(Hb + (size_t)zbest * npix, x, y, nx);
                    tie_reference_interval(b[i], n1, &lo1, &hi1);
                    most = max(most, n1);
                }
            }
            const double winner_lo = tie_fuse_real<OP>(lo0, lo1) * (1.0 - c8);
            for (int z = 0; z < nz; ++z) {
                const size_t i = (size_t)z * npix + p;
                const float v = tie_value<OP>(a, b, i);
                if (v >= thr) continue;  // within the gap: re-summed in the reference's order by the resolver (or the maximum itself)
                const uint32_t n0 = tie_votes_of(Ha + (size_t)z * npix, x, y, nx);
                double l0, h0, l1 = 0.0, h1 = 0.0;
                tie_reference_interval(a[i], n0, &l0, &h0);
                most = max(most, n0);
                if (OP != 0) {
                    const uint32_t n1 = tie_votes_of(Hb + (size_t)z * npix, x, y, nx);
                    tie_reference_interval(b[i], n1, &l1, &h1);
                    most = max(most, n1);
                }
                const double upper = tie_fuse_real<OP>(h0, h1) * (1.0 + c8);
                if (!(upper < winner_lo)) {  // cannot be excluded (strictly below: an equal value at a lower plane would win)
                    proven = false;
                    need = fmaxf(need, (best - v) / best);
                }
            }
        }
        atomicAdd(&s_stats[proven ? 0 : 1], 1u);
        if (!proven) {
            atomicMax(&s_stats[2], __float_as_uint(need));  // (non-negative floats order like their bits)
            // the column and the gap it needs, for callers that treat the few hard columns one by one (room for every pixel)
            if (unproven) unproven[atomicAdd(&stats[4], 1u)] = make_uint2((unsigned)p, __float_as_uint(need));
        }
        atomicMax(&s_stats[3], most);
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_stats[threadIdx.x]) atomicAdd(&stats[threadIdx.x], s_stats[threadIdx.x]);
    if (threadIdx.x >= 2 && threadIdx.x < 4 && s_stats[threadIdx.x]) atomicMax(&stats[threadIdx.x], s_stats[threadIdx.x]);
}

// votes[i] <- the votes of voxel vox[i] (z * ny * nx + y * nx + x) according to the counters H (tie_votes_of)
__global__ __launch_bounds__(256) void k_tie_votes_of(const uint32_t* __restrict__ H, const uint32_t* __restrict__ vox, int n, int nx, int ny,
                                                      uint32_t* __restrict__ votes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = vox[i], npix = (uint32_t)nx * (uint32_t)ny;
    const uint32_t z = v / npix, p = v - z * npix, y = p / (uint32_t)nx, x = p - y * (uint32_t)nx;
    votes[i] = tie_votes_of(H + (size_t)z * npix, (int)x, (int)y, nx);
}

hipError_t launch_tie_votes_of(hipStream_t s, const uint32_t* H, const uint32_t* vox, int n, int nx, int ny, uint32_t* votes)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tie_votes_of, dim3((n + 255) / 256), dim3(256), 0, s, H, vox, n, nx, ny, votes);
    return hipExtGetLastError();
}

// ---- the proof's pieces for ANY fusion topology (Alg. 2's camera-then-time fusion, process2.cpp:98-249): INTERVAL GRIDS ----
// lo / hi: two fp32 volumes that enclose, voxel by voxel, the value the reference holds -- made per camera DSI from the
// counted votes (k_interval_of_counts), carried through the SAME device ops the engine applies to the values (every one of the
// reference's voxel-wise ops is monotone non-decreasing in its operands, or the decreasing 1 / (0.01 + g) inside the harmonic
// accumulation whose final n / sum turns it around), widened after every step by that step's roundings (k_interval_widen),
// and finally compared column by column (k_prove_columns) like k_tie_prove does.
// (the next fp32 below / above a FINITE f, by its bits: there is no nextafterf on the device)
__device__ __forceinline__ float next_below(float f)
{
    const uint32_t b = __float_as_uint(f);
    if (f > 0.f) return __uint_as_float(b - 1u);
    if (f < 0.f) return __uint_as_float(b + 1u);
    return __uint_as_float(0x80000001u);  // -min denormal
}
__device__ __forceinline__ float next_above(float f)
{
    const uint32_t b = __float_as_uint(f);
    if (f > 0.f) return __uint_as_float(b + 1u);  // (max finite -> +inf: a bound that says nothing, correctly)
    if (f < 0.f) return __uint_as_float(b - 1u);
    return __uint_as_float(0x00000001u);
}
__device__ __forceinline__ float round_down(double v)
{
    const float f = (float)v;  // (round to nearest; v = +-inf or NaN pass through)
    if (f == __builtin_inff() && v < (double)__builtin_inff()) return 3.4028234663852886e38f;  // (a finite v beyond FLT_MAX)
    return ((double)f > v && f == f && fabsf(f) < __builtin_inff()) ? next_below(f) : f;
}
__device__ __forceinline__ float round_up(double v)
{
    const float f = (float)v;
    return ((double)f < v && f == f && fabsf(f) < __builtin_inff()) ? next_above(f) : f;
}

__global__ __launch_bounds__(256) void k_interval_of_counts(const float* __restrict__ dsi, const uint32_t* __restrict__ H, int nx, int ny,
                                                            int nz, float* __restrict__ lo, float* __restrict__ hi,
                                                            unsigned* __restrict__ most)
{
    const size_t npix = (size_t)nx * ny, n = npix * nz;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned votes = 0u;
    if (i < n) {
        const size_t z = i / npix, p = i - z * npix;
        const int y = (int)(p / nx), x = (int)(p - (size_t)y * nx);
        votes = tie_votes_of(H + z * npix, x, y, nx);
        double l, h;
        tie_reference_interval(dsi[i], votes, &l, &h);
        lo[i] = round_down(l);
        hi[i] = round_up(h);  // (+inf where a voxel has 2^23 votes or more: never proven)
    }
    // (most votes in a voxel: one atomic per wave)
    for (int off = 32; off > 0; off >>= 1) votes = max(votes, (unsigned)__shfl_xor((int)votes, off, 64));
    if ((threadIdx.x & 63) == 0 && votes) atomicMax(most, votes);
}

// lo <- lo (1 - k u) rounded down (never below 0), hi <- hi (1 + k u) rounded up: the k roundings of the step just applied
__global__ __launch_bounds__(256) void k_interval_widen(float* __restrict__ lo, float* __restrict__ hi, size_t n, double ku)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lo[i] = round_down(fmax(0.0, (double)lo[i] * (1.0 - ku)));
    hi[i] = round_up((double)hi[i] * (1.0 + ku));
}

// per column of `fused` (the engine's values: the resolver's threshold comes from them): every plane below
// best - rel_gap * best must have hi strictly below lo of the maximum's plane.  stats / unproven as k_tie_prove
__global__ __launch_bounds__(256) void k_prove_columns(const float* __restrict__ fused, const float* __restrict__ lo,
                                                       const float* __restrict__ hi, int npix, int nz, float rel_gap,
                                                       unsigned* __restrict__ stats, uint2* __restrict__ unproven)
{
    __shared__ unsigned s_stats[3];
    if (threadIdx.x < 3) s_stats[threadIdx.x] = 0u;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npix) {
        float best = fused[p];
        int zbest = 0;
        for (int z = 1; z < nz; ++z) {
            const float v = fused[(size_t)z * npix + p];
            if (best < v) {
                best = v;
                zbest = z;
            }
        }
        bool proven = true;
        float need = 0.f;
        if (best > 0.f) {
            const float thr = best - rel_gap * best;
            const float winner_lo = lo[(size_t)zbest * npix + p];
            for (int z = 0; z < nz; ++z) {
                const size_t i = (size_t)z * npix + p;
                const float v = fused[i];
                if (v >= thr) continue;
                if (!(hi[i] < winner_lo)) {
                    proven = false;
                    need = fmaxf(need, (best - v) / best);
                }
            }
        }
        atomicAdd(&s_stats[proven ? 0 : 1], 1u);
        if (!proven) {
            atomicMax(&s_stats[2], __float_as_uint(need));
            if (unproven) unproven[atomicAdd(&stats[4], 1u)] = make_uint2((unsigned)p, __float_as_uint(need));
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_stats[threadIdx.x]) atomicAdd(&stats[threadIdx.x], s_stats[threadIdx.x]);
    if (threadIdx.x == 2 && s_stats[2]) atomicMax(&stats[2], s_stats[2]);
}

hipError_t launch_interval_of_counts(hipStream_t s, const float* dsi, const uint32_t* H, int nx, int ny, int nz, float* lo, float* hi,
                                     unsigned* most)
{
    const size_t n = (size_t)nx * ny * nz;
    hipLaunchKernelGGL(k_interval_of_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dsi, H, nx, ny, nz, lo, hi, most);
    return hipExtGetLastError();
}

hipError_t launch_interval_widen(hipStream_t s, float* lo, float* hi, size_t n, int roundings)
{
    hipLaunchKernelGGL(k_interval_widen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lo, hi, n,
                       (double)roundings * 5.9604644775390625e-8);
    return hipExtGetLastError();
}

hipError_t launch_prove_columns(hipStream_t s, const float* fused, const float* lo, const float* hi, int npix, int nz, float rel_gap,
                                unsigned* stats5, uint2* unproven)
{
    hipLaunchKernelGGL(k_prove_columns, dim3((npix + 255) / 256), dim3(256), 0, s, fused, lo, hi, npix, nz, rel_gap, stats5, unproven);
    return hipExtGetLastError();
}

// ---- the same proof for n cameras fused by an n-ary mode (dsi_mapper_prove_near_ties_n: BASELINE configs[4]'s rig) ----
// fused: the grid the n-ary resolver took its near-tie columns from (dsi_grid_near_tie_voxels: the engine's fused values);
// e[c] / h[c]: camera c's DSI and vote counters.  MODE: DSI_ACC_GM_TREE (6; n = 2, 4, 8: the balanced tree of the
// reference's 2-ary sqrt(a * b), cartesian3dgrid.h:150-156), DSI_ACC_MIN (4), DSI_ACC_MAX (5), DSI_ACC_SUM (0: the arithmetic
// mean).  Every one of them is monotone non-decreasing in each camera's value, with at most 2 n + 2 roundings.
struct ProveSources {
    const float* e[8];
    const uint32_t* h[8];
    int n;
};

template <int MODE>
__device__ __forceinline__ double tie_fuse_real_n(const double* v, int n)
{
    if (MODE == 4 || MODE == 5) {
        double r = v[0];
        for (int c = 1; c < n; ++c) r = MODE == 4 ? fmin(r, v[c]) : fmax(r, v[c]);
        return r;
    }
    if (MODE == 0) {
        double r = 0.0;
        for (int c = 0; c < n; ++c) r += v[c];
        return r / (double)n;
    }
    double t[8];
    for (int c = 0; c < n; ++c) t[c] = v[c];
    for (int w = n; w > 1; w >>= 1)
        for (int c = 0; c < w / 2; ++c) t[c] = sqrt(t[2 * c] * t[2 * c + 1]);
    return t[0];
}

template <int MODE>
__global__ __launch_bounds__(256) void k_tie_prove_n(const float* __restrict__ fused, ProveSources src, int nx, int ny, int nz,
                                                     float rel_gap, unsigned* __restrict__ stats, uint2* __restrict__ unproven)
{
    __shared__ unsigned s_stats[4];
    if (threadIdx.x < 4) s_stats[threadIdx.x] = 0u;
    __syncthreads();
    const int npix = nx * ny, n = src.n;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npix) {
        const int y = p / nx, x = p - y * nx;
        float best = fused[p];
        int zbest = 0;
        for (int z = 1; z < nz; ++z) {
            const float v = fused[(size_t)z * npix + p];
            if (best < v) {
                best = v;
                zbest = z;
            }
        }
        bool proven = true;
        float need = 0.f;
        unsigned most = 0u;
        if (best > 0.f) {
            const float thr = best - rel_gap * best;  // k_tie_columns on the fused grid
            const double slack = (double)(2 * n + 2) * 5.9604644775390625e-8;
            double lo[8], hi[8];
            for (int c = 0; c < n; ++c) {
                const uint32_t votes = tie_votes_of(src.h[c] + (size_t)zbest * npix, x, y, nx);
                tie_reference_interval(src.e[c][(size_t)zbest * npix + p], votes, &lo[c], &hi[c]);
                most = max(most, votes);
            }
            const double winner_lo = tie_fuse_real_n<MODE>(lo, n) * (1.0 - slack);
            for (int z = 0; z < nz; ++z) {
                const size_t i = (size_t)z * npix + p;
                const float v = fused[i];
                if (v >= thr) continue;  // within the gap: re-summed in the reference's order (or the maximum itself)
                for (int c = 0; c < n; ++c) {
                    const uint32_t votes = tie_votes_of(src.h[c] + (size_t)z * npix, x, y, nx);
                    tie_reference_interval(src.e[c][i], votes, &lo[c], &hi[c]);
                    most = max(most, votes);
                }
                const double upper = tie_fuse_real_n<MODE>(hi, n) * (1.0 + slack);
                if (!(upper < winner_lo)) {
                    proven = false;
                    need = fmaxf(need, (best - v) / best);
                }
            }
        }
        atomicAdd(&s_stats[proven ? 0 : 1], 1u);
        if (!proven) {
            atomicMax(&s_stats[2], __float_as_uint(need));
            if (unproven) unproven[atomicAdd(&stats[4], 1u)] = make_uint2((unsigned)p, __float_as_uint(need));
        }
        atomicMax(&s_stats[3], most);
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_stats[threadIdx.x]) atomicAdd(&stats[threadIdx.x], s_stats[threadIdx.x]);
    if (threadIdx.x >= 2 && threadIdx.x < 4 && s_stats[threadIdx.x]) atomicMax(&stats[threadIdx.x], s_stats[threadIdx.x]);
}

hipError_t launch_tie_prove_n(hipStream_t s, const float* fused, const float* const* e, const uint32_t* const* h, int n, int mode, int nx,
                              int ny, int nz, float rel_gap, unsigned* stats5, uint2* unproven)
{
    if (n < 1 || n > 8) return hipErrorInvalidValue;
    if (mode == 6 && n != 2 && n != 4 && n != 8) return hipErrorInvalidValue;
    ProveSources src{};
    src.n = n;
    for (int c = 0; c < n; ++c) {
        src.e[c] = e[c];
        src.h[c] = h[c];
    }
    const dim3 grid((nx * ny + 255) / 256), block(256);
    switch (mode) {
    case 0: hipLaunchKernelGGL(k_tie_prove_n<0>, grid, block, 0, s, fused, src, nx, ny, nz, rel_gap, stats5, unproven); break;
    case 4: hipLaunchKernelGGL(k_tie_prove_n<4>, grid, block, 0, s, fused, src, nx, ny, nz, rel_gap, stats5, unproven); break;
    case 5: hipLaunchKernelGGL(k_tie_prove_n<5>, grid, block, 0, s, fused, src, nx, ny, nz, rel_gap, stats5, unproven); break;
    case 6: hipLaunchKernelGGL(k_tie_prove_n<6>, grid, block, 0, s, fused, src, nx, ny, nz, rel_gap, stats5, unproven); break;
    default: return hipErrorInvalidValue;
    }
    return hipExtGetLastError();
}

hipError_t launch_tie_prove(hipStream_t s, const float* a, const float* b, const uint32_t* Ha, const uint32_t* Hb, int op, int nx,
                            int ny, int nz, float rel_gap, unsigned* stats4, uint2* unproven)
{
    const dim3 grid((nx * ny + 255) / 256), block(256);
#define DSI_TIE_PROVE(OPV)                                                                                      \
    case OPV:                                                                                                   \
        hipLaunchKernelGGL(k_tie_prove<OPV>, grid, block, 0, s, a, b, Ha, Hb, nx, ny, nz, rel_gap, stats4, unproven); \
        break;
    switch (b ? op : 0) {
        DSI_TIE_PROVE(0) DSI_TIE_PROVE(1) DSI_TIE_PROVE(2) DSI_TIE_PROVE(3) DSI_TIE_PROVE(4) DSI_TIE_PROVE(5) DSI_TIE_PROVE(6)
    default: return hipErrorInvalidValue;
    }
#undef DSI_TIE_PROVE
    return hipExtGetLastError();
}

hipError_t launch_tie_candidates(hipStream_t s, const float* a, const float* b, int op, int npix, int nz, float rel_gap,
                                 unsigned* counters, uint32_t* cand, uint32_t cap, uint4* cols, uint32_t cols_cap)
{
    if (nz > 256) return hipErrorInvalidValue;
    const dim3 grid((npix + 255) / 256), block(256);
    const dim3 grid2((unsigned)std::min<size_t>(512, ((size_t)cols_cap + 15) / 16)), block2(1024);
#define DSI_TIE_CAND(OPV)                                                                                                           \
    case OPV:                                                                                                                       \
        hipLaunchKernelGGL(k_tie_columns<OPV>, grid, block, 0, s, a, b, npix, nz, rel_gap, counters, cols, cols_cap);                \
        hipLaunchKernelGGL(k_tie_contenders<OPV>, grid2, block2, 0, s, a, b, npix, nz, rel_gap, counters, cand, cap, cols, cols_cap); \
        break;
    switch (b ? op : 0) {
        DSI_TIE_CAND(0) DSI_TIE_CAND(1) DSI_TIE_CAND(2) DSI_TIE_CAND(3) DSI_TIE_CAND(4) DSI_TIE_CAND(5) DSI_TIE_CAND(6)
    default: return hipErrorInvalidValue;
    }
#undef DSI_TIE_CAND
    return hipExtGetLastError();
}

hipError_t launch_tie_desc(hipStream_t s, const uint32_t* vox, int n, int nx, int npix, uint2* desc, unsigned* plane_bits)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tie_desc, dim3((n + 255) / 256), dim3(256), 0, s, vox, n, nx, npix, desc, plane_bits);
    return hipExtGetLastError();
}

int tie_segment_records() { return kTieSeg; }
int tie_block_capacity_records() { return kTieMaxSegs * kTieSeg; }

hipError_t launch_tie_hits_binned(hipStream_t s, const uint16_t* ex, const uint16_t* ey, const uint32_t* packet_first, const float* H,
                                  const float2* lut, int sensor_w, int sensor_h, const float* centers, const float* planes, const Geom& g,
                                  int np, const uint2* desc, int nsv, unsigned rank_base, unsigned pos_bits, unsigned sentinel_rank,
                                  unsigned* seg_counter, unsigned cap_segs, unsigned* flags, unsigned long long* total_hits,
                                  unsigned long long* keys, float* wts)
{
    const TieEvents ev{ex, ey, packet_first, H, lut, sensor_w, sensor_h};
    if (np <= 0 || nsv <= 0) return hipSuccess;
    const TieBinGeom bg = tie_bin_geom(g.nx, g.ny);
    const size_t lds = tie_hits_lds_bytes(bg.tx_n * bg.ty_n, g.nz);
    if (lds > max_dynamic_lds() || g.nx > 0xffff || g.ny > 0xffff) return hipErrorInvalidValue;
    constexpr int kHitsBlock = 512;  // 8 waves per block, <= 64 VGPRs: up to 32 waves per CU next to four blocks' LDS
    if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(k_tie_hits_binned<kHitsBlock>), lds)) return e;
    // blocks take packets in turn (the order of the hits does not matter: they are sorted); as many as fit the chip
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, max_dynamic_lds() / lds));
    const int blocks = std::min(np, 256 * per_cu);
    // ... and when the packets alone do not (a 50 ms window), the voxels are dealt over blockIdx.y (each share bins the packet again)
    int vshares = std::max(1, std::min(std::min(16, (nsv + 1023) / 1024), (256 * per_cu) / blocks));
    vshares = std::max(vshares, (nsv + (1 << 22) - 1) >> 22);  // (a queue entry holds the voxel's index within its share in 22 bits)
    hipLaunchKernelGGL(k_tie_hits_binned<kHitsBlock>, dim3(blocks, vshares), dim3(kHitsBlock), lds, s, ev, centers, planes, g, np, desc, nsv, rank_base, pos_bits,
                       sentinel_rank, bg, seg_counter, cap_segs, flags, total_hits, keys, wts);
    return hipExtGetLastError();
}


// the recorded votes [0, n_rec) partitioned by rank into runs[starts[r] .. starts[r] + counts[r]) (any order inside a run;
// a word = event position << 32 | weight bits), then exact[] / count[] / diff[] (dsi_kernels.h).  n_ranks =
// n_cams * nsv; counts, cursor: n_ranks words, starts: n_ranks + 1 (scratch).  pos_bits <= 32, n_rec < 2^32.
size_t tie_partition_cursor_words(size_t n_ranks)
{
    return n_ranks <= (size_t)kTiePartLdsRanks ? n_ranks * (size_t)kTiePartStretches : n_ranks;
}

hipError_t launch_tie_partition_sums(hipStream_t s, const unsigned long long* keys, const float* wts, unsigned long long n_rec,
                                     unsigned pos_bits, uint32_t* counts, uint32_t* starts, uint32_t* cursor, unsigned long long* runs,
                                     const uint32_t* vox, int nsv, int n_cams, const float* grid0, const float* grid1, float* exact,
                                     uint32_t* count, float* diff)
{
    const unsigned n_ranks = (unsigned)nsv * (unsigned)n_cams;
    if (n_ranks == 0) return hipSuccess;
    if (pos_bits > 32u || n_rec >= (1ull << 32)) return hipErrorInvalidValue;
    if (hipError_t e = hipMemsetAsync(counts, 0, (size_t)n_ranks * sizeof(uint32_t), s)) return e;
    if (n_rec) {
        const int use_lds = n_ranks <= (unsigned)kTiePartLdsRanks ? 1 : 0;
        const size_t lds = use_lds ? (size_t)n_ranks * sizeof(uint32_t) : 0;
        // stretches of >= 8 k votes, at most kTiePartStretches of them (one workgroup per CU's worth): a workgroup pays
        // ~n_ranks LDS words per stretch.  LDS path: NO global atomic -- the counting launch leaves its counts in a table
        // [stretch][rank] (`cursor`: tie_partition_cursor_words), k_tie_colscan turns a rank's column into the offsets of
        // the stretches inside its run (17 us), the scattering launch reads its row instead of counting again (with one
        // returning atomic per (stretch, rank) on the runs' cursors -- 256 workgroups walking the ranks in the same order --
        // the scatter took 0.277 ms at configs[1]; now 0.240)
        const unsigned long long per = std::max<unsigned long long>(8192ull, (n_rec + kTiePartStretches - 1ull) / kTiePartStretches);
        const unsigned blocks = (unsigned)((n_rec + per - 1) / per);
        if (use_lds) {
            if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_tie_partition<false>), lds)) return e;
            if (hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(&k_tie_partition<true>), lds)) return e;
        }
        hipLaunchKernelGGL(k_tie_partition<false>, dim3(blocks), dim3(1024), lds, s, keys, wts, n_rec, per, pos_bits, n_ranks, use_lds,
                           use_lds ? cursor : counts, (const uint32_t*)nullptr, (unsigned long long*)nullptr);
        if (hipError_t e = hipExtGetLastError()) return e;
        if (use_lds) {
            hipLaunchKernelGGL(k_tie_colscan, dim3((n_ranks + 63) / 64), dim3(64), 0, s, cursor, n_ranks, blocks, counts);
            if (hipError_t e = hipExtGetLastError()) return e;
        }
        hipLaunchKernelGGL(k_tie_scan, dim3(1), dim3(1024), 0, s, counts, n_ranks, starts, use_lds ? (uint32_t*)nullptr : cursor);
        if (hipError_t e = hipExtGetLastError()) return e;
        hipLaunchKernelGGL(k_tie_partition<true>, dim3(blocks), dim3(1024), lds, s, keys, wts, n_rec, per, pos_bits, n_ranks, use_lds,
                           cursor, (const uint32_t*)starts, runs);
        if (hipError_t e = hipExtGetLastError()) return e;
    } else {
        hipLaunchKernelGGL(k_tie_scan, dim3(1), dim3(1024), 0, s, counts, n_ranks, starts, cursor);
        if (hipError_t e = hipExtGetLastError()) return e;
    }
    // (the unsorted weights have been consumed by the scatter: their array receives the weights in event order)
    float* sorted_w = const_cast<float*>(wts);
    if (n_rec) {
        hipLaunchKernelGGL(k_tie_sort_runs<1024>, dim3(n_ranks), dim3(256), 0, s, runs, starts, counts, pos_bits, (int)n_ranks, sorted_w);
        if (hipError_t e = hipExtGetLastError()) return e;
        hipLaunchKernelGGL(k_tie_sort_runs<2048>, dim3(n_ranks), dim3(512), 0, s, runs, starts, counts, pos_bits, (int)n_ranks, sorted_w);
        if (hipError_t e = hipExtGetLastError()) return e;
        hipLaunchKernelGGL(k_tie_sort_runs<kTieRunLds>, dim3(n_ranks), dim3(512), 0, s, runs, starts, counts, pos_bits, (int)n_ranks, sorted_w);
        if (hipError_t e = hipExtGetLastError()) return e;
    }
    hipLaunchKernelGGL(k_tie_add_runs, dim3((n_ranks + 15) / 16), dim3(256), 0, s, sorted_w, starts, counts, vox, nsv, n_cams, grid0, grid1,
                       exact, count, diff);
    return hipExtGetLastError();
}

hipError_t launch_tie_pick(hipStream_t s, int op, const uint4* cols, int n_cols, const uint32_t* vox, int nsv, int npix,
                           const float* exact, const uint32_t* count, const float* diff, const float* planes, float* conf,
                           uint8_t* idx, float* depth, unsigned* stats, float rel_gap)
{
    if (n_cols <= 0) return hipSuccess;
    const dim3 grid((n_cols + 255) / 256), block(256);
#define DSI_TIE_PICK(OPV)                                                                                                        \
    case OPV:                                                                                                                    \
        hipLaunchKernelGGL(k_tie_pick<OPV>, grid, block, 0, s, cols, n_cols, vox, nsv, npix, exact, count, diff, planes, conf, idx,   \
                           depth, stats, rel_gap);                                                                               \
        break;
    switch (op) {
        DSI_TIE_PICK(0) DSI_TIE_PICK(1) DSI_TIE_PICK(2) DSI_TIE_PICK(3) DSI_TIE_PICK(4) DSI_TIE_PICK(5) DSI_TIE_PICK(6)
    default: return hipErrorInvalidValue;
    }
#undef DSI_TIE_PICK
    return hipExtGetLastError();
}

hipError_t launch_tie_patch(hipStream_t s, const uint32_t* pix, const uint8_t* new_idx, const float* new_conf, int n,
                            const float* planes, float* conf, uint8_t* idx, float* depth)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tie_patch, dim3((n + 255) / 256), dim3(256), 0, s, pix, new_idx, new_conf, n, planes, conf, idx, depth);
    return hipExtGetLastError();
}

}  // namespace dsi
