// rasterizer.hip -- forward pass of the tile-binned Gaussian splat rasteriser + visible_filter
// (diff-gaussian-rasterization, Scaffold-GS fork; the zip is missing from the reference tree, see
// SURVEY.md App. E; call contract: HAC/gaussian_renderer/__init__.py:199-225, 268-303 and
// TC-GS/SIBR_viewers/src/projects/gaussianviewer/renderer/GaussianView.cpp:535-553, 660-688).
// Forward only: RD evaluation (render -> PSNR) never needs the backward.
//
// gfx950 mapping:
//   k_preprocess     one lane per Gaussian, coalesced SoA outputs (HBM-bound); counts the tiles the splat can reach (tile_touches: exact tile culling)
//   scan / sort      the library's own wave-ballot radix sort: the Gaussians by depth (4 passes), the duplicates emitted in that order, then by tile (2 passes)
//                    -- the permutation of the reference's single sort on (tile << 32 | depth bits), which GAUSPCC_RASTER_SORT2=0 still runs
//   k_tile_ranges    boundary detection on the sorted keys
//   k_render         one 128-lane workgroup per 16x16 tile, two pixels per lane; batches of 128 Gaussians staged in LDS
//                    (id -> xy, conic+opacity, rgb = 36 B each), every lane blends its pixels front to
//                    back and the workgroup leaves when all 256 pixels are saturated
#include "primitives.hpp"

using namespace gpcc;

namespace {

constexpr int TB = 256;
constexpr int BX = 16, BY = 16;

struct Cam {
    const float *view, *proj;   // the caller's 4 x 4 matrices, read where they lie (device memory): no blocking copies to the host in front of a frame
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy;
};

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.0f) * (float)S - 1.0f) * 0.5f; }

// Exact tile culling (round 5).  The reference bins a Gaussian into every tile of the 3-sigma bounding square of its larger axis; the blend then
// skips it at every pixel where alpha = op exp(power) < 1 / 255.  A pair (Gaussian, tile) whose `power` stays below -ln(255 op) over the whole
// tile is therefore dead weight in the sort and in the tile's list -- about a third of all pairs (corners of the squares, thin or faint splats).
// q(d) = -power(d) = 0.5 (A dx^2 + C dy^2) + B dx dy is convex with its minimum 0 at the centre, so its minimum over the tile's rectangle of
// pixel centres is 0 when the centre lies inside and is attained on one of the four edges otherwise (a 1-D quadratic each, clamped).  The pair is
// kept when that minimum <= ln(255 op) + margin; the margin leaves every borderline pixel to the exact test in the blend, so the image is
// bit-identical with and without culling (tests/test_gpu_rasterizer.py) and only the lists get shorter.  The margin has an absolute part (2e-3,
// twice k_render's own pre-test) and, since round 6, a part that SCALES WITH THE TERMS: for a long thin splat far from its centre each of the
// three terms of q is ~1e6 and they cancel down to q ~ 5, so the fp32 edge minimum here and the per-pixel power of k_render (another order of
// the same operations) may differ by several ulp of 1e6 -- far more than 2e-3.  8 eps x (|term| summed) bounds both sides' error.
__device__ __forceinline__ float edge_min(float c, float lo, float hi, float Ac, float Bm, float Cv, float inv_Cv)
{   // min over v in [lo, hi] of 0.5 Ac c^2 + Bm c v + 0.5 Cv v^2, less the rounding slack of its evaluation (Ac, Cv > 0).  The minimiser through the
    // reciprocal the caller computed once per Gaussian (four correctly rounded divisions per tile were 40 of this test's ~90 instructions): any v in
    // [lo, hi] gives a value >= the minimum, and an error of an ulp in v moves the value by its square
    const float v = fminf(hi, fmaxf(lo, -Bm * c * inv_Cv));
    const float t0 = 0.5f * Ac * c * c, t1 = Bm * c * v, t2 = 0.5f * Cv * v * v;
    return (t0 + t1 + t2) - 9.6e-7f * (t0 + fabsf(t1) + t2);
}
__device__ __forceinline__ bool tile_touches(float gx_, float gy_, float A, float B, float C, float inv_A, float inv_C, float lim, int tx, int ty)
{
    // d = centre - pixel, pixels tx*16 .. tx*16 + 15 (the last tiles' pixels beyond the image only make the rectangle larger)
    const float dx_hi = gx_ - (float)(tx * BX), dx_lo = dx_hi - (float)(BX - 1);
    const float dy_hi = gy_ - (float)(ty * BY), dy_lo = dy_hi - (float)(BY - 1);
    if (!(lim < __builtin_inff())) return true;     // conics outside the argument (cull_limit): the reference's lists
    if (dx_lo <= 0.0f && dx_hi >= 0.0f && dy_lo <= 0.0f && dy_hi >= 0.0f) return true;
    const float m = fminf(fminf(edge_min(dx_lo, dy_lo, dy_hi, A, B, C, inv_C), edge_min(dx_hi, dy_lo, dy_hi, A, B, C, inv_C)),
                          fminf(edge_min(dy_lo, dx_lo, dx_hi, C, B, A, inv_A), edge_min(dy_hi, dx_lo, dx_hi, C, B, A, inv_A)));
    return !(m > lim);                              // (a NaN anywhere keeps the pair)
}
// the bound on q for a Gaussian of opacity op: +inf for conics the argument above does not cover (never culled), -1 when op < 1 / 255 (never blended)
__device__ __forceinline__ float cull_limit(float A, float B, float C, float op)
{
    if (!(A > 0.0f && C > 0.0f && A * C - B * B > 0.0f)) return __builtin_inff();
    if (op != op) return __builtin_inff();          // a NaN opacity blends as alpha 0.99 (fminf): keep the pair
    if (!(op > 0.0f)) return -1.0f;
    return __logf(255.0f * op) + 2e-3f;
}

constexpr uint32_t REC_BIG = 0x80000000u;
constexpr int RECT_SLOTS = 256;

// One Gaussian of k_preprocess.  Returns the tile count of its bounding square when culling (what the reference calls num_rendered is their sum), else 0;
// *dbits: the depth's bit pattern of a visible Gaussian (visible depths are positive: the patterns order as the values do), 0xFFFFFFFF otherwise.
__device__ __forceinline__ uint32_t preprocess_one(int i, const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ rots,
                                                   const float *__restrict__ cov3d_pre, const float *__restrict__ opac, const Cam &cam, int *__restrict__ radii,
                                                   float2 *__restrict__ xy, float *__restrict__ depth, float4 *__restrict__ conic_op,
                                                   uint32_t *__restrict__ tiles_touched, int cull, uint32_t *dbits, uint32_t *cnt_out, uint4 *rec)
{
    radii[i] = 0;
    if (tiles_touched) tiles_touched[i] = 0;
    const float px = means[3 * i], py = means[3 * i + 1], pz = means[3 * i + 2];
    const float *__restrict__ V = cam.view, *__restrict__ M = cam.proj;   // wave-uniform addresses: scalar / broadcast loads
    // transformPoint4x3 / 4x4 with the row-vector (transposed) matrices the Python side passes
    const float tx0 = V[0] * px + V[4] * py + V[8] * pz + V[12];
    const float ty0 = V[1] * px + V[5] * py + V[9] * pz + V[13];
    const float tz = V[2] * px + V[6] * py + V[10] * pz + V[14];
    if (tz <= 0.2f) return 0;  // in_frustum
    const float hx = M[0] * px + M[4] * py + M[8] * pz + M[12];
    const float hy = M[1] * px + M[5] * py + M[9] * pz + M[13];
    const float hw = M[3] * px + M[7] * py + M[11] * pz + M[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float prx = hx * pw, pry = hy * pw;
    // 3-D covariance (upper triangle)
    float c[6];
    if (cov3d_pre) {
        for (int k = 0; k < 6; ++k) c[k] = cov3d_pre[6 * i + k];
    } else {
        const float sx = cam.scale_modifier * scales[3 * i], sy = cam.scale_modifier * scales[3 * i + 1], sz = cam.scale_modifier * scales[3 * i + 2];
        const float r = rots[4 * i], x = rots[4 * i + 1], y = rots[4 * i + 2], z = rots[4 * i + 3];
        // R (row-major) from the quaternion (r, x, y, z); M = S * R; Sigma = M^T M
        const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
        const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
        const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
        // Sigma = R diag(s^2) R^T  (== M^T M with glm's column-major M = S * R)
        const float ax = sx * sx, ay = sy * sy, az = sz * sz;
        c[0] = ax * R00 * R00 + ay * R01 * R01 + az * R02 * R02;
        c[1] = ax * R00 * R10 + ay * R01 * R11 + az * R02 * R12;
        c[2] = ax * R00 * R20 + ay * R01 * R21 + az * R02 * R22;
        c[3] = ax * R10 * R10 + ay * R11 * R11 + az * R12 * R12;
        c[4] = ax * R10 * R20 + ay * R11 * R21 + az * R12 * R22;
        c[5] = ax * R20 * R20 + ay * R21 * R21 + az * R22 * R22;
    }
    // 2-D covariance: A Sigma A^T, A = J * Rwc, with the +-1.3 tan(fov) clamp of the view-space position
    const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
    const float tx = fminf(limx, fmaxf(-limx, tx0 / tz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, ty0 / tz)) * tz;
    const float j00 = cam.focal_x / tz, j02 = -(cam.focal_x * tx) / (tz * tz);
    const float j11 = cam.focal_y / tz, j12 = -(cam.focal_y * ty) / (tz * tz);
    const float a0x = j00 * V[0] + j02 * V[2], a0y = j00 * V[4] + j02 * V[6], a0z = j00 * V[8] + j02 * V[10];
    const float a1x = j11 * V[1] + j12 * V[2], a1y = j11 * V[5] + j12 * V[6], a1z = j11 * V[9] + j12 * V[10];
    const float s0x = c[0] * a0x + c[1] * a0y + c[2] * a0z, s0y = c[1] * a0x + c[3] * a0y + c[4] * a0z, s0z = c[2] * a0x + c[4] * a0y + c[5] * a0z;
    const float s1x = c[0] * a1x + c[1] * a1y + c[2] * a1z, s1y = c[1] * a1x + c[3] * a1y + c[4] * a1z, s1z = c[2] * a1x + c[4] * a1y + c[5] * a1z;
    const float cxx = a0x * s0x + a0y * s0y + a0z * s0z + 0.3f;
    const float cxy = a0x * s1x + a0y * s1y + a0z * s1z;
    const float cyy = a1x * s1x + a1y * s1y + a1z * s1z + 0.3f;
    const float det = cxx * cyy - cxy * cxy;
    if (det == 0.0f) return 0;
    const float det_inv = 1.0f / det;
    const float mid = 0.5f * (cxx + cyy);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.0f * sqrtf(lam));  // lambda1 >= lambda2
    const float ix = ndc2pix(prx, cam.W), iy = ndc2pix(pry, cam.H);
    const int rx0 = min(cam.gx, max(0, (int)((ix - my_radius) / BX))), ry0 = min(cam.gy, max(0, (int)((iy - my_radius) / BY)));
    const int rx1 = min(cam.gx, max(0, (int)((ix + my_radius + BX - 1) / BX))), ry1 = min(cam.gy, max(0, (int)((iy + my_radius + BY - 1) / BY)));
    const int area = (rx1 - rx0) * (ry1 - ry0);
    if (area == 0) return 0;
    radii[i] = (int)my_radius;
    if ((int)my_radius > 0) *dbits = __float_as_uint(tz);
    if (!tiles_touched) return 0;
    const float4 co = make_float4(cyy * det_inv, -cxy * det_inv, cxx * det_inv, opac[i]);
    uint32_t cnt = (uint32_t)area;
    const int w = rx1 - rx0;
    uint64_t m = area >= 64 ? ~0ull : (1ull << area) - 1ull;   // bit (y - ry0) * w + (x - rx0): the tiles of the square that take the Gaussian
    if (cull) {
        const float lim = cull_limit(co.x, co.y, co.z, co.w);
        const float inv_a = 1.0f / co.x, inv_c = 1.0f / co.z;       // (used only when lim is finite: co.x, co.z > 0 then)
        cnt = 0; m = 0;
        int b = 0;
        for (int y = ry0; y < ry1; ++y)
            for (int x = rx0; x < rx1; ++x, ++b)
                if (tile_touches(ix, iy, co.x, co.y, co.z, inv_a, inv_c, lim, x, y)) { ++cnt; m |= 1ull << (b & 63); }
    }
    tiles_touched[i] = cnt;
    *cnt_out = cnt;
    *rec = make_uint4((uint32_t)m, (uint32_t)(m >> 32), (uint32_t)rx0 | ((uint32_t)ry0 << 16), area > 64 ? REC_BIG : (uint32_t)w);
    xy[i] = make_float2(ix, iy);
    depth[i] = tz;
    conic_op[i] = co;
    return cull ? (uint32_t)area : 0u;
}

// dkeys / dvals / recs (optional, together): the (depth, index) pairs of the two-level sort's first level, written here instead of by a pass of their
// own.  The key's upper word carries the Gaussian's tile count (the sort looks at the lower 32 bits only and moves whole keys), so the counts come out
// of the sort in depth order; recs[i] is what k_duplicate_sorted needs of Gaussian i in ONE 16-byte gather: the square's origin and width and the
// mask of its tiles that take the Gaussian (squares of more than 64 tiles: REC_BIG, that kernel walks the square again).
// rect_total (optional, zeroed by the caller): one atomic add per wave of the bounding squares' tile counts.
__global__ __launch_bounds__(TB) void k_preprocess(int P, const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ rots,
                                                   const float *__restrict__ cov3d_pre, const float *__restrict__ opac, Cam cam, int *__restrict__ radii,
                                                   float2 *__restrict__ xy, float *__restrict__ depth, float4 *__restrict__ conic_op,
                                                   uint32_t *__restrict__ tiles_touched, int cull, unsigned long long *__restrict__ rect_total,
                                                   uint64_t *__restrict__ dkeys, uint32_t *__restrict__ dvals, uint4 *__restrict__ recs)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    uint32_t dbits = 0xFFFFFFFFu, area = 0, cnt = 0;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (i < P) {
        area = preprocess_one(i, means, scales, rots, cov3d_pre, opac, cam, radii, xy, depth, conic_op, tiles_touched, cull, &dbits, &cnt, &rec);
        if (dkeys) { dkeys[i] = ((uint64_t)cnt << 32) | dbits; dvals[i] = (uint32_t)i; recs[i] = rec; }
    }
    if (rect_total) {   // RECT_SLOTS partial sums: 73 k wave atomics on ONE address serialise at ~10 ns each (0.7 ms at 4.66 M Gaussians, measured)
        for (int d = 32; d; d >>= 1) area += __shfl_xor(area, d);
        if ((threadIdx.x & 63) == 0 && area) atomicAdd(rect_total + ((blockIdx.x * (TB / 64) + (threadIdx.x >> 6)) & (RECT_SLOTS - 1)), (unsigned long long)area);
    }
}

__global__ __launch_bounds__(RECT_SLOTS) void k_sum_slots(const unsigned long long *__restrict__ slots, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[RECT_SLOTS / 64];
    unsigned long long acc = slots[threadIdx.x];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += (unsigned long long)__shfl_xor((long long)acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < RECT_SLOTS / 64; ++k) t += part[k];
        *total = t;
    }
}

__global__ __launch_bounds__(TB) void k_duplicate(int P, const float2 *__restrict__ xy, const float *__restrict__ depth, const uint32_t *__restrict__ offs,
                                                  const int *__restrict__ radii, int gx, int gy, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                  const float4 *__restrict__ conic_op, int cull)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= P || radii[i] <= 0) return;
    uint32_t off = offs[i];
    const float r = (float)radii[i];
    const float2 p = xy[i];
    const int rx0 = min(gx, max(0, (int)((p.x - r) / BX))), ry0 = min(gy, max(0, (int)((p.y - r) / BY)));
    const int rx1 = min(gx, max(0, (int)((p.x + r + BX - 1) / BX))), ry1 = min(gy, max(0, (int)((p.y + r + BY - 1) / BY)));
    const uint32_t dbits = __float_as_uint(depth[i]);
    const float4 co = conic_op[i];
    const float lim = cull ? cull_limit(co.x, co.y, co.z, co.w) : 0.0f;
    const float inv_a = 1.0f / co.x, inv_c = 1.0f / co.z;       // as in k_preprocess: the count there and the pairs here come from the same test
    for (int y = ry0; y < ry1; ++y)
        for (int x = rx0; x < rx1; ++x) {
            if (cull && !tile_touches(p.x, p.y, co.x, co.y, co.z, inv_a, inv_c, lim, x, y)) continue;
            keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
            vals[off] = (uint32_t)i;
            ++off;
        }
}

// Two-level form of the (tile | depth) sort (round 5).  The reference sorts the L duplicated (tile << 32 | depth) keys in one radix sort
// (six 8-bit passes over 18 M pairs at 1 M anchors: the largest part of a frame).  A stable sort by depth of the P Gaussians (four passes over
// 5.4 M), the duplicates emitted in THAT order, and a stable sort of the duplicates by tile alone (two passes) give the same permutation:
// within a tile, depth ascending, equal depths in Gaussian order -- exactly what the stable sort of the combined key produces.  (The first
// level's (depth, index) pairs come out of k_preprocess.)
__global__ __launch_bounds__(TB) void k_sorted_counts(int P, const uint64_t *__restrict__ keys, uint32_t *__restrict__ out)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i < P) out[i] = (uint32_t)(keys[i] >> 32);
}

__global__ __launch_bounds__(TB) void k_duplicate_sorted(int P, const uint32_t *__restrict__ perm, const uint4 *__restrict__ recs, const float2 *__restrict__ xy,
                                                         const uint32_t *__restrict__ offs, const int *__restrict__ radii, int gx, int gy, uint64_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals, const float4 *__restrict__ conic_op, int cull)
{
    const int s = blockIdx.x * TB + threadIdx.x;
    if (s >= P) return;
    const uint32_t i = perm[s];
    const uint4 rec = recs[i];
    uint32_t off = offs[s];
    if (!(rec.w & REC_BIG)) {     // invisible Gaussians have an empty mask
        uint64_t m = (uint64_t)rec.x | ((uint64_t)rec.y << 32);
        const uint32_t w = rec.w;
        const uint64_t row_mask = w >= 64u ? ~0ull : (1ull << w) - 1ull;
        for (uint32_t row = (rec.z >> 16) * (uint32_t)gx + (rec.z & 0xFFFFu); m; m = w >= 64u ? 0ull : m >> w, row += (uint32_t)gx)
            for (uint64_t bits = m & row_mask; bits; bits &= bits - 1ull) {
                keys[off] = (uint64_t)(row + (uint32_t)__builtin_ctzll(bits));
                vals[off] = i;
                ++off;
            }
        return;
    }
    const float r = (float)radii[i];
    const float2 p = xy[i];
    const int rx0 = min(gx, max(0, (int)((p.x - r) / BX))), ry0 = min(gy, max(0, (int)((p.y - r) / BY)));
    const int rx1 = min(gx, max(0, (int)((p.x + r + BX - 1) / BX))), ry1 = min(gy, max(0, (int)((p.y + r + BY - 1) / BY)));
    const float4 co = conic_op[i];
    const float lim = cull ? cull_limit(co.x, co.y, co.z, co.w) : 0.0f;
    const float inv_a = 1.0f / co.x, inv_c = 1.0f / co.z;       // as in k_preprocess: the count there and the pairs here come from the same test
    for (int y = ry0; y < ry1; ++y)
        for (int x = rx0; x < rx1; ++x) {
            if (cull && !tile_touches(p.x, p.y, co.x, co.y, co.z, inv_a, inv_c, lim, x, y)) continue;
            keys[off] = (uint64_t)(uint32_t)(y * gx + x);
            vals[off] = i;
            ++off;
        }
}

__global__ __launch_bounds__(TB) void k_tile_ranges(int L, const uint64_t *__restrict__ keys, int shift, uint2 *__restrict__ ranges)
{
    const int i = blockIdx.x * TB + threadIdx.x;
    if (i >= L) return;
    const uint32_t t = (uint32_t)(keys[i] >> shift);
    if (i == 0) ranges[t].x = 0;
    else {
        const uint32_t pt = (uint32_t)(keys[i - 1] >> shift);
        if (t != pt) { ranges[pt].y = (uint32_t)i; ranges[t].x = (uint32_t)i; }
    }
    if (i == L - 1) ranges[t].y = (uint32_t)L;
}

// One 16 x 16 tile per workgroup of RT = 128 lanes (two waves); a lane blends TWO pixels, (x, y) and (x, y + 8): one fetch of a
// Gaussian from LDS (three broadcast reads) and one pass of the loop's control feed two blends.  The blend is predicated instead of
// three data-dependent `continue`s (no exec-mask juggling), the exponential is one v_exp_f32 on power * log2(e) -- `power` itself is the
// reference's expression, so its sign test is the reference's --, and a wave leaves a batch as soon as all its pixels are saturated.
// Round 4 measured the one-pixel-per-lane form at 36 % of the VALU peak (three LDS waits, a libm exponential and three branches for ~15
// useful VALU instructions per Gaussian and pixel: profiles/r04_render_counters.txt).
constexpr int RT = 128;
__device__ __forceinline__ void blend_one(float power, float op, float r, float g, float b, bool &done, float &T, float &C0, float &C1, float &C2)
{
    const float e = __builtin_amdgcn_exp2f(power * 1.44269504088896341f);
    const float alpha = fminf(0.99f, op * e);
    const bool ok = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
    const float tt = T * (1.0f - alpha);
    const bool sat = ok && tt < 0.0001f;
    done = done || sat;
    const bool upd = ok && !sat;
    const float w = upd ? alpha * T : 0.0f;
    C0 += r * w; C1 += g * w; C2 += b * w;
    T = upd ? tt : T;
}

// dispatch order of the tiles: longest list first (the lists of a frame differ by two orders of magnitude; a long tile that starts last is the
// frame's tail).  The order is a schedule, not a result: lengths in steps of 64 entries, saturating at 16 k, make it ONE radix pass (20-bit keys: three).
__global__ __launch_bounds__(TB) void k_tile_order_keys(const uint2 *__restrict__ ranges, int ntiles, uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    const int t = blockIdx.x * TB + threadIdx.x;
    if (t >= ntiles) return;
    const uint32_t c = ranges[t].y - ranges[t].x;
    key[t] = 255ull - (uint64_t)min(c >> 6, 255u);
    idx[t] = (uint32_t)t;
}

__global__ __launch_bounds__(RT) void k_render(const uint2 *__restrict__ ranges, const uint32_t *__restrict__ tile_order, const uint32_t *__restrict__ point_list, int W, int H, int gx,
                                               const float2 *__restrict__ xy, const float *__restrict__ colors, const float4 *__restrict__ conic_op,
                                               const float *__restrict__ bg, float *__restrict__ out)
{
    __shared__ float4 s_a[RT];   // x, y, conic.x, conic.y
    __shared__ float4 s_b[RT];   // conic.z, opacity, r, g
    __shared__ float2 s_c[RT];   // b, skip threshold: a Gaussian contributes to a pixel only if thr <= power <= 0
    const int tile = (int)tile_order[blockIdx.x];
    const int tbx = tile % gx, tby = tile / gx;
    const int pxi = tbx * BX + (threadIdx.x & 15), py0 = tby * BY + (threadIdx.x >> 4), py1 = py0 + 8;
    const bool in0 = pxi < W && py0 < H, in1 = pxi < W && py1 < H;
    const float pxf = (float)pxi, pyf0 = (float)py0, pyf1 = (float)py1;
    const uint2 range = ranges[tile];
    bool done0 = !in0, done1 = !in1;
    float T0 = 1.0f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
    float T1 = 1.0f, B0 = 0.f, B1 = 0.f, B2 = 0.f;
    for (uint32_t b0 = range.x; b0 < range.y; b0 += RT) {
        if (__syncthreads_count(done0 && done1) == RT) break;
        const uint32_t k = b0 + threadIdx.x;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            const float2 p = xy[id];
            const float4 co = conic_op[id];
            s_a[threadIdx.x] = make_float4(p.x, p.y, co.x, co.y);
            s_b[threadIdx.x] = make_float4(co.z, co.w, colors[3 * id], colors[3 * id + 1]);
            // alpha = op * exp(power) >= 1 / 255  <=>  power >= -ln(255 op); a margin keeps the borderline pixels on the exact test below
            // (a NaN opacity blends as alpha 0.99 in the reference -- fminf(0.99, NaN) -- so it must pass the pre-test: threshold -inf)
            s_c[threadIdx.x] = make_float2(colors[3 * id + 2], co.w > 0.0f ? -__logf(255.0f * co.w) - 1e-3f : (co.w != co.w ? -__builtin_inff() : __builtin_inff()));
        } else {
            // behind the list: a Gaussian of opacity 0 (alpha 0 < 1 / 255: never blended), so that the loop below runs in pairs
            s_a[threadIdx.x] = make_float4(0.f, 0.f, 1.f, 0.f); s_b[threadIdx.x] = make_float4(1.f, 0.f, 0.f, 0.f); s_c[threadIdx.x] = make_float2(0.f, __builtin_inff());
        }
        __syncthreads();
        const int cnt = (int)min((uint32_t)RT, range.y - b0);
        for (int j = 0; j < cnt; j += 2) {
            if (__builtin_amdgcn_ballot_w64(!(done0 && done1)) == 0ull) break;   // the wave's 128 pixels are saturated
            // two Gaussians per trip: their six LDS reads are in flight together
            const float4 a = s_a[j], bq = s_b[j], a2 = s_a[j + 1], bq2 = s_b[j + 1];
            const float2 cb = s_c[j], cb2 = s_c[j + 1];
            const float dxa = a.x - pxf, dya0 = a.y - pyf0, dya1 = a.y - pyf1;
            const float pa0 = -0.5f * (a.z * dxa * dxa + bq.x * dya0 * dya0) - a.w * dxa * dya0;
            const float pa1 = -0.5f * (a.z * dxa * dxa + bq.x * dya1 * dya1) - a.w * dxa * dya1;
            const float dxb = a2.x - pxf, dyb0 = a2.y - pyf0, dyb1 = a2.y - pyf1;
            const float pb0 = -0.5f * (a2.z * dxb * dxb + bq2.x * dyb0 * dyb0) - a2.w * dxb * dyb0;
            const float pb1 = -0.5f * (a2.z * dxb * dxb + bq2.x * dyb1 * dyb1) - a2.w * dxb * dyb1;
            // a Gaussian whose footprint misses all 128 pixels of the wave (most of a tile's list at the corners of the 3-sigma boxes, and
            // every Gaussian that only touches the tile's other half) costs the twelve instructions above and no exponential
            if (__builtin_amdgcn_ballot_w64((!done0 && !(pa0 < cb.y) && !(pa0 > 0.0f)) || (!done1 && !(pa1 < cb.y) && !(pa1 > 0.0f))) != 0ull) {   // (a NaN power passes, as in the reference's `if (power > 0) continue`)
                blend_one(pa0, bq.y, bq.z, bq.w, cb.x, done0, T0, A0, A1, A2);
                blend_one(pa1, bq.y, bq.z, bq.w, cb.x, done1, T1, B0, B1, B2);
            }
            if (__builtin_amdgcn_ballot_w64((!done0 && !(pb0 < cb2.y) && !(pb0 > 0.0f)) || (!done1 && !(pb1 < cb2.y) && !(pb1 > 0.0f))) != 0ull) {
                blend_one(pb0, bq2.y, bq2.z, bq2.w, cb2.x, done0, T0, A0, A1, A2);
                blend_one(pb1, bq2.y, bq2.z, bq2.w, cb2.x, done1, T1, B0, B1, B2);
            }
        }
    }
    const size_t plane = (size_t)W * H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (in0) {
        const size_t pix = (size_t)py0 * W + pxi;
        out[pix] = A0 + T0 * bg0; out[plane + pix] = A1 + T0 * bg1; out[2 * plane + pix] = A2 + T0 * bg2;
    }
    if (in1) {
        const size_t pix = (size_t)py1 * W + pxi;
        out[pix] = B0 + T1 * bg0; out[plane + pix] = B1 + T1 * bg1; out[2 * plane + pix] = B2 + T1 * bg2;
    }
}

int make_cam(Cam *cam, int W, int H, const float *view_dev, const float *proj_dev, float tan_fovx, float tan_fovy, float scale_modifier)
{
    cam->view = view_dev; cam->proj = proj_dev;
    cam->tan_fovx = tan_fovx; cam->tan_fovy = tan_fovy;
    cam->focal_x = (float)W / (2.0f * tan_fovx); cam->focal_y = (float)H / (2.0f * tan_fovy);
    cam->scale_modifier = scale_modifier;
    cam->W = W; cam->H = H; cam->gx = (W + BX - 1) / BX; cam->gy = (H + BY - 1) / BY;
    return GPCC_OK;
}

}  // namespace

extern "C" int gsr_visible_filter(gpcc_ctx *ctx, int P, int W, int H, const float *means3D, const float *scales, float scale_modifier,
                                  const float *rotations, const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                  float tan_fovx, float tan_fovy, int prefiltered, int *radii, void *stream)
{
    (void)prefiltered;
    if (!ctx || !means3D || !viewmatrix || !projmatrix || !radii) return fail(GPCC_ERR_ARG, "null argument");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(GPCC_ERR_ARG, "provide scales + rotations or cov3D_precomp");
    if (P <= 0) return GPCC_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    Cam cam;
    GP_TRY(make_cam(&cam, W, H, viewmatrix, projmatrix, tan_fovx, tan_fovy, scale_modifier));
    k_preprocess<<<(unsigned)cdiv(P, TB), TB, 0, st>>>(P, means3D, scales, rotations, cov3D_precomp, nullptr, cam, radii, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    LAUNCH_CHECK();
    return GPCC_OK;
}

extern "C" int gsr_forward(gpcc_ctx *ctx, int P, const float *background, int W, int H, const float *means3D, const float *colors_precomp,
                           const float *opacities, const float *scales, float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, float tan_fovx, float tan_fovy, int prefiltered, float *out_color,
                           int *radii, int64_t *num_rendered_out, void *stream)
{
    (void)prefiltered;
    if (!ctx || !background || !viewmatrix || !projmatrix || !out_color) return fail(GPCC_ERR_ARG, "null argument");
    if (P > 0 && (!means3D || !colors_precomp || !opacities || !radii)) return fail(GPCC_ERR_ARG, "null argument");
    if (P > 0 && !cov3D_precomp && (!scales || !rotations)) return fail(GPCC_ERR_ARG, "provide scales + rotations or cov3D_precomp");
    if (W <= 0 || H <= 0) return fail(GPCC_ERR_ARG, "bad image size");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    Cam cam;
    GP_TRY(make_cam(&cam, W, H, viewmatrix, projmatrix, tan_fovx, tan_fovy, scale_modifier));
    const int ntiles = cam.gx * cam.gy;
    size_t want = (size_t)std::max(P, 1) * 112 + (size_t)ntiles * 8 + ((size_t)8 << 20);
    uint32_t L = 0;
    // the two counts a frame reads back mid-way land in pinned memory (copies into stack variables are staged by the runtime, one blocking hop each)
    GP_TRY(ctx->hstage.reserve(64));
    volatile uint32_t *hL = reinterpret_cast<volatile uint32_t *>(ctx->hstage.p);
    volatile unsigned long long *hrect = reinterpret_cast<volatile unsigned long long *>(ctx->hstage.p + 8);
    for (int attempt = 0; attempt < 2; ++attempt) {
        GP_TRY(ctx->arena.reserve(want + (size_t)L * 40));
        ctx->arena.reset();
        TAKE(xy, float2, std::max(P, 1)); TAKE(depth, float, std::max(P, 1)); TAKE(conic_op, float4, std::max(P, 1));
        TAKE(touched, uint32_t, std::max(P, 1) + 1); TAKE(ranges, uint2, ntiles);
        HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)ntiles, st));
        static const bool two_level = dev_env_int("GAUSPCC_RASTER_SORT2", 1) != 0;
        static const int cull = dev_env_int("GAUSPCC_RASTER_CULL", 1) != 0 ? 1 : 0;   // exact tile culling (tile_touches); 0: the reference's lists
        TAKE(rect_total, unsigned long long, RECT_SLOTS + 1);      // the slots of k_preprocess, then their sum
        HIP_TRY(hipMemsetAsync(rect_total, 0, 8 * (RECT_SLOTS + 1), st));
        unsigned long long rect_host = 0;
        const uint32_t *perm = nullptr;          // Gaussians in depth order (two-level sort)
        const uint32_t *offs = touched;
        uint4 *recs = nullptr;
        if (P > 0) {
            uint64_t *dka = nullptr, *dkb = nullptr; uint32_t *dva = nullptr, *dvb = nullptr, *ts = nullptr;
            if (two_level) {
                dka = ctx->arena.take<uint64_t>(P); dkb = ctx->arena.take<uint64_t>(P); dva = ctx->arena.take<uint32_t>(P); dvb = ctx->arena.take<uint32_t>(P);
                ts = ctx->arena.take<uint32_t>((size_t)P + 1); recs = ctx->arena.take<uint4>(P);
                if (!dka || !dkb || !dva || !dvb || !ts || !recs) return fail(GPCC_ERR_NOMEM, "rasteriser workspace");
            }
            k_preprocess<<<(unsigned)cdiv(P, TB), TB, 0, st>>>(P, means3D, scales, rotations, cov3D_precomp, opacities, cam, radii, xy, depth, conic_op, touched, cull,
                                                               cull ? rect_total : nullptr, dka, dva, recs);
            LAUNCH_CHECK();
            if (cull) { k_sum_slots<<<1, RECT_SLOTS, 0, st>>>(rect_total, rect_total + RECT_SLOTS); LAUNCH_CHECK(); }
            if (two_level) {
                uint64_t *k0 = dka, *k1 = dkb; uint32_t *v0 = dva, *v1 = dvb;
                GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, P, 32));
                k_sorted_counts<<<(unsigned)cdiv(P, TB), TB, 0, st>>>(P, k0, ts);
                LAUNCH_CHECK();
                GP_TRY(exclusive_scan_u32(ctx, st, ts, ts, P, ts + P));
                HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(hL), ts + P, 4, hipMemcpyDeviceToHost, st));
                perm = v0; offs = ts;
            } else {
                GP_TRY(exclusive_scan_u32(ctx, st, touched, touched, P, touched + P));
                HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(hL), touched + P, 4, hipMemcpyDeviceToHost, st));
            }
            if (cull) HIP_TRY(hipMemcpyAsync(const_cast<unsigned long long *>(hrect), rect_total + RECT_SLOTS, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            L = *hL;
            if (cull) rect_host = *hrect;
        }
        static const bool stats = dev_env_int("GAUSPCC_RASTER_STATS", 0) != 0;
        if (stats) fprintf(stderr, "[gauspcc] rasteriser: %u (Gaussian, tile) pairs sorted and blended, %llu in the bounding squares\n", L, cull ? rect_host : (unsigned long long)L);
        if (num_rendered_out) *num_rendered_out = cull ? (int64_t)rect_host : (int64_t)L;   // the reference's count: every tile of every bounding square
        uint32_t *vals_sorted = nullptr;
        if (L > 0) {
            uint64_t *ka = ctx->arena.take<uint64_t>(L), *kb = ctx->arena.take<uint64_t>(L);
            uint32_t *va = ctx->arena.take<uint32_t>(L), *vb = ctx->arena.take<uint32_t>(L);
            if (!ka || !kb || !va || !vb || ctx->arena.cap - ctx->arena.off < (size_t)L * 2 + ((size_t)2 << 20)) { want += (size_t)L * 4; continue; }  // grow and redo
            int tbits = 1;
            while ((1 << tbits) < ntiles) ++tbits;
            uint64_t *k0 = ka, *k1 = kb; uint32_t *v0 = va, *v1 = vb;
            if (perm) {
                k_duplicate_sorted<<<(unsigned)cdiv(P, TB), TB, 0, st>>>(P, perm, recs, xy, offs, radii, cam.gx, cam.gy, ka, va, conic_op, cull);
                LAUNCH_CHECK();
                GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, L, tbits));
                k_tile_ranges<<<(unsigned)cdiv(L, TB), TB, 0, st>>>((int)L, k0, 0, ranges);
            } else {
                k_duplicate<<<(unsigned)cdiv(P, TB), TB, 0, st>>>(P, xy, depth, offs, radii, cam.gx, cam.gy, ka, va, conic_op, cull);
                LAUNCH_CHECK();
                GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, L, 32 + tbits));
                k_tile_ranges<<<(unsigned)cdiv(L, TB), TB, 0, st>>>((int)L, k0, 32, ranges);
            }
            LAUNCH_CHECK();
            vals_sorted = v0;
        }
        uint32_t *tile_order = nullptr;
        {
            uint64_t *oka = ctx->arena.take<uint64_t>(ntiles), *okb = ctx->arena.take<uint64_t>(ntiles);
            uint32_t *ova = ctx->arena.take<uint32_t>(ntiles), *ovb = ctx->arena.take<uint32_t>(ntiles);
            if (!oka || !okb || !ova || !ovb) { want += (size_t)ntiles * 32; continue; }
            k_tile_order_keys<<<(unsigned)cdiv(ntiles, TB), TB, 0, st>>>(ranges, ntiles, oka, ova);
            LAUNCH_CHECK();
            uint64_t *k0 = oka, *k1 = okb; uint32_t *v0 = ova, *v1 = ovb;
            GP_TRY(radix_sort_u64(ctx, st, &k0, &k1, &v0, &v1, ntiles, 8));
            tile_order = v0;
        }
        k_render<<<(unsigned)ntiles, RT, 0, st>>>(ranges, tile_order, vals_sorted, W, H, cam.gx, xy, colors_precomp, conic_op, background, out_color);
        LAUNCH_CHECK();
        HIP_TRY(hipStreamSynchronize(st));
        return device_error_check(ctx);
    }
    return fail(GPCC_ERR_NOMEM, "rasteriser workspace");
}
