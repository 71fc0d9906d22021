// rt_host_grid.h — host-side build of the uniform grid the Projector kernel's continuing rays walk.
//
// Every scan-path entry j (a sphere, or the bounding sphere of a rotation-0 box) is registered in each cell that its box
// [p - (r + E_j), p + (r + E_j)] touches.  E_j covers everything the float arithmetic of the device can get wrong about WHERE
// a hit is.  With X = o + t d the point the reference's t (sphere.rs:13-26, 48) stands for and D^ its float discriminant,
//   a (|X - p|^2 - r^2) = (a t + half_b)^2 - disc = (D^ - disc) + 2 sqrt(D^) (err(half_b) + err(t) a) + ...
// |D^ - disc| <= 25 u a L^2 (u = 2^-24, L = |o - p|: the rounding of the five products and sums behind it), sqrt(D^) <=
// |d| r' with r' = sqrt(r^2 + 25 u L^2), the bracket <= 6 u (L + r) |d|.  Hence |X - p|^2 <= r^2 + 25 u L^2 + 12 u (L + r) r'.
// With every admitted ray origin within 2 * radius of the grid centre, L <= 3 * radius =: D, and
//   E_j = 1.1 * (sqrt(r^2 + 25 u D^2 + 12 u (D + r) r') - r) + 2e-4 * D,
// the last term for the walk itself, which places points and cell boundaries with an error of a few hundred u times the
// coordinates.  (Until the end of round 2 one E = 3e-3 * D served every sphere: the r -> 0 limit of the above, 0.7 units on a
// scene whose spheres have radii 0.5 .. 2; an entry sat in 3.6 cells instead of 2.5.)
// Rays that start farther away do not use the grid (DevGrid::origin_r2).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "rt_internal.h"

#ifndef RT_GRID_CELL_SCALE
#define RT_GRID_CELL_SCALE 0.8  // cell edge in units of cbrt(volume / entries); C2 frame at 0.65 / 0.7 / 0.8 / 0.9 / 1.0 (two entries per pass): - / 3.84 / 3.76 / 3.84 / 3.93 ms
#endif

namespace rt {
namespace host {

struct GridHost {
    DevGrid g{};                  // pointers left null; the caller uploads cells / items
    std::vector<uint32_t> cells;
    std::vector<uint16_t> items;
};

constexpr uint32_t kGridMinSpheres = 96;      // below this the table scan is cheaper than the walk
constexpr uint32_t kGridMaxSpheres = 32767;   // 15-bit item indices
constexpr uint32_t kGridMaxDim = 96;

// sph_exact[0 .. n): {p, r}; sph_box[j].w != 0 marks a box entry.
inline GridHost build_grid(const std::vector<float4>& sph_exact, const std::vector<float4>& sph_box, uint32_t n)
{
    GridHost out;
    if (n < kGridMinSpheres || n > kGridMaxSpheres) return out;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t j = 0; j < n; ++j) {
        const double p[3] = {sph_exact[j].x, sph_exact[j].y, sph_exact[j].z}, r = std::fabs((double)sph_exact[j].w);
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]) || !std::isfinite(r)) return out;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a] - r);
            hi[a] = std::max(hi[a], p[a] + r);
        }
    }
    double diag2 = 0.0, c[3];
    for (int a = 0; a < 3; ++a) {
        diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
        c[a] = 0.5 * (lo[a] + hi[a]);
    }
    const double radius0 = 0.5 * std::sqrt(diag2);
    if (!(radius0 > 0.0) || !(radius0 < 1e15)) return out;
    // magnitudes: the float coordinates the device works with must resolve E
    double cmax = 0.0;
    for (int a = 0; a < 3; ++a) cmax = std::max(cmax, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    const double D = 3.0 * radius0 * 1.01;
    const double u25 = 25.0 * 0x1p-24, u12 = 12.0 * 0x1p-24;
    auto margin = [&](double r) {
        const double r1 = std::sqrt(r * r + u25 * D * D);
        return 1.1 * (std::sqrt(r * r + u25 * D * D + u12 * (D + r) * r1) - r) + 2e-4 * D;
    };
    const double E = margin(0.0);  // the largest margin any entry can have: bounds and the cell size floor use it
    if (!(E > 64.0 * 0x1p-24 * (cmax + 3.0 * radius0))) return out;  // scene far from the coordinate origin relative to its size
    for (int a = 0; a < 3; ++a) {
        lo[a] -= 2.0 * E;
        hi[a] += 2.0 * E;
    }
    double vol = 1.0, ext[3];
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        vol *= ext[a];
    }
    double cs = RT_GRID_CELL_SCALE * std::cbrt(vol / (double)n);
    // every cell at least as wide as the typical inflated sphere box would make a sphere span 8+ cells: keep cells >= 2 E
    cs = std::max(cs, 2.0 * E);
    uint32_t dim[3];
    for (;;) {
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const double d = std::ceil(ext[a] / cs);
            dim[a] = (uint32_t)std::max(1.0, d);
            if (d > (double)kGridMaxDim) ok = false;
        }
        if (ok) break;
        cs *= 1.25;
    }
    const float csf = (float)cs, inv = 1.0f / csf;
    const uint64_t n_cells = (uint64_t)dim[0] * dim[1] * dim[2];
    std::vector<uint32_t> count(n_cells, 0);
    // cell range of a sphere: computed in double from the same float cell size the device uses, one extra cell of safety is
    // not needed because E already contains the device's placement error
    auto range = [&](uint32_t j, int a, uint32_t& c0, uint32_t& c1) {
        const double p = a == 0 ? sph_exact[j].x : a == 1 ? sph_exact[j].y : sph_exact[j].z;
        const double r0 = std::fabs((double)sph_exact[j].w), r = r0 + margin(r0);
        const double f0 = std::floor((p - r - lo[a]) / (double)csf), f1 = std::floor((p + r - lo[a]) / (double)csf);
        c0 = (uint32_t)std::min(std::max(f0, 0.0), (double)dim[a] - 1.0);
        c1 = (uint32_t)std::min(std::max(f1, 0.0), (double)dim[a] - 1.0);
    };
    uint64_t total = 0;
    for (uint32_t j = 0; j < n; ++j) {
        uint32_t x0, x1, y0, y1, z0, z1;
        range(j, 0, x0, x1);
        range(j, 1, y0, y1);
        range(j, 2, z0, z1);
        total += (uint64_t)(x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1);
        if (total > (uint64_t)16 * n + 4096) return out;  // a few huge spheres would sit in every cell: not worth a grid
        for (uint32_t z = z0; z <= z1; ++z)
            for (uint32_t y = y0; y <= y1; ++y)
                for (uint32_t x = x0; x <= x1; ++x) ++count[((uint64_t)z * dim[1] + y) * dim[0] + x];
    }
    if (total >= (1u << 20)) return out;
    // The device's array has a shell of kGridOutside records around the grid's cells (rt_grid.h: a ray that steps into one has left
    // the grid); cell (x, y, z) is at shelled(k).
    const uint32_t rowx = dim[0] + 2, rowy = dim[1] + 2;
    auto shelled = [&](uint64_t k) {
        const uint64_t x = k % dim[0], y = k / dim[0] % dim[1], z = k / dim[0] / dim[1];
        return ((z + 1) * rowy + y + 1) * rowx + x + 1;
    };
    out.cells.assign((size_t)rowx * rowy * (dim[2] + 2), kGridOutside);
    uint32_t run = 0;
    for (uint64_t k = 0; k < n_cells; ++k) {
        if (count[k] > 4094) return GridHost{};  // 4095 entries at offset 0xfffff would read as kGridOutside
        out.cells[shelled(k)] = run | (count[k] << 20);
        run += count[k];
        count[k] = 0;
    }
    out.items.assign(total + kGridItemPad, 0);  // the walk reads up to kGridItemPad entries past a cell's list (and drops them)
    for (uint32_t j = 0; j < n; ++j) {  // ascending j inside every cell
        uint32_t x0, x1, y0, y1, z0, z1;
        range(j, 0, x0, x1);
        range(j, 1, y0, y1);
        range(j, 2, z0, z1);
        const uint16_t item = (uint16_t)(j | (sph_box[j].w != 0.0f ? 0x8000u : 0u));
        for (uint32_t z = z0; z <= z1; ++z)
            for (uint32_t y = y0; y <= y1; ++y)
                for (uint32_t x = x0; x <= x1; ++x) {
                    const uint64_t k = ((uint64_t)z * dim[1] + y) * dim[0] + x;
                    out.items[(out.cells[shelled(k)] & 0xfffffu) + count[k]++] = item;
                }
    }
    DevGrid& g = out.g;
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = (float)lo[a];
        g.hi[a] = (float)(lo[a] + (double)csf * dim[a]);
        if ((double)g.hi[a] < hi[a]) g.hi[a] = std::nextafter((float)hi[a], std::numeric_limits<float>::infinity());
        g.dim[a] = dim[a];
        g.centre[a] = (float)c[a];
    }
    g.cell = csf;
    g.inv_cell = inv;
    g.radius = (float)(radius0 * 1.01);
    g.origin_r2 = (float)(4.0 * radius0 * radius0);  // |o - centre| <= 2 * radius0
    g.slack_k = 2e-3f;
    g.n_cells = (uint32_t)out.cells.size();  // what the device holds: shell and spare entries included
    g.n_items = (uint32_t)out.items.size();
    g.enabled = 1;
    return out;
}

// Clusters of the scan-path entries for the explicit-ray kernel's bundle cull (rt_paths.hip): the entries in Morton order of
// their centres, kClusterSize at a time, each group with a bounding sphere {c, R}: R >= |p_j - c| + r_j for every member j.
// The cull tests the cluster bounds first (64 per row) and only the members of those that pass.  Tables of up to kClusterMaxEntries
// entries (48 bytes of LDS per cluster) with finite data only (the caller builds clusters when the grid exists).
struct ClusterHost {
    std::vector<float4> bound;    // [n_clu]
    std::vector<uint16_t> items;  // [n_clu * kClusterSize], 0xffff = none
};
inline ClusterHost build_clusters(const std::vector<float4>& sph_exact, uint32_t n)
{
    ClusterHost out;
    if (n == 0 || n > kClusterMaxEntries) return out;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t j = 0; j < n; ++j) {
        const double p[3] = {sph_exact[j].x, sph_exact[j].y, sph_exact[j].z};
        for (int a = 0; a < 3; ++a) {
            if (!std::isfinite(p[a]) || !std::isfinite((double)sph_exact[j].w)) return out;
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    auto spread = [](uint32_t v) {  // 10 bits -> every third bit
        v &= 0x3ffu;
        v = (v | (v << 16)) & 0x030000ffu;
        v = (v | (v << 8)) & 0x0300f00fu;
        v = (v | (v << 4)) & 0x030c30c3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    // one scale for the three axes, so that Morton cells are cubes
    const double ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), std::max(hi[2] - lo[2], 1e-300));
    std::vector<std::pair<uint32_t, uint32_t>> order(n);
    for (uint32_t j = 0; j < n; ++j) {
        const double p[3] = {sph_exact[j].x, sph_exact[j].y, sph_exact[j].z};
        uint32_t code = 0;
        for (int a = 0; a < 3; ++a) {
            const double f = (p[a] - lo[a]) / ext * 1023.0;
            code |= spread((uint32_t)std::min(std::max(f, 0.0), 1023.0)) << a;
        }
        order[j] = {code, j};
    }
    std::sort(order.begin(), order.end());
    const uint32_t n_clu = (n + kClusterSize - 1) / kClusterSize;
    out.bound.resize(n_clu);
    out.items.assign((size_t)n_clu * kClusterSize, (uint16_t)0xffffu);
    for (uint32_t c = 0; c < n_clu; ++c) {
        const uint32_t k0 = c * kClusterSize, k1 = std::min(n, k0 + kClusterSize);
        double ctr[3] = {0, 0, 0};
        for (uint32_t k = k0; k < k1; ++k) {
            const float4& e = sph_exact[order[k].second];
            ctr[0] += e.x;
            ctr[1] += e.y;
            ctr[2] += e.z;
        }
        for (int a = 0; a < 3; ++a) ctr[a] /= (double)(k1 - k0);
        const float cf[3] = {(float)ctr[0], (float)ctr[1], (float)ctr[2]};  // the centre the device sees
        double R = 0.0;
        for (uint32_t k = k0; k < k1; ++k) {
            const float4& e = sph_exact[order[k].second];
            const double dx = (double)e.x - cf[0], dy = (double)e.y - cf[1], dz = (double)e.z - cf[2];
            R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)e.w));
            out.items[(size_t)c * kClusterSize + (k - k0)] = (uint16_t)order[k].second;
        }
        float Rf = (float)(R * (1.0 + 1e-6));
        if ((double)Rf < R) Rf = std::nextafter(Rf, std::numeric_limits<float>::infinity());
        out.bound[c] = make_float4(cf[0], cf[1], cf[2], Rf);
    }
    return out;
}

}  // namespace host
}  // namespace rt
