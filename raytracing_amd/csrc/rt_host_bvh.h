// rt_host_bvh.h — host-side BVH build for triangle meshes (extension).  Median split on the longest
// centroid axis twice per node (4-wide nodes), leaves of <= 4 triangles, each node carrying the boxes of its children.  A hit must lie inside its triangle's padded box (part of the mesh
// semantics), node boxes contain those boxes, so the traversal never excludes a triangle the exact test would accept.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <numeric>
#include <vector>

#include "rt_internal.h"

namespace rt {
namespace host {

struct BvhBuild {
    std::vector<BvhNode>& nodes;  // global arrays (all meshes)
    std::vector<TriDev>& tris;
    const RtMesh& m;
    std::vector<uint32_t> order;  // triangle permutation of this mesh
    std::vector<float> cx, cy, cz;

    static float bits(uint32_t u)
    {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    const float* vert(uint32_t tri, int corner) const { return m.vertices + (size_t)m.indices[(size_t)tri * 3 + corner] * 3; }

    // One axis of a triangle's padded box, the same f32 expression as tri_box_holds (rt_complex.h) and the oracle.
    static void tri_axis(float a, float b1, float c1, float& lo, float& hi)
    {
        const float e1 = b1 - a, e2 = c1 - a;
        const float b = a + e1, c = a + e2;
        lo = std::fmin(a, std::fmin(b, c));
        hi = std::fmax(a, std::fmax(b, c));
        const float mag = std::fmax(std::fabs(lo), std::fabs(hi));
        const float pad = 1e-5f * (mag + (hi - lo)) + 1e-6f;
        lo = lo - pad;
        hi = hi + pad;
    }

    // Union of the triangles' padded boxes, padded once more for the rounding of the slab test itself.
    void bounds(size_t b, size_t e, float lo[3], float hi[3]) const
    {
        for (int c = 0; c < 3; ++c) {
            lo[c] = INFINITY;
            hi[c] = -INFINITY;
        }
        for (size_t k = b; k < e; ++k) {
            const float *v0 = vert(order[k], 0), *v1 = vert(order[k], 1), *v2 = vert(order[k], 2);
            for (int c = 0; c < 3; ++c) {
                float l, h;
                tri_axis(v0[c], v1[c], v2[c], l, h);
                lo[c] = std::min(lo[c], l);
                hi[c] = std::max(hi[c], h);
            }
        }
        for (int c = 0; c < 3; ++c) {
            const float mag = std::max(std::fabs(lo[c]), std::fabs(hi[c]));
            const float pad = 1e-5f * (mag + (hi[c] - lo[c])) + 1e-6f;
            lo[c] -= pad;
            hi[c] += pad;
        }
    }

    uint32_t make_leaf(size_t b, size_t e)
    {
        const uint32_t first = (uint32_t)tris.size();
        for (size_t k = b; k < e; ++k) {
            const uint32_t t = order[k];
            const float *a = vert(t, 0), *p1 = vert(t, 1), *p2 = vert(t, 2);
            TriDev td{};
            td.v0 = make_float4(a[0], a[1], a[2], bits(t));
            td.e1 = make_float4(p1[0] - a[0], p1[1] - a[1], p1[2] - a[2], 0.f);
            td.e2 = make_float4(p2[0] - a[0], p2[1] - a[1], p2[2] - a[2], 0.f);
            tris.push_back(td);
        }
        return kBvhLeaf | (first << 3) | (uint32_t)(e - b);
    }

    // Median split of order[b, e) on its longest centroid axis; returns the split position.
    size_t split(size_t b, size_t e)
    {
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = b; k < e; ++k) {
            const float c[3] = {cx[order[k]], cy[order[k]], cz[order[k]]};
            for (int a = 0; a < 3; ++a) {
                clo[a] = std::min(clo[a], c[a]);
                chi[a] = std::max(chi[a], c[a]);
            }
        }
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const std::vector<float>& key = axis == 0 ? cx : (axis == 1 ? cy : cz);
        const size_t mid = b + (e - b) / 2;
        std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e, [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
        return mid;
    }

    // Builds the tree of order[0, n) (more than one leaf's worth), breadth first so that the top levels — the nodes every
    // ray reads — sit at the front of the array, where the render kernel's LDS copy takes them from.  Per node two levels
    // of median splits give up to four parts; a part that fits a leaf becomes one, the others are queued.  Each level
    // quarters the range: depth <= 14 for the 2^28 triangles allowed.
    uint32_t build_all(size_t n)
    {
        struct Job {
            uint32_t self;
            size_t b, e;
        };
        std::deque<Job> todo;
        const uint32_t root = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{});
        todo.push_back({root, 0, n});
        while (!todo.empty()) {
            const Job j = todo.front();
            todo.pop_front();
            size_t cut[5];
            int parts = 0;
            const size_t mid = split(j.b, j.e);
            cut[parts++] = j.b;
            if (mid - j.b > (size_t)kBvhLeafTris) cut[parts++] = split(j.b, mid);
            cut[parts++] = mid;
            if (j.e - mid > (size_t)kBvhLeafTris) cut[parts++] = split(mid, j.e);
            cut[parts] = j.e;
            BvhNode nd = dead_node();
            for (int i = 0; i < parts; ++i) {
                float lo[3], hi[3];
                bounds(cut[i], cut[i + 1], lo, hi);
                uint32_t ref;
                if (cut[i + 1] - cut[i] <= (size_t)kBvhLeafTris) {
                    ref = make_leaf(cut[i], cut[i + 1]);
                } else {
                    ref = (uint32_t)nodes.size();
                    nodes.push_back(BvhNode{});
                    todo.push_back({ref, cut[i], cut[i + 1]});
                }
                nd.lo[i] = make_float4(lo[0], lo[1], lo[2], bits(ref));
                nd.hi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
            }
            nodes[j.self] = nd;
        }
        return root;
    }

    // A node none of whose children can be hit (NaN bounds): the root of an empty mesh, and the blank every node starts from.
    static BvhNode dead_node()
    {
        const float nan = std::numeric_limits<float>::quiet_NaN();
        BvhNode n{};
        for (int i = 0; i < 4; ++i) n.lo[i] = n.hi[i] = make_float4(nan, nan, nan, 0.f);
        return n;
    }
};

// Appends the BVH of mesh `m` to the global arrays; returns the root node index.
inline uint32_t build_mesh_bvh(const RtMesh& m, std::vector<BvhNode>& nodes, std::vector<TriDev>& tris)
{
    BvhBuild bb{nodes, tris, m, {}, {}, {}, {}};
    const size_t n = (size_t)m.n_triangles;
    bb.order.resize(n);
    std::iota(bb.order.begin(), bb.order.end(), 0u);
    bb.cx.resize(n);
    bb.cy.resize(n);
    bb.cz.resize(n);
    for (size_t t = 0; t < n; ++t) {
        const float *a = bb.vert((uint32_t)t, 0), *b = bb.vert((uint32_t)t, 1), *c = bb.vert((uint32_t)t, 2);
        bb.cx[t] = (a[0] + b[0] + c[0]) * (1.0f / 3.0f);
        bb.cy[t] = (a[1] + b[1] + c[1]) * (1.0f / 3.0f);
        bb.cz[t] = (a[2] + b[2] + c[2]) * (1.0f / 3.0f);
    }
    if (n > (size_t)kBvhLeafTris) return bb.build_all(n);
    BvhNode root = BvhBuild::dead_node();  // 0..4 triangles: a root whose only child (if any) is that leaf
    if (n) {
        float lo[3], hi[3];
        bb.bounds(0, n, lo, hi);
        root.lo[0] = make_float4(lo[0], lo[1], lo[2], BvhBuild::bits(bb.make_leaf(0, n)));
        root.hi[0] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    nodes.push_back(root);
    return (uint32_t)nodes.size() - 1;
}

}  // namespace host
}  // namespace rt
