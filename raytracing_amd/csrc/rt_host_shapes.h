// rt_host_shapes.h — host side of the complex-object path: the Transform matrices of an object
// (object_config.rs:24-28, transform.rs:11-17) in glam 0.29's f32 operation order, and the expansion of every
// ShapeConfig arm (shape_config.rs:53-149, wedge.rs:10-78, lens.rs:11-76) into a post-order CxNode program.
// Compiled with -ffp-contract=off like everything else.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rt_internal.h"

namespace rt {
namespace host {

struct Mat4 {
    float c[16];  // column-major: c[col * 4 + row]
};

inline Mat4 mat_mul(const Mat4& a, const Mat4& b)
{
    // glam mul_mat4: column j = a.x_axis * b[j].x + a.y_axis * b[j].y + a.z_axis * b[j].z + a.w_axis * b[j].w (left to right)
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float v = a.c[0 * 4 + i] * b.c[j * 4 + 0];
            v = v + a.c[1 * 4 + i] * b.c[j * 4 + 1];
            v = v + a.c[2 * 4 + i] * b.c[j * 4 + 2];
            v = v + a.c[3 * 4 + i] * b.c[j * 4 + 3];
            r.c[j * 4 + i] = v;
        }
    return r;
}

// glam Mat4::inverse (cofactor expansion, GLM lineage).
inline Mat4 mat_inverse(const Mat4& m)
{
    const float m00 = m.c[0], m01 = m.c[1], m02 = m.c[2], m03 = m.c[3];
    const float m10 = m.c[4], m11 = m.c[5], m12 = m.c[6], m13 = m.c[7];
    const float m20 = m.c[8], m21 = m.c[9], m22 = m.c[10], m23 = m.c[11];
    const float m30 = m.c[12], m31 = m.c[13], m32 = m.c[14], m33 = m.c[15];
    const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    const float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    const float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    const float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    const float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
    const float vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
    const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    Mat4 inv;
    for (int i = 0; i < 4; ++i) {
        const float i0 = (vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i];
        const float i1 = (vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i];
        const float i2 = (vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i];
        const float i3 = (vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i];
        inv.c[0 * 4 + i] = i0 * sa[i];
        inv.c[1 * 4 + i] = i1 * sb[i];
        inv.c[2 * 4 + i] = i2 * sa[i];
        inv.c[3 * 4 + i] = i3 * sb[i];
    }
    const float d0 = m.c[0] * inv.c[0], d1 = m.c[1] * inv.c[4], d2 = m.c[2] * inv.c[8], d3 = m.c[3] * inv.c[12];
    const float det = ((d0 + d1) + d2) + d3;
    const float rcp = 1.0f / det;
    for (int k = 0; k < 16; ++k) inv.c[k] = inv.c[k] * rcp;
    return inv;
}

// Transform matrix of an ObjectConfig: T(position) * R_y(rotation_y_deg.to_radians())  (object_config.rs:24-26).
inline void object_matrices(const RtObject& o, Mat4& m, Mat4& minv)
{
    const float rads_per_deg = 3.14159265358979323846f / 180.0f;
    const float ang = o.rotation_y_deg * rads_per_deg;
    const float s = std::sin(ang), c = std::cos(ang);  // glam resolves sin_cos through the libm crate: parity unpinned for angle != 0
    Mat4 t = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, o.position[0], o.position[1], o.position[2], 1}};
    Mat4 r = {{c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1}};
    m = mat_mul(t, r);
    minv = mat_inverse(m);
}

struct Program {
    std::vector<CxNode> nodes;  // post-order
    int max_hits = 0;           // largest list any node can produce
    int max_stack = 0;          // deepest list stack of the evaluation
};

inline CxNode leaf(uint32_t kind, float ax, float ay, float az, float aw, float bx, float by, float bz)
{
    CxNode n{};
    n.kind = kind;
    n.a = make_float4(ax, ay, az, aw);
    n.b = make_float4(bx, by, bz, 0.f);
    return n;
}

// Appends the post-order program of one shape; returns the hit bound of its result list, or -1 on a malformed tree.
// `depth` is the list-stack height before this subtree runs.
inline int emit_shape(uint32_t shape, const float* sp, const RtShapeNode* tree, uint64_t n_tree, Program& pr, int depth, int guard, std::string& err);

inline int emit_op(uint32_t op, Program& pr, int hl, int hr)
{
    CxNode n{};
    n.kind = op;
    pr.nodes.push_back(n);
    const int h = hl + hr;
    if (h > pr.max_hits) pr.max_hits = h;
    return h;
}
inline int emit_leaf(const CxNode& n, int bound, Program& pr, int depth)
{
    pr.nodes.push_back(n);
    if (bound > pr.max_hits) pr.max_hits = bound;
    if (depth + 1 > pr.max_stack) pr.max_stack = depth + 1;
    return bound;
}

inline int emit_tree_node(const RtShapeNode* tree, uint64_t n_tree, uint32_t idx, Program& pr, int depth, int guard, std::string& err)
{
    if (idx >= n_tree) {
        err = "shape node index out of range";
        return -1;
    }
    if (guard > 64) {
        err = "shape tree too deep (or cyclic)";
        return -1;
    }
    const RtShapeNode& nd = tree[idx];
    if (nd.shape >= RT_SHAPE_UNION && nd.shape <= RT_SHAPE_DIFFERENCE) {
        const int hl = emit_tree_node(tree, n_tree, nd.a, pr, depth, guard + 1, err);
        if (hl < 0) return -1;
        const int hr = emit_tree_node(tree, n_tree, nd.b, pr, depth + 1, guard + 1, err);
        if (hr < 0) return -1;
        return emit_op(nd.shape == RT_SHAPE_UNION ? CX_UNION : (nd.shape == RT_SHAPE_INTERSECTION ? CX_INTERSECTION : CX_DIFFERENCE), pr, hl, hr);
    }
    if (nd.shape == RT_SHAPE_TREE) {
        err = "RT_SHAPE_TREE is not valid inside a shape node";
        return -1;
    }
    return emit_shape(nd.shape, nd.shape_param, tree, n_tree, pr, depth, guard + 1, err);
}

inline int emit_shape(uint32_t shape, const float* sp, const RtShapeNode* tree, uint64_t n_tree, Program& pr, int depth, int guard, std::string& err)
{
    const float rads_per_deg = 3.14159265358979323846f / 180.0f;
    switch (shape) {
        case RT_SHAPE_SPHERE: return emit_leaf(leaf(CX_SPHERE, 0, 0, 0, sp[0], 0, 0, 0), 2, pr, depth);
        case RT_SHAPE_PLANE: return emit_leaf(leaf(CX_PLANE, 0, 0, 0, 0, sp[0], sp[1], sp[2]), 1, pr, depth);
        case RT_SHAPE_BOX: {  // shape_config.rs:61-68
            const float hx = sp[0] / 2.0f, hy = sp[1] / 2.0f, hz = sp[2] / 2.0f;
            return emit_leaf(leaf(CX_BOX, -hx, -hy, -hz, 0, hx, hy, hz), 2, pr, depth);
        }
        case RT_SHAPE_CYLINDER: {  // shape_config.rs:74-103
            const float hh = sp[0] / 2.0f;
            int h = emit_leaf(leaf(CX_INFCYL, 0, 0, 0, sp[1], 0, 1, 0), 2, pr, depth);
            h = emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, hh, 0, 0, 0, -1, 0), 1, pr, depth + 1));
            return emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, -hh, 0, 0, 0, 1, 0), 1, pr, depth + 1));
        }
        case RT_SHAPE_CONE: {  // shape_config.rs:104-120, infinite_cone.rs:16-23
            const float cs = std::cos(sp[0] * rads_per_deg);
            const float len = std::sqrt((0.0f * 0.0f + 1.0f * 1.0f) + 0.0f * 0.0f), rcp = 1.0f / len;  // Vec3::Y.normalize()
            int h = emit_leaf(leaf(CX_INFCONE, 0, 0, 0, cs * cs, 0.0f * rcp, 1.0f * rcp, 0.0f * rcp), 2, pr, depth);
            return emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, sp[1], 0, 0, 0, -1, 0), 1, pr, depth + 1));
        }
        case RT_SHAPE_WEDGE: {  // wedge.rs:10-78
            const float half_depth = sp[2] / 2.0f;
            const float ang = sp[3] * rads_per_deg;
            const float ac = std::cos(ang), as = std::sin(ang);
            int h = emit_leaf(leaf(CX_PLANE, 0, 0, 0, 0, 0, 1, 0), 1, pr, depth);
            h = emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, 0, 0, 0, 1, 0, 0), 1, pr, depth + 1));
            h = emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, 0, 0, 0, -as, ac, 0.0f), 1, pr, depth + 1));
            h = emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, 0, half_depth, 0, 0, 0, -1), 1, pr, depth + 1));
            return emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_PLANE, 0, 0, -half_depth, 0, 0, 0, 1), 1, pr, depth + 1));
        }
        case RT_SHAPE_LENS: {  // lens.rs:11-76
            const float ht = sp[0] / 2.0f, r1 = sp[2], r2 = sp[3];
            int h1 = std::isfinite(r1) ? emit_leaf(leaf(CX_SPHERE, 0, 0, -ht + r1, std::fabs(r1), 0, 0, 0), 2, pr, depth)
                                       : emit_leaf(leaf(CX_PLANE, 0, 0, -ht, 0, 0, 0, 1), 1, pr, depth);
            int h2 = std::isfinite(r2) ? emit_leaf(leaf(CX_SPHERE, 0, 0, ht + r2, std::fabs(r2), 0, 0, 0), 2, pr, depth + 1)
                                       : emit_leaf(leaf(CX_PLANE, 0, 0, ht, 0, 0, 0, -1), 1, pr, depth + 1);
            int h = emit_op(CX_INTERSECTION, pr, h1, h2);
            return emit_op(CX_INTERSECTION, pr, h, emit_leaf(leaf(CX_INFCYL, 0, 0, 0, sp[1] / 2.0f, 0, 0, 1), 2, pr, depth + 1));
        }
        case RT_SHAPE_TREE: {
            uint32_t root;
            std::memcpy(&root, &sp[0], 4);
            return emit_tree_node(tree, n_tree, root, pr, depth, guard + 1, err);
        }
        default: err = "unknown shape code " + std::to_string(shape); return -1;
    }
}

}  // namespace host
}  // namespace rt
