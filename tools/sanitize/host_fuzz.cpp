// host_fuzz.cpp — AddressSanitizer + UBSan exercise of the hand-written host code (the build's sanitizer run).
// CPU only; never built for or run on the GPU.  Covers what parses untrusted text or builds index structures:
//   raytracing_amd/host/toml_lite.h, scene_config.h       (TOML reader, SceneConfig flatten)
//   raytracing_amd/csrc/rt_host_grid.h, rt_host_bvh.h, rt_host_shapes.h   (uniform grid, BVH, shape programs / Mat4)
// usage: host_fuzz N_MUTATIONS file.toml...   — parses every file as given, then N mutated copies of each (byte flips, cuts,
// duplicated spans, structural characters), then builds grids / BVHs / shape programs from random and degenerate inputs.
// Any sanitizer report aborts (-fno-sanitize-recover); exceptions of the documented kinds are the expected outcome of bad input.
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../raytracing_amd/host/scene_config.h"
#include "../../raytracing_amd/csrc/rt_host_bvh.h"
#include "../../raytracing_amd/csrc/rt_host_grid.h"
#include "../../raytracing_amd/csrc/rt_host_shapes.h"

static std::mt19937_64 rng(0xC0FFEE);
static uint32_t rnd(uint32_t n) { return n ? (uint32_t)(rng() % n) : 0u; }
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)((rng() >> 11) * (1.0 / 9007199254740992.0)); }

static long g_ok = 0, g_rejected = 0;
static void try_config(const std::string& text)
{
    try {
        rtcfg::SimulationConfig cfg = rtcfg::from_toml(text);
        // walk what was built the way main.cpp does: every node index must be in range, every object a known shape
        for (const RtObject& o : cfg.objects) {
            std::string err;
            rt::host::Program pr;
            rt::host::Mat4 m, mi;
            rt::host::object_matrices(o, m, mi);
            (void)rt::host::emit_shape(o.shape, o.shape_param, cfg.nodes.data(), cfg.nodes.size(), pr, 0, 0, err);
        }
        ++g_ok;
    } catch (const rtcfg::ConfigError&) {
        ++g_rejected;
    } catch (const std::runtime_error&) {  // toml_lite's parse errors
        ++g_rejected;
    }
}

static std::string mutate(const std::string& src)
{
    std::string s = src;
    static const char structural[] = "[]{}=.,\"'#\n\t -+:tf";  // no digits / exponents: counts stay the size the file gave them
    const int n_edits = 1 + (int)rnd(4);
    for (int e = 0; e < n_edits && !s.empty(); ++e) {
        const size_t pos = rnd((uint32_t)s.size());
        switch (rnd(6)) {
            case 0: {  // flip a byte to a structural character (digits only to digits)
                const char c = s[pos];
                s[pos] = (c >= '0' && c <= '9') ? (char)('0' + rnd(10)) : structural[rnd(sizeof structural - 1)];
                break;
            }
            case 1: s.erase(pos, 1 + rnd(8)); break;
            case 2: s.insert(pos, 1, structural[rnd(sizeof structural - 1)]); break;
            case 3: s.resize(pos); break;  // truncated file
            case 4: {  // duplicate a span (repeated keys, repeated tables)
                const size_t len = 1 + rnd(120);
                s.insert(pos, s.substr(pos, len));
                break;
            }
            default: s[pos] = (char)rnd(256); break;  // any byte, NUL and UTF-8 fragments included
        }
    }
    return s;
}

static void fuzz_grids()
{
    for (int it = 0; it < 300; ++it) {
        const uint32_t n = it < 6 ? (uint32_t[]){0u, 1u, 95u, 96u, 97u, 4000u}[it] : 96u + rnd(3000);
        std::vector<float4> ex(n + 4), bx(n + 4);
        const float span = it % 7 == 0 ? 1e-3f : (it % 11 == 0 ? 1e6f : 50.0f);
        for (uint32_t j = 0; j < n; ++j) {
            ex[j] = make_float4(frand(-span, span), frand(-span, span), it % 5 == 0 ? 0.0f : frand(-span, span), frand(0.0f, span * 0.05f));
            bx[j] = rnd(9) == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (n && it % 13 == 0) ex[rnd(n)].x = std::numeric_limits<float>::quiet_NaN();
        if (n && it % 17 == 0) ex[rnd(n)].w = std::numeric_limits<float>::infinity();
        if (n && it % 19 == 0) ex[rnd(n)].w = span * 40.0f;  // one giant among the small: long item lists
        if (n && it % 23 == 0) for (uint32_t j = 0; j < n; ++j) ex[j] = ex[0];  // all coincident
        const rt::host::GridHost g = rt::host::build_grid(ex, bx, n);
        if (g.g.enabled) {
            // the array the device walks: (dim + 2)^3 records, the outermost layer "outside", every other cell's item range -- plus the
            // spare entries a walk pass reads past it -- inside the item array
            const uint32_t rx = g.g.dim[0] + 2, ry = g.g.dim[1] + 2, rz = g.g.dim[2] + 2;
            if (g.cells.size() != (size_t)rx * ry * rz || g.g.n_cells != g.cells.size() || g.g.n_items != g.items.size()) {
                std::fprintf(stderr, "grid array sizes do not match the dimensions\n");
                std::abort();
            }
            for (uint32_t z = 0; z < rz; ++z)
                for (uint32_t y = 0; y < ry; ++y)
                    for (uint32_t x = 0; x < rx; ++x) {
                        const uint32_t c = g.cells[((size_t)z * ry + y) * rx + x];
                        const bool shell = x == 0 || y == 0 || z == 0 || x == rx - 1 || y == ry - 1 || z == rz - 1;
                        if (shell != (c == rt::kGridOutside)) {
                            std::fprintf(stderr, "grid shell record misplaced\n");
                            std::abort();
                        }
                        const uint32_t first = c & 0xfffffu, cnt = c >> 20;
                        if (!shell && (size_t)first + cnt + rt::kGridItemPad > g.items.size()) {
                            std::fprintf(stderr, "grid cell range out of bounds\n");
                            std::abort();
                        }
                    }
            for (uint16_t i : g.items)
                if ((uint32_t)(i & 0x7fffu) >= n) {
                    std::fprintf(stderr, "grid item index out of range\n");
                    std::abort();
                }
        }
    }
}

static void fuzz_bvh()
{
    for (int it = 0; it < 200; ++it) {
        const uint32_t nt = it < 8 ? (uint32_t)it : 1u + rnd(3000);
        const uint32_t nv = 3 + rnd(nt + 8);
        std::vector<float> v((size_t)nv * 3);
        for (float& x : v) x = it % 5 == 0 ? 1.0f : frand(-100.f, 100.f);  // every fifth mesh: all vertices coincide
        if (it % 7 == 0) v[rnd(nv * 3)] = std::numeric_limits<float>::quiet_NaN();
        if (it % 9 == 0) v[rnd(nv * 3)] = std::numeric_limits<float>::infinity();
        std::vector<uint32_t> idx((size_t)nt * 3);
        for (uint32_t& i : idx) i = rnd(nv);
        RtMesh m{v.data(), nv, idx.data(), nt};
        std::vector<rt::BvhNode> nodes;
        std::vector<rt::TriDev> tris;
        const uint32_t root = rt::host::build_mesh_bvh(m, nodes, tris);
        if (root >= nodes.size() || tris.size() != nt) {
            std::fprintf(stderr, "bvh: bad root or triangle count\n");
            std::abort();
        }
        for (const rt::BvhNode& n : nodes)
            for (int c = 0; c < 4; ++c) {
                uint32_t ref;
                std::memcpy(&ref, &n.lo[c].w, 4);
                if (n.lo[c].x != n.lo[c].x) continue;  // absent child
                if (ref & rt::kBvhLeaf) {
                    const uint32_t first = (ref & 0x7fffffffu) >> 3, cnt = ref & 7u;
                    if (cnt == 0 || cnt > 4 || (size_t)first + cnt > tris.size()) {
                        std::fprintf(stderr, "bvh: leaf range out of bounds\n");
                        std::abort();
                    }
                } else if (ref >= nodes.size()) {
                    std::fprintf(stderr, "bvh: child index out of range\n");
                    std::abort();
                }
            }
    }
}

static void fuzz_shapes()
{
    for (int it = 0; it < 20000; ++it) {
        const uint32_t n = rnd(12);
        std::vector<RtShapeNode> nodes(n);
        for (RtShapeNode& nd : nodes) {
            const uint32_t kinds[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 17, 18, 99};
            nd.shape = kinds[rnd(13)];
            nd.a = rnd(n + 2);  // may point past the array, at itself, or form a cycle
            nd.b = rnd(n + 2);
            for (float& p : nd.shape_param) p = rnd(10) == 0 ? std::numeric_limits<float>::infinity() : frand(-5.f, 5.f);
        }
        RtObject o{};
        o.shape = rnd(4) == 0 ? 7u /* tree */ : rnd(9);
        for (float& p : o.shape_param) p = frand(-5.f, 5.f);
        if (o.shape == 7u) {
            const uint32_t root = rnd(n + 2);
            std::memcpy(&o.shape_param[0], &root, 4);
        }
        o.rotation_y_deg = frand(-720.f, 720.f);
        for (float& p : o.position) p = frand(-1e3f, 1e3f);
        rt::host::Mat4 m, mi;
        rt::host::object_matrices(o, m, mi);
        std::string err;
        rt::host::Program pr;
        (void)rt::host::emit_shape(o.shape, o.shape_param, nodes.data(), nodes.size(), pr, 0, 0, err);
    }
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: host_fuzz N_MUTATIONS file.toml...\n");
        return 2;
    }
    const long n_mut = std::atol(argv[1]);
    for (int a = 2; a < argc; ++a) {
        std::ifstream f(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        try_config(text);
        for (long k = 0; k < n_mut; ++k) try_config(mutate(text));
    }
    // hand-written malformed inputs (the cases tests/test_native_cli.py feeds the binary)
    const char* bad[] = {"", "[", "[[", "a = ", "a = [1, 2", "a = {b = 1", "[scene]\n[scene]\n", "a.b = 1\na = 2\n", "x = \"unterminated",
                         "[simulation_settings]\ninfinity_distance = 1e999\nmax_bounces = -1\n[scene]\n",
                         "[simulation_settings]\ninfinity_distance = 1\nmax_bounces = 99999999999999999999\n[scene]\n",
                         "[simulation_settings]\ninfinity_distance = 1\nmax_bounces = 1\n[scene]\n[[scene.ray_generators]]\ntype = \"Projector\"\ncount_u = 4000000000\ncount_v = 4000000000\ncurrent_ior = 1\norigin = [0.0, 0.0, 0.0]\ntarget_corner = [0.0, 0.0, 1.0]\ntarget_u = [1.0, 0.0, 0.0]\ntarget_v = [0.0, 1.0, 0.0]\n"};
    for (const char* b : bad) try_config(b);
    fuzz_grids();
    fuzz_bvh();
    fuzz_shapes();
    std::printf("host_fuzz: %ld configs accepted, %ld rejected with an error; grids, BVHs and shape programs built; no sanitizer report\n", g_ok, g_rejected);
    return 0;
}
