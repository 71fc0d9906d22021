// `raytracing` — native CLI on the HIP path, mirror of crates/raytracing_cli/src/cli.rs:7-33 (src/main.rs:5-7).
//
// Same invocation as the reference binary: no arguments, reads ./simulation.toml, writes ./dist/path_{i}.csv
// (header x,y,z; one row per polyline vertex; f32 printed like Rust's `f32::to_string()`), prints the same progress
// lines.  The Bevy window (cli.rs:15) is not part of the hot path and is omitted.  Like the reference it does not
// create ./dist.  Extras for harnesses only:  --config PATH  --out DIR  --seed N  --device N  --dump (print the
// flattened objects / nodes / rays as hex floats and exit without touching the GPU).
#include <charconv>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/rt_hip.h"
#include "scene_config.h"

namespace {

// Rust `impl Display for f32`: shortest digits that round-trip, positional notation, no trailing ".0".
std::string format_f32(float x)
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char buf[128];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

int fail_rt(const char* what)
{
    std::fprintf(stderr, "Error: %s: %s\n", what, rt_hip_last_error());
    return 1;
}

void dump(const rtcfg::SimulationConfig& cfg)
{
    std::printf("settings %a %u\n", cfg.settings.infinity_distance, cfg.settings.max_bounces);
    for (auto& o : cfg.objects) {
        uint32_t sp0;
        std::memcpy(&sp0, &o.shape_param[0], 4);
        std::printf("object %u %u %a %a %a %a %a %a %a %a %a %u\n", o.shape, o.material_kind, o.material_param, o.rotation_y_deg, o.position[0],
                    o.position[1], o.position[2], o.shape_param[0], o.shape_param[1], o.shape_param[2], o.shape_param[3], sp0);
    }
    for (auto& n : cfg.nodes)
        std::printf("node %u %u %u %a %a %a %a\n", n.shape, n.a, n.b, n.shape_param[0], n.shape_param[1], n.shape_param[2], n.shape_param[3]);
    for (auto& r : cfg.rays)
        std::printf("ray %a %a %a %a %a %a %a\n", r.origin[0], r.origin[1], r.origin[2], r.direction[0], r.direction[1], r.direction[2], r.current_ior);
}

}  // namespace

int main(int argc, char** argv)
{
    std::string config = "simulation.toml", out_dir = "./dist";
    uint64_t seed = 0;
    int device = 0;
    bool dump_only = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--config" && i + 1 < argc) config = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--dump") dump_only = true;
        else {
            std::fprintf(stderr, "usage: raytracing [--config PATH] [--out DIR] [--seed N] [--device N] [--dump]\n");
            return 2;
        }
    }
    std::printf("設定ファイル simulation.toml を読み込んでいます...\n");
    rtcfg::SimulationConfig cfg;
    try {
        std::ifstream f(config, std::ios::binary);
        if (!f) throw std::runtime_error("No such file or directory: " + config);
        std::stringstream ss;
        ss << f.rdbuf();
        cfg = rtcfg::from_toml(ss.str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());  // cli.rs:12 `?` -> main returns Err
        return 1;
    }
    if (dump_only) {
        dump(cfg);
        return 0;
    }
    if (rt_hip_init(device) != RT_OK) return fail_rt("rt_hip_init");
    rt_scene* scene = nullptr;
    if (rt_hip_scene_create_ex(cfg.objects.data(), cfg.objects.size(), cfg.nodes.empty() ? nullptr : cfg.nodes.data(), cfg.nodes.size(), &scene) != RT_OK)
        return fail_rt("rt_hip_scene_create_ex");
    const size_t n = cfg.rays.size(), stride = (size_t)cfg.settings.max_bounces + 1;
    std::vector<float> pts(n * stride * 3 + 1);
    std::vector<uint32_t> npts(n + 1);
    if (rt_hip_simulate(scene, cfg.rays.data(), n, cfg.settings, seed, 0, pts.data(), npts.data(), nullptr) != RT_OK) return fail_rt("rt_hip_simulate");
    rt_hip_scene_destroy(scene);
    for (size_t i = 0; i < n; ++i) {
        const std::string file_name = out_dir + "/path_" + std::to_string(i) + ".csv";
        std::ofstream w(file_name, std::ios::binary);
        if (!w) {
            std::fprintf(stderr, "Error: No such file or directory (os error 2): %s\n", file_name.c_str());  // csv::Writer::from_path, cli.rs:19
            return 1;
        }
        w << "x,y,z\n";
        for (uint32_t k = 0; k < npts[i]; ++k) {
            const float* p = &pts[(i * stride + k) * 3];
            w << format_f32(p[0]) << ',' << format_f32(p[1]) << ',' << format_f32(p[2]) << '\n';
        }
        w.flush();
        std::printf("光路 %zu を '%s' に出力しました。\n", i, file_name.c_str());
    }
    rt_hip_shutdown();
    return 0;
}
