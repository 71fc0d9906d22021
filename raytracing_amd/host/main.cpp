// `raytracing` — native CLI on the HIP path, mirror of crates/raytracing_cli/src/cli.rs:7-33 (src/main.rs:5-7).
//
// Same invocation as the reference binary: no arguments, reads ./simulation.toml, writes ./dist/path_{i}.csv
// (header x,y,z; one row per polyline vertex; f32 printed like Rust's `f32::to_string()`), prints the same progress
// lines.  The Bevy window (cli.rs:15) is not part of the hot path and is omitted.  Like the reference it does not
// create ./dist.  Extras for harnesses only:  --config PATH  --out DIR  --seed N  --device N  --dump (print the
// flattened objects / nodes / rays as hex floats and exit without touching the GPU).
//
// Render-loop extension (no reference counterpart):
//   --render SPP [--generator K]   render ray generator K of the config with SPP paths per cell, write OUT/frame.pfm
//   --frames DIR                   with --render: one frame per *.toml in DIR (sorted by name; each an unchanged
//                                  simulation.toml, i.e. the per-frame scene update), frame f -> OUT/frame_%05d.pfm;
//                                  --rank R --world W (default: RANK / WORLD_SIZE of the environment, else 0 / 1)
//                                  deal frames round-robin, f = R, R+W, ...; ranks exchange nothing.
// PFM: "PF\n<count_u> <count_v>\n-1.0\n", rows bottom-up = j ascending, columns = i, RGB little-endian f32.
#include <algorithm>
#include <charconv>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <deque>
#include <dirent.h>
#include <fstream>
#include <future>
#include <iostream>
#include <sstream>

#include "../../include/rt_hip.h"
#include "scene_config.h"

namespace {

// Rust `impl Display for f32`: shortest digits that round-trip, positional notation (never an exponent), no trailing ".0".
// std::to_chars' shortest *scientific* form gives exactly those digits; they are then laid out positionally, padded with
// zeros on either side of the point (1234567936f32 -> "1234568000", 1e-7f -> "0.0000001"), as Rust does.
std::string format_f32(float x)
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "inf" : "-inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);
    std::string sci(buf, r.ptr);  // [-]d[.ddd]e[+-]XX
    std::string out;
    size_t pos = 0;
    if (sci[0] == '-') {
        out = "-";
        pos = 1;
    }
    const size_t e = sci.find('e');
    std::string digits;
    for (size_t k = pos; k < e; ++k)
        if (sci[k] != '.') digits += sci[k];
    const int exp10 = std::atoi(sci.c_str() + e + 1);  // value = d.ddd * 10^exp10
    const int n = (int)digits.size();
    if (digits == "0") return out + "0";
    if (exp10 >= n - 1) {  // integer: digits followed by zeros
        out += digits;
        out.append((size_t)(exp10 - (n - 1)), '0');
    } else if (exp10 >= 0) {  // point inside the digit string
        out += digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    } else {  // 0.000ddd
        out += "0.";
        out.append((size_t)(-exp10 - 1), '0');
        out += digits;
    }
    return out;
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

rtcfg::SimulationConfig load_config(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("No such file or directory: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return rtcfg::from_toml(ss.str());
}

// fb is cell-major (cell = i * count_v + j, scene_config.rs:74-75); the file wants rows of constant j.
bool write_pfm(const std::string& path, const std::vector<float>& fb, uint32_t cu, uint32_t cv)
{
    std::vector<float> rows((size_t)cu * cv * 3);
    for (uint32_t j = 0; j < cv; ++j)
        for (uint32_t i = 0; i < cu; ++i)
            std::memcpy(&rows[((size_t)j * cu + i) * 3], &fb[((size_t)i * cv + j) * 3], 12);
    std::ofstream w(path, std::ios::binary);
    if (!w) return false;
    w << "PF\n" << cu << ' ' << cv << "\n-1.0\n";
    w.write(reinterpret_cast<const char*>(rows.data()), (std::streamsize)(rows.size() * 4));
    return (bool)w.flush();
}

struct Frame {
    rtcfg::SimulationConfig cfg;
    std::string error;
};

Frame load_frame(const std::string& path)
{
    Frame f;
    try {
        f.cfg = load_config(path);
    } catch (const std::exception& e) {
        f.error = e.what();
    }
    return f;
}

// One frame: scene from the config's object list, generator `gen_index`, spp paths per cell.
int render_frame(const rtcfg::SimulationConfig& cfg, size_t gen_index, uint32_t spp, uint64_t seed, std::vector<float>& fb, uint32_t& cu,
                 uint32_t& cv, uint64_t& segments)
{
    if (gen_index >= cfg.generators.size()) {
        std::fprintf(stderr, "Error: the config has %zu ray generators, --generator %zu asked\n", cfg.generators.size(), gen_index);
        return 1;
    }
    const RtRayGenerator& g = cfg.generators[gen_index];
    rt_scene* scene = nullptr;
    if (rt_hip_scene_create_ex(cfg.objects.data(), cfg.objects.size(), cfg.nodes.empty() ? nullptr : cfg.nodes.data(), cfg.nodes.size(), &scene) != RT_OK)
        return fail_rt("rt_hip_scene_create_ex");
    cu = g.count_u;
    cv = g.count_v;
    const uint64_t cells = (uint64_t)cu * cv;
    fb.assign(cells * 3 + 1, 0.f);
    const int rc = rt_hip_render(scene, &g, spp, cfg.settings, seed, 0, cells, 1, fb.data(), nullptr, nullptr, nullptr, &segments);
    rt_hip_scene_destroy(scene);
    if (rc != RT_OK) return fail_rt("rt_hip_render");
    fb.resize(cells * 3);
    return 0;
}

int run_frames(const std::string& frames_dir, const std::string& out_dir, size_t gen_index, uint32_t spp, uint64_t seed, int rank, int world)
{
    std::vector<std::string> files;
    if (DIR* d = opendir(frames_dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() > 5 && n.compare(n.size() - 5, 5, ".toml") == 0) files.push_back(n);
        }
        closedir(d);
    } else {
        std::fprintf(stderr, "Error: cannot open directory %s\n", frames_dir.c_str());
        return 1;
    }
    std::sort(files.begin(), files.end());
    // the host side of frame f+world (TOML parse, flatten) and the image write of frame f-world overlap the GPU work of f
    std::future<Frame> next;
    std::deque<std::future<bool>> writing;  // up to kWriters images being reordered / written while the GPU renders on
    constexpr size_t kWriters = 3;
    auto drain = [&](size_t keep) {
        while (writing.size() > keep) {
            if (!writing.front().get()) return false;
            writing.pop_front();
        }
        return true;
    };
    size_t f = (size_t)rank;
    if (f < files.size()) next = std::async(std::launch::async, load_frame, frames_dir + "/" + files[f]);
    uint64_t total = 0;
    size_t done = 0;
    for (; f < files.size(); f += (size_t)world) {
        Frame fr = next.get();
        if (f + (size_t)world < files.size()) next = std::async(std::launch::async, load_frame, frames_dir + "/" + files[f + (size_t)world]);
        if (!fr.error.empty()) {
            std::fprintf(stderr, "Error: %s: %s\n", files[f].c_str(), fr.error.c_str());
            return 1;
        }
        std::vector<float> fb;
        uint32_t cu = 0, cv = 0;
        uint64_t segs = 0;
        if (int rc = render_frame(fr.cfg, gen_index, spp, seed, fb, cu, cv, segs)) return rc;
        total += segs;
        if (!drain(kWriters - 1)) {
            std::fprintf(stderr, "Error: cannot write an image into %s\n", out_dir.c_str());
            return 1;
        }
        char name[64];
        std::snprintf(name, sizeof name, "/frame_%05zu.pfm", f);
        writing.push_back(std::async(std::launch::async, [path = out_dir + name, fb = std::move(fb), cu, cv] { return write_pfm(path, fb, cu, cv); }));
        ++done;
    }
    if (!drain(0)) {
        std::fprintf(stderr, "Error: cannot write an image into %s\n", out_dir.c_str());
        return 1;
    }
    std::printf("rank %d/%d: %zu of %zu frames, %" PRIu64 " ray segments\n", rank, world, done, files.size(), total);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    std::string config = "simulation.toml", out_dir = "./dist";
    uint64_t seed = 0;
    int device = 0;
    bool dump_only = false;
    uint32_t render_spp = 0;
    size_t gen_index = 0;
    std::string frames_dir;
    int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
    int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--config" && i + 1 < argc) config = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
        else if (a == "--seed" && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 0);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--dump") dump_only = true;
        else if (a == "--format-f32") {  // diagnostic: print the CSV text of each following f32 bit pattern (hex), one per line
            for (++i; i < argc; ++i) {
                const uint32_t bits = (uint32_t)std::strtoul(argv[i], nullptr, 16);
                float f;
                std::memcpy(&f, &bits, 4);
                std::printf("%s\n", format_f32(f).c_str());
            }
            return 0;
        }
        else if (a == "--render" && i + 1 < argc) render_spp = (uint32_t)std::strtoul(argv[++i], nullptr, 0);
        else if (a == "--generator" && i + 1 < argc) gen_index = std::strtoul(argv[++i], nullptr, 0);
        else if (a == "--frames" && i + 1 < argc) frames_dir = argv[++i];
        else if (a == "--rank" && i + 1 < argc) rank = std::atoi(argv[++i]);
        else if (a == "--world" && i + 1 < argc) world = std::atoi(argv[++i]);
        else {
            std::fprintf(stderr, "usage: raytracing [--config PATH] [--out DIR] [--seed N] [--device N] [--dump]\n"
                                 "                  [--render SPP [--generator K] [--frames DIR [--rank R --world W]]]\n");
            return 2;
        }
    }
    if (world < 1 || rank < 0 || rank >= world || (!frames_dir.empty() && render_spp == 0)) {
        std::fprintf(stderr, "Error: need 0 <= rank < world, and --render SPP with --frames\n");
        return 2;
    }
    if (!frames_dir.empty()) {
        if (rt_hip_init(device) != RT_OK) return fail_rt("rt_hip_init");
        const int rc = run_frames(frames_dir, out_dir, gen_index, render_spp, seed, rank, world);
        rt_hip_shutdown();
        return rc;
    }
    std::printf("設定ファイル simulation.toml を読み込んでいます...\n");
    rtcfg::SimulationConfig cfg;
    try {
        cfg = load_config(config);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());  // cli.rs:12 `?` -> main returns Err
        return 1;
    }
    if (dump_only) {
        dump(cfg);
        return 0;
    }
    if (rt_hip_init(device) != RT_OK) return fail_rt("rt_hip_init");
    if (render_spp) {
        std::vector<float> fb;
        uint32_t cu = 0, cv = 0;
        uint64_t segs = 0;
        if (int rc = render_frame(cfg, gen_index, render_spp, seed, fb, cu, cv, segs)) return rc;
        const std::string path = out_dir + "/frame.pfm";
        if (!write_pfm(path, fb, cu, cv)) {
            std::fprintf(stderr, "Error: No such file or directory (os error 2): %s\n", path.c_str());
            return 1;
        }
        std::printf("%ux%u cells, %u spp, %" PRIu64 " ray segments -> '%s'\n", cu, cv, render_spp, segs, path.c_str());
        rt_hip_shutdown();
        return 0;
    }
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
