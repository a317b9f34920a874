// Test driver for include/rayn_host.hpp (the C++ host mirror).
//   host_mirror desc W H volumes out.bin                -> raw rayn_world_desc of setup::setup()   (no GPU needed)
//   host_mirror render W H SAMPLES BOUNCES volumes out.bin [n_devices] -> film: color(3n) alpha(n) background(3n) normal(3n) floats
//       n_devices > 0: a multi-device Film over n_devices entries of GPU 0 (rayn_hip_create_multi)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../include/rayn_host.hpp"

int main(int argc, char** argv) {
    using namespace rayn;
    if (argc < 6) return 2;
    const uint32_t W = atoi(argv[2]), H = atoi(argv[3]);
    if (!strcmp(argv[1], "desc")) {
        auto [cam, world] = setup::setup(Extent2u(W, H), atoi(argv[4]) != 0);
        rayn_world_desc d = world.to_desc(cam);
        FILE* f = fopen(argv[5], "wb");
        fwrite(&d, sizeof d, 1, f);
        fclose(f);
        return 0;
    }
    if (argc < 8) return 2;
    auto [cam, world] = setup::setup(Extent2u(W, H), atoi(argv[6]) != 0);
    try {
        Film dup({ChannelKind::Color, ChannelKind::Color}, Extent2u(W, H));
        return 3; // must have thrown (Film::new returns Err for duplicate kinds)
    } catch (const std::invalid_argument&) {}
    const std::vector<ChannelKind> all = {ChannelKind::Color, ChannelKind::Alpha, ChannelKind::Background, ChannelKind::WorldNormal};
    const int n_dev = argc > 8 ? atoi(argv[8]) : 0;
    Film film_single(all, Extent2u(W, H)), film_multi(all, Extent2u(W, H), std::vector<int>((size_t)(n_dev > 0 ? n_dev : 1), 0));
    Film& film = n_dev > 0 ? film_multi : film_single;
    PathTracingIntegrator integ{(size_t)atoi(argv[5]), 2};
    const float t0 = 1.0f * (1.0f / 24.0f);
    film.render_frame_into(world, cam, integ, BlackmanHarrisFilter::new_(1.5f), Extent2u(16, 16), 1, {t0, t0 + 1.0f / 24.0f}, (size_t)atoi(argv[4]));
    { // per-device statistics of the one context (rayn_hip_get_entry_stats): every entry rendered tiles, the entries sum to the frame
        const rayn_stats tot = film.stats();
        uint64_t paths = 0, segments = 0, tiles = 0;
        for (int e = 0; e < film.device_count(); e++) {
            const rayn_stats s = film.device_stats(e);
            if (s.tiles == 0 || s.ms_total <= 0.0) return 4;
            paths += s.paths; segments += s.segments; tiles += s.tiles;
        }
        if (paths != tot.paths || segments != tot.segments || tiles != tot.tiles || film.device_count() != (n_dev > 0 ? n_dev : 1)) return 5;
    }
    FILE* f = fopen(argv[7], "wb");
    fwrite(film.color.data(), 4, film.color.size(), f);
    fwrite(film.alpha.data(), 4, film.alpha.size(), f);
    fwrite(film.background.data(), 4, film.background.size(), f);
    fwrite(film.world_normal.data(), 4, film.world_normal.size(), f);
    fclose(f);
    return 0;
}
