// rayn_main.cpp — the reference's driver (src/main.rs:28-98) on top of the C++ host mirror: build the
// shipped scene, render one frame, print "Done in {} seconds.", write the colour image (binary PPM,
// Film::save_to's post-process: Color+Background, saturate, gamma 2.2, rows flipped; src/film.rs:247-263).
//   rayn_main [width height SAMPLES bounces volumes(0|1) out.ppm]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../include/rayn_host.hpp"

int main(int argc, char** argv) {
    using namespace rayn;
    const uint32_t W = argc > 1 ? atoi(argv[1]) : 1280, H = argc > 2 ? atoi(argv[2]) : 720;
    const size_t SAMPLES = argc > 3 ? atoi(argv[3]) : 2, BOUNCES = argc > 4 ? atoi(argv[4]) : 3;
    const bool volumes = argc > 5 ? atoi(argv[5]) != 0 : true;
    const char* out = argc > 6 ? argv[6] : "render_color.ppm";
    auto [camera, world] = setup::setup(Extent2u(W, H), volumes);
    Film film({ChannelKind::Color, ChannelKind::Alpha, ChannelKind::Background, ChannelKind::WorldNormal}, Extent2u(W, H));
    const int frame_rate = 24;
    const float shutter_speed = 1.0f / 24.0f;
    BlackmanHarrisFilter filter = BlackmanHarrisFilter::new_(1.5f);
    PathTracingIntegrator integrator{BOUNCES, 2};
    for (size_t frame = 1; frame < 2; frame++) {
        auto start = std::chrono::steady_clock::now();
        float frame_start = (float)frame * (1.0f / (float)frame_rate);
        float frame_end = frame_start + shutter_speed;
        film.render_frame_into(world, camera, integrator, filter, Extent2u(16, 16), frame, {frame_start, frame_end}, SAMPLES);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        printf("Done in %g seconds.\n", secs);
        rayn_stats st = film.stats();
        printf("%llu paths, %llu segments, %.1f Mpath-samples/s (device %.1f ms)\n", (unsigned long long)st.paths,
               (unsigned long long)st.segments, st.paths / secs / 1e6, st.ms_total);
        FILE* f = fopen(out, "wb");
        if (!f) return 1;
        fprintf(f, "P6\n%u %u\n255\n", W, H);
        for (uint32_t y = 0; y < H; y++)
            for (uint32_t x = 0; x < W; x++) {
                size_t i = x + (size_t)(H - 1 - y) * W;
                for (int c = 0; c < 3; c++) {
                    float v = film.color[3 * i + c] + film.background[3 * i + c];
                    v = std::pow(std::fmin(std::fmax(v, 0.0f), 1.0f), 1.0f / 2.2f);
                    fputc((int)std::fmin(std::fmax(v * 255.0f, 0.0f), 255.0f), f);
                }
            }
        fclose(f);
        printf("Saving to %s...\n", out);
    }
    return 0;
}
