// render.cpp -- the reference's main.rs (main.rs:45-67) as a plain C++ program over the C ABI: no HIP
// headers, no Python.  Build:  g++ -O2 examples/render.cpp -Iinclude -Lrobigo_luculenta_amd
//                                  -lrobigo_luculenta -Wl,-rpath,$PWD/robigo_luculenta_amd -o examples/render
// Usage:  examples/render [batches=64] [width=1280] [height=720] [output=output.png]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "robigo_luculenta.h"

int main(int argc, char** argv) {
    RlAppConfig cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.max_batches = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 64;
    cfg.width = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1280;  // main.rs:47
    cfg.height = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 720;  // main.rs:48
    cfg.output_ppm = argc > 4 ? argv[4] : "output.png";          // main.rs:61
    cfg.device = 0;
    cfg.concurrency = 2;
    cfg.seed = 1;
    cfg.builtin_scene = RL_SCENE_DEMO;
    cfg.tonemap_interval_ms = 30000;                             // task_scheduler.rs:44-46
    cfg.fused = 1;
    cfg.verbose = 0;
    if (rl_device_count() < 1) {
        std::fprintf(stderr, "no GPU visible: this renderer has no CPU path\n");
        return 2;
    }
    std::printf("rendering %llu batches at %ux%u on %s\n", (unsigned long long)cfg.max_batches, cfg.width, cfg.height, rl_version());
    RlAppStats st;
    const int rc = rl_app_run(&cfg, &st, nullptr);
    if (rc != RL_OK) {
        std::fprintf(stderr, "rl_app_run failed (%d): %s\n", rc, rl_last_error());
        return 1;
    }
    std::printf("%llu batches, %.1f Mrays in %.2f s (%.1f Mrays/s, %.1f batches/sec); wrote image to %s\n",
                (unsigned long long)st.batches, st.segments / 1e6, st.seconds, st.segments / st.seconds / 1e6,
                st.batches / st.seconds, cfg.output_ppm);
    return 0;
}
