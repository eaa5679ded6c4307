// CPU test of csrc/tile_pool.hpp (the host threads of armon_hip_mgpu_cycle), built with -fsanitize=thread by
// tests/test_tile_pool.py: ordering across the barriers, failure propagation, reuse, many short runs, teardown.
#include "../../armon.jl_amd/csrc/tile_pool.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>

static thread_local std::string tl_error;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); exit(1); } \
    } while (0)

int main(int argc, char** argv)
{
    const int runs = argc > 1 ? atoi(argv[1]) : 2000;
    for (int n : {2, 3, 8}) {
        tile_pool pool(n, [] { return tl_error; });
        std::mt19937 rng(1234 + n);
        // done[s][k]: plain (non-atomic) ints written by tile k in step s and read by every tile in step s + 1 — the
        // barrier is the only synchronisation, so a missing happens-before is a data race ThreadSanitizer reports
        for (int r = 0; r < runs; r++) {
            const int n_steps = 1 + (int)(rng() % 9);
            std::vector<std::vector<int>> done(n_steps, std::vector<int>(n, 0));
            const int fail_step = (r % 7 == 3) ? (int)(rng() % n_steps) : -1, fail_tile = (int)(rng() % n);
            std::vector<tile_pool::step_fn> steps;
            for (int s = 0; s < n_steps; s++)
                steps.push_back([&, s](size_t k) -> int {
                    if (s > 0)
                        for (int q = 0; q < n; q++)
                            if (done[s - 1][q] != 1) { tl_error = "a step started before the previous one was complete"; return 99; }
                    if ((rng.max() & (k + s + r)) % 5 == 0) std::this_thread::yield();       // uneven arrival at the barrier
                    if (s == fail_step && (int)k == fail_tile) { tl_error = "injected failure"; return 7; }
                    done[s][k] = 1;
                    return 0;
                });
            int tile = -1;
            std::string message;
            const int rc = pool.run(steps, &tile, &message);
            if (fail_step < 0) {
                CHECK(rc == 0);
                for (int s = 0; s < n_steps; s++)
                    for (int k = 0; k < n; k++) CHECK(done[s][k] == 1);
            } else {
                CHECK(rc == 7 && tile == fail_tile && message == "injected failure");
                for (int s = fail_step + 1; s < n_steps; s++)          // nothing after the failing step ran, on any tile
                    for (int k = 0; k < n; k++) CHECK(done[s][k] == 0);
                for (int s = 0; s < fail_step; s++)
                    for (int k = 0; k < n; k++) CHECK(done[s][k] == 1);
            }
        }
    }
    {   // created and destroyed without ever running; and an empty list of steps
        tile_pool idle(4, nullptr);
        tile_pool empty(3, nullptr);
        std::vector<tile_pool::step_fn> none;
        CHECK(empty.run(none, nullptr, nullptr) == 0);
    }
    printf("tile_pool OK\n");
    return 0;
}
