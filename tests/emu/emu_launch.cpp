// TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h): the work-group runner of the host-side emulation -- one pthread per
// work-item, workgroups one after another (the threads of a launch are reused from workgroup to workgroup).
#include <hip/hip_runtime.h>

#include <algorithm>

namespace emu {
thread_local Group *g_group = nullptr;
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

void launch(const std::function<void()> &body, dim3 grid, dim3 block) {
    const unsigned nt = block.x;
    if (grid.x == 0 || nt == 0) return;
    Group g;
    pthread_barrier_init(&g.all, nullptr, nt);
    const unsigned nw = (nt + 63) / 64;
    g.wave.resize(nw);
    for (unsigned w = 0; w < nw; w++) pthread_barrier_init(&g.wave[w], nullptr, std::min(64u, nt - 64 * w));
    g.slot.assign(nt, 0);
    pthread_barrier_t next;                                  // between two workgroups (a work-item that left its kernel early waits here)
    pthread_barrier_init(&next, nullptr, nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            g_group = &g;
            t_threadIdx = dim3(t); t_blockDim = block; t_gridDim = grid;
            for (unsigned b = 0; b < grid.x; b++) {
                t_blockIdx = dim3(b);
                body();
                pthread_barrier_wait(&next);
            }
        });
    for (std::thread &x : th) x.join();
    pthread_barrier_destroy(&next);
    pthread_barrier_destroy(&g.all);
    for (unsigned w = 0; w < nw; w++) pthread_barrier_destroy(&g.wave[w]);
}
}  // namespace emu
