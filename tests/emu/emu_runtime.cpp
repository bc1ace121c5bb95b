// TEST INFRASTRUCTURE ONLY (see tests/emu/hip/hip_runtime.h): the work-group runner and the few miblast_* entry points the
// chaining-stage sources and bin/paffy's main() need, so that `emu_paffy` runs mp_chain.cpp + mp_kernels.hip on the host.
#include "mb_pipeline.h"

#include <string>


namespace mb {
void host_parallel_for(size_t n, const std::function<void(size_t)> &f) { for (size_t i = 0; i < n; i++) f(i); }
HostHot::HostHot() {}
HostHot::~HostHot() {}
int host_threads() { return 4; }                              // several chunks, run one after another: the merge paths are exercised
unsigned hist[8192];                                          // the dynamic LDS of k_tile
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
const std::string &last_error_text() { return g_err; }
}  // namespace mb

extern "C" {
int miblast_device_count(void) { return 1; }
int miblast_frontend_runtime_defaults(int) { return 0; }      // (the front end asks for polling waits: nothing to poll here)
int miblast_ctx_create(int, miblast_ctx **ctx) { *ctx = new miblast_ctx(); return MIBLAST_OK; }
void miblast_ctx_destroy(miblast_ctx *ctx) { if (ctx) mb::chain_cache_destroy(ctx->c.chain_cache); delete ctx; }
const char *miblast_last_error(void) { return mb::g_err.c_str(); }
void miblast_free(void *p) { free(p); }
}
