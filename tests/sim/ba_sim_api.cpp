// tests/sim/ba_sim_api.cpp -- TEST AID: the few non-BA pieces of the C-ABI that the BA sources need (ctx life cycle, error
// text, debug knobs), so that csrc/mvo_api_ba.cpp + csrc/ba_host.cpp + csrc/ba_kernels.hip -- the product sources,
// unmodified -- can be built against the CPU emulation of the HIP runtime (tests/sim/hip_emu) into
// tests/sim/_build/libmvo_ba_sim.so.  The CPU suite runs the BA kernel SOURCE thread for thread through it and compares
// with the oracle bit for bit; the product library has no CPU path and never links any of this.
#include <cstdio>
#include <cstring>

#include "mvo_internal.h"

int mvo_set_err(mvo_ctx* c, int code, const char* what, hipError_t e) {
    if (c) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s (%d)", what, (int)e);
        c->err = buf;
    }
    return code;
}
void mvo_prof_begin(mvo_ctx*, const char*) {}
void mvo_prof_end(mvo_ctx*) {}
void mvo_prof_collect(mvo_ctx*) {}
int ba_debug_set(const char* key, int value);

extern "C" {

int mvo_create(mvo_ctx** out, int device) {
    mvo_ctx* ctx = new mvo_ctx();
    ctx->device = device;
    (void)hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    *out = ctx;
    return MVO_OK;
}
void mvo_destroy(mvo_ctx* ctx) {
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    ba_pool_release(ctx);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
const char* mvo_last_error(const mvo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int mvo_debug_set(const char* key, int value) {
    if (key && !std::strncmp(key, "ba_", 3)) return ba_debug_set(key, value);
    return MVO_ERR_INVALID;
}
}
