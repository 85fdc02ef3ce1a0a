// Library configuration and error reporting (host code only).
//
// Every kernel-selection switch lives in ONE explicit, caller-visible table (ss_config_set / ss_config_get) instead of
// function-local statics that latch an environment variable at first use.  The SS_* environment variables only provide the
// table's INITIAL values (read once, when the library is loaded), so existing command lines keep working; a host can change a
// key between calls (the parity tests run all three arithmetic modes in one process this way).  The table is process-wide and
// not synchronised: set it from one thread while no call is in flight.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct Key { const char* name; int SsTuning::* field; const char* help; };

const Key KEYS[] = {
    {"x6", &SsTuning::x6, "1: AUTO uses the 16-bit matrix-core contraction with fp32-grade operand splits; 0: fp32 MFMA instructions only"},
    {"x3h", &SsTuning::x3h, "1: two fp16 pieces, three products (default); 0: exact three-piece bf16 split, six products"},
    {"x3h_direct", &SsTuning::x3h_direct, "x3h also in the direct (non-Winograd) convolutions and weight gradients"},
    {"x6p", &SsTuning::x6p, "pre-split-plane LDS-DMA GEMM for the Winograd passes: 0 off, 1 for launches of >= 1024 workgroups, 2 always"},
    {"winograd", &SsTuning::winograd, "0: no Winograd path (direct implicit GEMMs everywhere)"},
    {"wino_r", &SsTuning::wino_r, "Winograd output tile edge: 4 = F(4x4,3x3) (default), 2 = F(2x2,3x3)"},
    {"wgrad_c1", &SsTuning::wgrad_c1, "MFMA weight-gradient kernel for the one-channel 7x7 stem / head"},
    {"norm_fused_pix", &SsTuning::norm_fused_pix, "largest group (pixels) normalised by the one-launch kernel; 0 disables"},
    {"gconv_fast", &SsTuning::gconv_fast, "0: generic gather loaders in gconv_mfma (measurement)"},
    {"gconv_nt512", &SsTuning::nt512, "measurement: 512-thread 128x128 tile variant"},
    {"gconv_tile256", &SsTuning::tile256, "measurement: 256x128 tile variant"},
    {"tile_conv", &SsTuning::tile_conv, "LDS-staged tile kernel for small-channel stride-1 convolutions (MultiResUNet full-resolution layers)"},
    {"weight_cache", &SsTuning::weight_cache, "reserved"},
};

bool env_is(const char* name, char c) { const char* v = getenv(name); return v && v[0] == c; }

SsTuning from_env() {
    SsTuning v;
    v.x6 = env_is("SS_X6", '0') ? 0 : 1;
    v.x3h = env_is("SS_X3H", '0') ? 0 : 1;
    v.x3h_direct = env_is("SS_X3H_DIRECT", '0') ? 0 : 1;
    v.x6p = env_is("SS_X6P", '0') ? 0 : (env_is("SS_X6P", 'f') ? 2 : 1);
    v.winograd = getenv("SS_NO_WINOGRAD") ? 0 : 1;
    v.wino_r = env_is("SS_WINO_R", '2') ? 2 : 4;
    v.wgrad_c1 = env_is("SS_WGRAD_C1", '0') ? 0 : 1;
    v.norm_fused_pix = getenv("SS_NORM_FUSED_PIX") ? atoi(getenv("SS_NORM_FUSED_PIX")) : 1024;
    v.gconv_fast = getenv("SS_GCONV_NOFAST") ? 0 : 1;
    v.nt512 = getenv("SS_GCONV_NT512") ? 1 : 0;
    v.tile256 = getenv("SS_GCONV_256") ? 1 : 0;
    v.tile_conv = env_is("SS_TILE_CONV", '0') ? 0 : 1;
    v.weight_cache = 1;
    return v;
}

SsTuning g_tuning = from_env();
thread_local char g_err[512] = "";

}  // namespace

const SsTuning& ss_tuning() { return g_tuning; }

void ss_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int ss_config_set(const char* key, int64_t value) {
    if (!key) return SS_ERR_INVALID;
    for (const Key& k : KEYS)
        if (!strcmp(k.name, key)) {
            g_tuning.*(k.field) = (int)value;
            return SS_OK;
        }
    ss_set_error("ss_config_set: unknown key '%s'", key);
    return SS_ERR_INVALID;
}

int64_t ss_config_get(const char* key) {
    if (key)
        for (const Key& k : KEYS)
            if (!strcmp(k.name, key)) return g_tuning.*(k.field);
    ss_set_error("ss_config_get: unknown key '%s'", key ? key : "(null)");
    return INT64_MIN;
}

const char* ss_config_key(int index) {
    return (index >= 0 && index < (int)(sizeof(KEYS) / sizeof(KEYS[0]))) ? KEYS[index].name : nullptr;
}

const char* ss_last_error(void) { return g_err; }

}  // extern "C"
