// Library configuration and error reporting (host code only).
//
// Every kernel-selection switch lives in ONE explicit, caller-visible table (ss_config_set / ss_config_get) instead of
// function-local statics that latch an environment variable at first use.  The SS_* environment variables only provide the
// table's INITIAL values (read once, when the library is loaded), so existing command lines keep working; a host can change a
// key between calls (the parity tests run all three arithmetic modes in one process this way).  The table is process-wide and
// not synchronised: set it from one thread while no call is in flight.
#include "common.h"

#include <mutex>
#include <string>
#include <vector>

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
    {"tile_th", &SsTuning::tile_th, "tile kernel rows per tile: 0 auto, 4 or 8 (measurement)"},
    {"tile_dbg", &SsTuning::tile_dbg, "measurement: phase-skipping bit mask of the tile kernel"},
    {"tile_stagger", &SsTuning::tile_stagger, "tile kernel: start delay between co-resident workgroups (units of 2048 cycles)"},
    {"weight_cache", &SsTuning::weight_cache, "reserved"},
    {"gemm_persistent", &SsTuning::gemm_persistent, "pre-split-plane GEMMs: one persistent workgroup per CU whose operand stream runs across tile boundaries; 0: one workgroup per tile"},
    {"x6p_wide", &SsTuning::x6p_wide, "OPT-IN (default 0): Winograd GEMMs with 256-multiple output channels on 256 x 256 tiles; their planes carry the plain low piece (values below 2^-17 of their tile's maximum keep absolute, not relative, precision)"},
    {"gconv_v2", &SsTuning::gconv_v2, "gather convolutions: two-stage one-barrier kernel with LDS-DMA weight planes (conv_mfma_x6v2.hip) where it applies; 0: gconv_x6_kernel only"},
    {"twgrad_x3h", &SsTuning::twgrad_x3h, "tile weight gradient (MultiResUNet full-resolution layers) on the fp16 matrix cores: two fp16 pieces per operand under per-tile scales, transposing LDS reads; 0: fp32 MFMA (v_mfma_f32_32x32x2_f32)"},
    {"wino16_products", &SsTuning::wino16_products, "Winograd forward / data gradient on 16-bit stored activations: 1 = one fp16 plane per operand, one product (fast; the F(4x4,3x3) transforms amplify the 2^-12 operand rounding to ~3e-3 per layer), 3 = two planes, three products (fp32-grade arithmetic, only the storage is 16-bit)"},
    {"c1_mfma", &SsTuning::c1_mfma, "one-channel stem / head layers (1 -> C, C -> 1, full resolution) on the fp16 matrix cores with the x3h arithmetic; 0: LDS-tiled VALU kernels"},
    {"x6p_pp", &SsTuning::x6p_pp, "x3h Winograd GEMMs: ping-pong schedule (the two waves of a SIMD half a K step apart: one multiplies while the other loads); 0: one-phase kernel (same results bit for bit)"},
    {"wino_save", &SsTuning::wino_save, "Winograd x3h forward keeps its transformed input planes for the weight gradient when the caller provides ss_conv_desc::saved_operand (ss_conv2d_saved_operand_bytes > 0); 0: the weight gradient transforms x again"},
    {"gemm_ilv", &SsTuning::gemm_ilv, "gemm_x6p / gconv_x6v2: fragment reads issued between the MFMAs of a half step instead of in a burst in front of them (bit-identical); 0: burst form.  (gemm_tn_x3h keeps the burst form: measured slower interleaved)"},
    {"gemm_cus", &SsTuning::gemm_cus, "persistent pre-split-plane GEMMs: workgroups (= CUs occupied; a multiple of 8) per launch, 0 = all CUs.  Fewer leave whole CUs to the kernels of the other HIP stream (a GEMM workgroup takes a CU's whole LDS: nothing else starts beside it)"},
    {"gconv_phases", &SsTuning::gconv_phases, "OPT-IN (default 0; measured slower or equal: the phases' weight planes then compete for one L2): stride-2 data gradients / transposed convolutions with the four sub-pixel phases in ONE launch of the x3h gather kernels (same results bit for bit); 0: one launch per phase"},
    {"phases_fused", &SsTuning::phases_fused, "stride-2 data gradients / transposed convolutions (fp32 storage, x3h): the four sub-pixel phases of an input tile in ONE workgroup -- the tile is loaded and split once per 32-channel chunk and every tap reads it from LDS (conv_phase.hip); 0: one gather-kernel launch per phase"},
    {"phases_split", &SsTuning::phases_split, "the fused sub-pixel kernel with TWO phases per wave pair (64 pixels x 64 channels x 2 phases per wave: one A and one B fragment read per MFMA triple instead of 1.5; the weight stream interleaves the two wave pairs' taps); 0: all four phases in every wave"},
    {"norm_order", &SsTuning::norm_order, "bit mask, order in which the norm kernels' workgroups walk a tensor (results bit-identical): 1 = backward statistics from the END (last group / chunk first: what the producer of dy wrote last is still in the Infinity Cache), 2 = apply kernels take the GROUPS from the end as well (they always take the row chunks of a group from the end), 4 = backward apply walks forward (pairs with 1)"},
    {"norm_fuse_fin", &SsTuning::norm_fuse_fin, "norm passes over mid-size tensors reduce the statistics partials of their own channel block in the APPLY kernel's prologue (fixed order, fp64) instead of a separate finalize launch; bit 0: forward passes (OPT-IN: as accurate per op as the finalize kernels -- tools/norm_fuse_diag.py -- but another rounding of the statistics, and the reference-generated CycleGAN vectors contain constant tiles whose ReLU masks hang on that rounding: profiles/r06_norm_fused_finalize.md), bit 1: backward passes (the default: 2); 0: always the finalize kernels"},
    {"wgrad_mfma_x6", &SsTuning::wgrad_mfma_x6, "generic weight-gradient kernel (channel counts that are not multiples of 32, unaligned views: the MultiResUNet's odd widths) on the bf16 matrix cores with the exact three-piece split formed in registers (six products); 0: v_mfma_f32_32x32x2_f32"},
    {"wgrad_stage", &SsTuning::wgrad_stage, "weight gradient of the stride-2 3 x 3 / 4 x 4 layers with the operands staged once per spatial tile and every tap served from LDS (conv_wgrad_stage.hip); 0: wgrad_x6_kernel; 2 (measurement): the same kernel with its phases in lockstep"},
    {"norm_bwd_resident", &SsTuning::norm_bwd_resident, "OPT-IN (default 0; built, correct, measured 2.2x SLOWER than the two passes: profiles/r06_experiments.md section 9): InstanceNorm backward of fp32 tensors with <= 16384 pixels per sample and 32-multiple channels in ONE pass -- the workgroups of a (sample, 32-channel block) keep dy and x in registers across a group-local barrier (norm.hip)"},
    {"gemm_tn_rounds", &SsTuning::gemm_tn_rounds, "Winograd weight-gradient GEMM: the K range is split so that the (tile, split) units fill whole rounds of the CUs the launch occupies; 0 (measurement): the earlier rule (units for ~4 rounds); n > 1 (measurement): n splits"},
    {"gconv16_ragged", &SsTuning::gconv16_ragged, "OPT-IN (default 0): 16-bit activation storage, gconv_x6_kernel takes odd channel counts / strides in the stored type (ragged loader: config 2 889 -> 911 tiles/s); bits: 1 stride-1 problems, 2 strided-output problems, 4 one-tap problems (7 = all).  Off by default: with it on, the MultiResUNet's fp16 training on the publication's data ended in NaN on 2 of 8 generated data sets (never with bf16, never with it off; not reproduced on synthetic data) -- profiles/r06_experiments.md section 11"},
    {"x6p_wide1", &SsTuning::x6p_wide1, "16-bit activation storage: the one-plane Winograd GEMMs with 256-multiple output channels on 256 x 256 tiles (same bits as the 256 x 128 kernel); 0: off"},
    {"wino16_m16", &SsTuning::wino16_m16, "16-bit activation storage, one-plane Winograd layers: the GEMM writes its Winograd-domain product as fp16 (under a fixed power-of-two scale) and the output transform reads that -- half the bytes of the product's round trip; 0: fp32 product"},
    {"wgrad_tn", &SsTuning::wgrad_tn, "Winograd weight gradient on pre-split K-major fp16 planes with transposing LDS reads (gemm_tn_x3h.hip); 0: in-kernel split"},
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
    v.wgrad_tn = env_is("SS_WGRAD_TN", '0') ? 0 : 1;
    v.c1_mfma = env_is("SS_C1_MFMA", '0') ? 0 : 1;
    v.twgrad_x3h = env_is("SS_TWGRAD_X3H", '0') ? 0 : 1;
    v.wino16_products = env_is("SS_WINO16_PRODUCTS", '3') ? 3 : 1;
    v.gconv_v2 = env_is("SS_GCONV_V2", '0') ? 0 : 1;
    v.x6p_wide = env_is("SS_X6P_WIDE", '1') ? 1 : 0;
    v.gemm_persistent = env_is("SS_GEMM_PERSISTENT", '0') ? 0 : 1;
    v.x6p_pp = env_is("SS_X6P_PP", '1') ? 1 : 0;
    v.wino_save = env_is("SS_WINO_SAVE", '0') ? 0 : 1;
    v.gemm_ilv = env_is("SS_GEMM_ILV", '0') ? 0 : 1;
    v.gconv_phases = env_is("SS_GCONV_PHASES", '1') ? 1 : 0;
    v.phases_fused = getenv("SS_PHASES_FUSED") ? atoi(getenv("SS_PHASES_FUSED")) : 1;
    v.phases_split = getenv("SS_PHASES_SPLIT") ? atoi(getenv("SS_PHASES_SPLIT")) : 1;
    v.gemm_cus = getenv("SS_GEMM_CUS") ? atoi(getenv("SS_GEMM_CUS")) : 0;
    v.norm_fused_pix = getenv("SS_NORM_FUSED_PIX") ? atoi(getenv("SS_NORM_FUSED_PIX")) : 1024;
    v.gconv_fast = getenv("SS_GCONV_NOFAST") ? 0 : 1;
    v.nt512 = getenv("SS_GCONV_NT512") ? 1 : 0;
    v.tile256 = getenv("SS_GCONV_256") ? 1 : 0;
    v.tile_conv = env_is("SS_TILE_CONV", '0') ? 0 : 1;
    v.tile_th = getenv("SS_TILE_TH") ? atoi(getenv("SS_TILE_TH")) : 0;
    v.tile_dbg = getenv("SS_TILE_DBG") ? atoi(getenv("SS_TILE_DBG")) : 0;
    v.tile_stagger = getenv("SS_TILE_STAGGER") ? atoi(getenv("SS_TILE_STAGGER")) : 2;
    v.norm_order = getenv("SS_NORM_ORDER") ? atoi(getenv("SS_NORM_ORDER")) : 0;
    v.norm_fuse_fin = getenv("SS_NORM_FUSE_FIN") ? atoi(getenv("SS_NORM_FUSE_FIN")) : 2;
    v.wgrad_mfma_x6 = env_is("SS_WGRAD_MFMA_X6", '0') ? 0 : 1;
    v.wgrad_stage = env_is("SS_WGRAD_STAGE", '0') ? 0 : 1;
    v.x6p_wide1 = env_is("SS_X6P_WIDE1", '0') ? 0 : 1;
    v.gconv16_ragged = getenv("SS_GCONV16_RAGGED") ? atoi(getenv("SS_GCONV16_RAGGED")) : 0;
    v.gemm_tn_rounds = env_is("SS_GEMM_TN_ROUNDS", '0') ? 0 : 1;
    v.norm_bwd_resident = env_is("SS_NORM_BWD_RESIDENT", '1') ? 1 : 0;
    v.wino16_m16 = env_is("SS_WINO16_M16", '0') ? 0 : (env_is("SS_WINO16_M16", '2') ? 2 : 1);
    v.weight_cache = 1;
    return v;
}

SsTuning g_tuning = from_env();
thread_local char g_err[512] = "";

}  // namespace

const SsTuning& ss_tuning() { return g_tuning; }

// ---- in-process kernel timing (bench.py roofline leg): HIP events recorded on the LAUNCH stream around instrumented kernels ----
namespace {
struct ProfSlot { std::string name; long launches = 0; double flops = 0, bytes = 0, ms = 0; };
struct ProfRec { int slot; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfSlot> g_prof_slots;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_drain() {       // fold finished records into the slots (synchronises on each end event)
    for (ProfRec& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) g_prof_slots[r.slot].ms += ms;
        g_prof_pool.push_back(r.e0);
        g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
}
}  // namespace

SsProfScope::SsProfScope(const char* name, double flops, double bytes, hipStream_t s) : rec(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int slot = -1;
    for (size_t i = 0; i < g_prof_slots.size(); ++i)
        if (g_prof_slots[i].name == name) { slot = (int)i; break; }
    if (slot < 0) { g_prof_slots.push_back(ProfSlot()); slot = (int)g_prof_slots.size() - 1; g_prof_slots[slot].name = name; }
    g_prof_slots[slot].launches += 1;
    g_prof_slots[slot].flops += flops;
    g_prof_slots[slot].bytes += bytes;
    ProfRec r{slot, prof_event(), prof_event()};
    (void)hipEventRecord(r.e0, s);
    g_prof_recs.push_back(r);
    rec = (int)g_prof_recs.size() - 1;
}
SsProfScope::~SsProfScope() {
    if (rec < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (rec < (int)g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[rec].e1, stream);
}

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

int ss_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return SS_OK;
}
int ss_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    g_prof_slots.clear();
    return SS_OK;
}
int ss_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_prof_slots.size();
}
int ss_prof_get(int index, ss_prof_entry* out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!out || index < 0 || index >= (int)g_prof_slots.size()) return SS_ERR_INVALID;
    prof_drain();
    const ProfSlot& p = g_prof_slots[index];
    snprintf(out->name, sizeof(out->name), "%s", p.name.c_str());
    out->launches = p.launches;
    out->total_ms = p.ms;
    out->flops = p.flops;
    out->bytes = p.bytes;
    return SS_OK;
}

}  // extern "C"
