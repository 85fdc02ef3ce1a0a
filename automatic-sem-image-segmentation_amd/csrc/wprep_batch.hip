// Batched weight preparation: record the weight-derived-operand launches of a network once, replay them as one launch per job type.
//
// Where it comes from: at per-GPU batch 1 a train step is ~2 900 dispatches of which ~430 prepare operands that depend on the weights
// alone (weight maxima and their memsets, tap-wise transposes for the data gradient, x3h split planes of the gather convolutions,
// Winograd-transformed planes of the trunk) -- layer by layer, 4 - 14 us each, the same kernels with the same pointers every step
// (the weights live in one arena, the operands in caller-owned per-layer caches: ss_wcache).  A dependent dispatch costs ~10 us
// whoever issues it (profiles/r05_unet_hipgraph_probe.txt), so the refresh is recorded ONCE (ss_wprep_record_begin / _end: the hooked
// launch sites push SsWJob records instead of launching) and replayed per step by ss_wprep_run: zero the maxima, then one launch per
// job type with a workgroup -> job map.  Same kernels' bodies, same operands, bit for bit.
#include "common.h"

#include <string.h>

#include <vector>

namespace {

struct Region { char* p; size_t n; bool covered; };
struct Recorder {
    bool on = false, incomplete = false;
    std::vector<SsWJob> jobs;
    std::vector<Region> regions;
};
thread_local Recorder g_rec;

constexpr uint32_t PLAN_MAGIC = 0x53575031u;      // "SWP1"
struct PlanHeader {
    uint32_t magic;
    int32_t n_jobs, complete, n_amax;
    int32_t first[SS_WJ_TYPES], count[SS_WJ_TYPES], blocks[SS_WJ_TYPES], map_off[SS_WJ_TYPES];      // map_off: in ints from maps_off
    uint64_t jobs_off, maps_off, total_bytes;
};

inline long job_blocks(const SsWJob& j) { return (long)j.gx * j.gy * j.gz; }

size_t plan_bytes(const std::vector<SsWJob>& jobs) {
    long blocks = 0;
    for (const SsWJob& j : jobs) blocks += job_blocks(j);
    return ss_align_up(sizeof(PlanHeader), 256) + ss_align_up(jobs.size() * sizeof(SsWJob), 256) + ss_align_up((size_t)blocks * sizeof(int32_t), 256);
}

// maxima are raised with atomics: the words start from zero (what the hipMemsetAsync in front of every single-job launch did)
__global__ __launch_bounds__(256) void wbatch_zero_kernel(const SsWJob* __restrict__ jobs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *(unsigned int*)jobs[i].dst = 0u;
}

}  // namespace

bool ss_wrec_on() { return g_rec.on; }

void ss_wrec_note_region(void* ptr, size_t bytes) {
    if (g_rec.on) g_rec.regions.push_back(Region{(char*)ptr, bytes, false});
}

void ss_wrec_unbatched() {
    if (g_rec.on) g_rec.incomplete = true;
}

void ss_wrec_push(const SsWJob& j) {
    if (!g_rec.on) return;
    // a job whose destination is not a cache entry allocated during this recording writes into the caller's workspace: nothing a later
    // step could use -- dropped (the per-layer refresh computes and discards it as well)
    bool keep = false;
    for (Region& r : g_rec.regions) {
        const char* d1 = (const char*)j.dst;
        const char* d2 = (const char*)j.dst2;
        if (d1 >= r.p && d1 < r.p + r.n) { r.covered = true; keep = true; }
        if (d2 && d2 >= r.p && d2 < r.p + r.n) r.covered = true;
    }
    if (keep) g_rec.jobs.push_back(j);
}

extern "C" {

int ss_wprep_record_begin(void) {
    if (g_rec.on) { ss_set_error("ss_wprep_record_begin: a recording is already active on this thread"); return SS_ERR_INVALID; }
    g_rec = Recorder();
    g_rec.on = true;
    return SS_OK;
}

int ss_wprep_record_end(size_t* needed_bytes, int32_t* n_jobs, int32_t* complete) {
    if (!g_rec.on) { ss_set_error("ss_wprep_record_end: no recording is active on this thread"); return SS_ERR_INVALID; }
    g_rec.on = false;
    bool all = !g_rec.incomplete;
    for (const Region& r : g_rec.regions) all = all && r.covered;          // an entry nobody recorded a fill for: a launch site without a hook
    g_rec.incomplete = !all;
    if (needed_bytes) *needed_bytes = plan_bytes(g_rec.jobs);
    if (n_jobs) *n_jobs = (int32_t)g_rec.jobs.size();
    if (complete) *complete = all ? 1 : 0;
    return SS_OK;
}

int ss_wprep_plan_write(void* plan_host, size_t bytes) {
    if (g_rec.on) { ss_set_error("ss_wprep_plan_write: the recording is still active (ss_wprep_record_end first)"); return SS_ERR_INVALID; }
    const size_t need = plan_bytes(g_rec.jobs);
    if (!plan_host || bytes < need) return SS_ERR_WORKSPACE;
    memset(plan_host, 0, need);
    PlanHeader* h = (PlanHeader*)plan_host;
    h->magic = PLAN_MAGIC;
    h->n_jobs = (int32_t)g_rec.jobs.size();
    h->complete = g_rec.incomplete ? 0 : 1;
    h->jobs_off = ss_align_up(sizeof(PlanHeader), 256);
    h->maps_off = h->jobs_off + ss_align_up(g_rec.jobs.size() * sizeof(SsWJob), 256);
    h->total_bytes = need;
    SsWJob* out = (SsWJob*)((char*)plan_host + h->jobs_off);
    int32_t* map = (int32_t*)((char*)plan_host + h->maps_off);
    int ji = 0;
    long mo = 0;
    for (int t = 0; t < SS_WJ_TYPES; ++t) {          // jobs sorted by type (recording order inside a type), one block map per type
        h->first[t] = ji;
        h->map_off[t] = (int32_t)mo;
        long blk = 0;
        for (const SsWJob& j : g_rec.jobs) {
            if (j.type != t) continue;
            SsWJob q = j;
            q.blk0 = (int32_t)blk;
            const long nb = job_blocks(j);
            for (long b = 0; b < nb; ++b) map[mo + blk + b] = ji;
            blk += nb;
            out[ji++] = q;
        }
        h->count[t] = ji - h->first[t];
        h->blocks[t] = (int32_t)blk;
        mo += blk;
    }
    h->n_amax = h->count[SS_WJ_AMAX];
    g_rec = Recorder();
    return SS_OK;
}

int ss_wprep_run(const void* plan_host, const void* plan_dev, size_t bytes, void* stream) {
    const int rc = ss_wprep_run_part(plan_host, plan_dev, bytes, 0, stream);
    return rc != SS_OK ? rc : ss_wprep_run_part(plan_host, plan_dev, bytes, 1, stream);
}

int ss_wprep_run_part(const void* plan_host, const void* plan_dev, size_t bytes, int part, void* stream) {
    const PlanHeader* h = (const PlanHeader*)plan_host;
    if (!h || !plan_dev || bytes < sizeof(PlanHeader) || h->magic != PLAN_MAGIC || h->total_bytes > bytes) {
        ss_set_error("ss_wprep_run: not a plan written by ss_wprep_plan_write (or a truncated copy)");
        return SS_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    const SsWJob* jobs = (const SsWJob*)((const char*)plan_dev + h->jobs_off);
    const int32_t* maps = (const int32_t*)((const char*)plan_dev + h->maps_off);
    if (part == 1) return h->blocks[SS_WJ_WINO_H] > 0 ? ss_wbatch_launch_wino(jobs, maps + h->map_off[SS_WJ_WINO_H], h->blocks[SS_WJ_WINO_H], s) : SS_OK;
    if (part != 0) return SS_ERR_INVALID;
    if (h->n_amax > 0) {
        hipLaunchKernelGGL(wbatch_zero_kernel, dim3((h->n_amax + 255) / 256), dim3(256), 0, s, jobs + h->first[SS_WJ_AMAX], h->n_amax);
        SS_LAUNCH_CHECK();
    }
    // maxima and transposes first: the split / transformed planes read them
    int rc = SS_OK;
    if (h->blocks[SS_WJ_AMAX] > 0) rc = ss_wbatch_launch_amax(jobs, maps + h->map_off[SS_WJ_AMAX], h->blocks[SS_WJ_AMAX], s);
    if (rc == SS_OK && h->blocks[SS_WJ_TRANSPOSE] > 0) rc = ss_wbatch_launch_transpose(jobs, maps + h->map_off[SS_WJ_TRANSPOSE], h->blocks[SS_WJ_TRANSPOSE], s);
    if (rc == SS_OK && h->blocks[SS_WJ_WPREP_H] > 0) rc = ss_wbatch_launch_wprep(true, jobs, maps + h->map_off[SS_WJ_WPREP_H], h->blocks[SS_WJ_WPREP_H], s);
    if (rc == SS_OK && h->blocks[SS_WJ_WPREP_3] > 0) rc = ss_wbatch_launch_wprep(false, jobs, maps + h->map_off[SS_WJ_WPREP_3], h->blocks[SS_WJ_WPREP_3], s);
    return rc;
}

}  // extern "C"
