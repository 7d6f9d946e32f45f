// prof.hip — optional per-kernel-family timing with HIP events on the launch stream, so that
// bench.py can report the dominant kernel's achieved rate measured live (roofline.achieved).
#include "common.hpp"
#include "../../include/unet_hip.h"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace unet {

struct ProfRec { hipEvent_t a, b; int family; double flops; char tag[96]; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void prof_begin(int family, double flops, hipStream_t st, const char *tag)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRec r{get_event(), get_event(), family, flops, {0}};
    if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); }
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
}

void prof_end(hipStream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
}

}  // namespace unet

using namespace unet;

extern "C" {

int unet_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}

int unet_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    return 0;
}

int unet_profile_read(int family, double *ms_total, long *launches, double *flops_total)
{
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0.0, fl = 0.0;
    long n = 0;
    for (auto &r : g_recs) {
        if (r.family != family) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.flops; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = n;
    if (flops_total) *flops_total = fl;
    return 0;
}

/* writes one line per recorded launch: family,ms,gflop,tag */
int unet_profile_dump(const char *path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    FILE *f = fopen(path, "w");
    if (!f) { set_error("unet_profile_dump: cannot open %s", path); return -2; }
    fprintf(f, "family,ms,gflop,tag\n");
    for (auto &r : g_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        fprintf(f, "%d,%.6f,%.6f,%s\n", r.family, t, r.flops / 1e9, r.tag);
    }
    fclose(f);
    return 0;
}

}  // extern "C"
