// prof.hip — optional per-launch timing with HIP events on the launch stream, so that bench.py can report every
// kernel's achieved rate measured live inside its timed region (roofline block and the per-layer table).
// Every launch site brackets its kernel with prof_begin / prof_end and states the launch's algorithmic FLOPs
// (2*MAC of the op, in-bounds taps only), the FLOPs the matrix cores actually execute for it (Winograd: 16/36 of the
// direct count; tile padding included) and its algorithmic HBM bytes (inputs once + outputs once + weights once).
#include "common.hpp"
#include "../../include/unet_hip.h"

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace unet {

struct ProfRec { hipEvent_t a, b; int kind; double flops, exec_flops, bytes; char row[40]; char tag[96]; };
static std::mutex g_mu;
static bool g_on = false;
static unsigned g_kinds = 0xFFFFFFFFu;     // launch kinds (PK_*) that get events while profiling is on (unet_profile_select)
static bool g_open = false;                 // the last prof_begin pushed a record that still waits for its end event
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static thread_local char g_row[40] = "";

static hipEvent_t get_event()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void prof_scope(const char *row)
{
    if (!row) { g_row[0] = 0; return; }
    strncpy(g_row, row, sizeof(g_row) - 1);
    g_row[sizeof(g_row) - 1] = 0;
}

void prof_begin(int kind, const char *tag, hipStream_t st, double alg_flops, double exec_flops, double alg_bytes)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_open = false;
    if (!((g_kinds >> kind) & 1u)) return;
    ProfRec r{get_event(), get_event(), kind, alg_flops, exec_flops, alg_bytes, {0}, {0}};
    if (!r.a || !r.b) {                      // no record: give back what was obtained, prof_end then does nothing
        if (r.a) g_pool.push_back(r.a);
        if (r.b) g_pool.push_back(r.b);
        return;
    }
    memcpy(r.row, g_row, sizeof(r.row));
    if (tag) strncpy(r.tag, tag, sizeof(r.tag) - 1);
    if (hipEventRecord(r.a, st) != hipSuccess) { g_pool.push_back(r.a); g_pool.push_back(r.b); return; }
    g_recs.push_back(r);
    g_open = true;
}

bool prof_active() { return g_on; }

void prof_end(hipStream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open && !g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
    g_open = false;
}

}  // namespace unet

using namespace unet;

extern "C" {

int unet_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    g_open = false;
    return 0;
}

int unet_profile_select(unsigned kinds)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_kinds = kinds;
    return 0;
}

int unet_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    g_open = false;
    return 0;
}

int unet_profile_read(int kind, double *ms_total, long *launches, double *flops_total, double *exec_flops_total, double *bytes_total)
{
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0.0, fl = 0.0, ex = 0.0, by = 0.0;
    long n = 0;
    for (auto &r : g_recs) {
        if (r.kind != kind) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        ms += t; fl += r.flops; ex += r.exec_flops; by += r.bytes; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (launches) *launches = n;
    if (flops_total) *flops_total = fl;
    if (exec_flops_total) *exec_flops_total = ex;
    if (bytes_total) *bytes_total = by;
    return 0;
}

/* one line per recorded launch: kind,ms,gflop,exec_gflop,mbytes,row,tag */
int unet_profile_dump(const char *path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    FILE *f = fopen(path, "w");
    if (!f) { set_error("unet_profile_dump: cannot open %s", path); return -2; }
    fprintf(f, "kind,ms,gflop,exec_gflop,mbytes,row,tag\n");
    for (auto &r : g_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        fprintf(f, "%d,%.6f,%.6f,%.6f,%.6f,%s,%s\n", r.kind, t, r.flops / 1e9, r.exec_flops / 1e9, r.bytes / 1e6, r.row, r.tag);
    }
    fclose(f);
    return 0;
}

}  // extern "C"
