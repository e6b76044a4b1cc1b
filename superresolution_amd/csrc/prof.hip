// Optional per-kernel timing: when enabled, every launcher brackets its kernel
// with two HIP events on the launch stream and tags it with its algorithmic
// FLOPs and bytes.  rst_profile_end() waits for the events and returns one JSON
// object per kernel label.  Off by default; when off a launch pays one branch.
#include "kernels.hpp"

#include <map>
#include <mutex>
#include <vector>

namespace rst {

struct ProfRecord { const char* label; double flops, bytes; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRecord> g_prof;
static std::mutex g_prof_mu;

bool prof_enabled() { return g_prof_on; }

ProfScope::ProfScope(hipStream_t s, const char* label, double flops, double bytes)
    : stream_(s), label_(label), flops_(flops), bytes_(bytes), on_(g_prof_on), e0_(nullptr) {
    if (!on_) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { on_ = false; return; }
    e0_ = e;
    (void)hipEventRecord(e, stream_);
}
ProfScope::~ProfScope() {
    if (!on_) return;
    hipEvent_t e1;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(static_cast<hipEvent_t>(e0_)); return; }
    (void)hipEventRecord(e1, stream_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(ProfRecord{label_, flops_, bytes_, static_cast<hipEvent_t>(e0_), e1});
}

static void prof_clear() {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
}

}  // namespace rst

using namespace rst;

extern "C" {

int rst_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_clear();
    g_prof_on = true;
    return RST_OK;
}

int rst_profile_end(char* json, size_t capacity) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : g_prof) {
        RST_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        RST_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        Agg& a = agg[r.label];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    prof_clear();
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += buf;
        first = false;
    }
    out += "}";
    RST_REQUIRE(json != nullptr && capacity > out.size(), "profile_end: buffer too small (%zu needed)", out.size() + 1);
    memcpy(json, out.c_str(), out.size() + 1);
    return RST_OK;
}

}  // extern "C"
