// prof.hip — see prof.h.  A bounded pool of hipEvent pairs; nbss_profile_read() synchronises and sums.
#include "prof.h"

#include <mutex>
#include <vector>

#include "../../include/nbss_hip.h"

#ifndef NBSS_EMU
namespace {
struct Pair { hipEvent_t a, b; int id; };
std::mutex g_mu;
std::vector<Pair> g_pairs;
std::vector<hipEvent_t> g_free;
long long g_mask = 0;
const size_t kMaxPairs = 200000;
thread_local hipEvent_t t_open[PK_COUNT];

hipEvent_t get_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

// a stream under graph capture takes no timing events (recording one there invalidates the capture): a captured step is simply not profiled
static bool capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

void prof_begin(int id, hipStream_t st) {
    if (!((g_mask >> id) & 1)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pairs.size() >= kMaxPairs || capturing(st)) { t_open[id] = nullptr; return; }
    hipEvent_t e = get_event();
    t_open[id] = e;
    if (e) (void)hipEventRecord(e, st);
}

void prof_end(int id, hipStream_t st) {
    if (!((g_mask >> id) & 1)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!t_open[id]) return;
    hipEvent_t e = get_event();
    if (!e) return;
    (void)hipEventRecord(e, st);
    g_pairs.push_back({t_open[id], e, id});
    t_open[id] = nullptr;
}
#else
void prof_begin(int, hipStream_t) {}
void prof_end(int, hipStream_t) {}
#endif

extern "C" {

int nbss_profile_enable(int64_t mask) {
#ifndef NBSS_EMU
    std::lock_guard<std::mutex> lk(g_mu);
    g_mask = mask;
#else
    (void)mask;
#endif
    return NBSS_OK;
}

int nbss_profile_kernels(void) { return PK_COUNT; }

const char* nbss_profile_name(int id) {
    static const char* names[PK_COUNT] = {"encoder_fwd", "fconv_fwd", "full_fwd", "mhsa_fwd", "tconvffn_fwd", "decoder_fwd", "decoder_bwd",
                                          "tconvffn_bwd", "mhsa_bwd", "fconv_bwd", "full_bwd", "wgrad", "stft_norm", "inorm_istft",
                                          "inorm_istft_bwd", "pit_sisdr", "clip_adam", "pack"};
    return (id >= 0 && id < PK_COUNT) ? names[id] : "?";
}

// total_ms[PK_COUNT], count[PK_COUNT]; waits for the recorded events, then clears them
int nbss_profile_read(double* total_ms, int64_t* count) {
    if (!total_ms || !count) return NBSS_EINVAL;
    for (int i = 0; i < PK_COUNT; ++i) { total_ms[i] = 0; count[i] = 0; }
#ifndef NBSS_EMU
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pairs) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            total_ms[p.id] += ms;
            count[p.id] += 1;
        }
        g_free.push_back(p.a);
        g_free.push_back(p.b);
    }
    g_pairs.clear();
#endif
    return NBSS_OK;
}

}  // extern "C"
