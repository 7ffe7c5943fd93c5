// C-ABI housekeeping: version + thread-local error string (include/lpm_hip.h).
#include "lpm_common.h"
#include <mutex>
#include <vector>

namespace lpm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel timing with HIP events attached to the launch itself ---------------------------------------------------------
// hipExtLaunchKernelGGL stamps a start and a stop event around ONE kernel on its own stream: the elapsed time is the kernel's
// duration (what rocprofv3 --kernel-trace reports), free of the dispatch gaps an event pair recorded around the launch call
// also brackets once a second stream is active.  bench.py switches it on for the timed steps (roofline of K2, K1 figure).
namespace {
struct TimedLaunch { int tag; hipEvent_t e0, e1; };
std::mutex g_timing_mu;
std::vector<TimedLaunch> g_timed;
bool g_timing_on = false;
}  // namespace
bool timing_request(int tag, hipEvent_t* e0, hipEvent_t* e1) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (!g_timing_on) return false;
    if (hipEventCreate(e0) != hipSuccess || hipEventCreate(e1) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    g_timed.push_back({tag, *e0, *e1});
    return true;
}
}  // namespace lpm

extern "C" void lpm_kernel_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(lpm::g_timing_mu);
    lpm::g_timing_on = on != 0;
}
extern "C" int lpm_kernel_timing_read(int tag, float* ms, int max) {
    std::lock_guard<std::mutex> lk(lpm::g_timing_mu);
    int n = 0;
    std::vector<lpm::TimedLaunch> keep;
    for (auto& t : lpm::g_timed) {
        if (t.tag != tag) { keep.push_back(t); continue; }
        float v = 0.f;
        if (hipEventSynchronize(t.e1) == hipSuccess && hipEventElapsedTime(&v, t.e0, t.e1) == hipSuccess && n < max) ms[n++] = v;
        else (void)hipGetLastError();
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    lpm::g_timed.swap(keep);
    return n;
}

extern "C" int lpm_version(void) { return LPM_VERSION; }
extern "C" const char* lpm_last_error(void) { return lpm::g_err; }

// ---- measurement: the shader clock over time ---------------------------------------------------------------------------------------------
// One wave that does nothing but sample: every `period_ticks` ticks of the constant 100 MHz counter (s_memrealtime) it stores the pair
// (constant counter, shader-clock counter).  Launched on a stream of its own before the work to be observed, it sits in one wave slot
// while other streams' kernels run; the ratio of the two counters' differences between neighbouring samples is the shader clock in
// that interval (tools/clock_trace.py: the DVFS behaviour inside one training step).  lpm_clock_marker stores one such pair from the
// stream it is launched on -- a time stamp in the sampler's time base between two pieces of work.
namespace lpm {
__global__ __launch_bounds__(64) void clock_sampler_kernel(unsigned long long* __restrict__ out, int n, int period_ticks) {
    if (threadIdx.x != 0) return;
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        unsigned long long t;
        do { t = __builtin_amdgcn_s_memrealtime(); } while (t < next);
        const unsigned long long c = __builtin_readcyclecounter();
        __builtin_nontemporal_store(t, out + 2 * i);
        __builtin_nontemporal_store(c, out + 2 * i + 1);
        next = t + (unsigned long long)period_ticks;
    }
}
__global__ __launch_bounds__(64) void clock_marker_kernel(unsigned long long* __restrict__ out, int slot) {
    if (threadIdx.x != 0) return;
    out[2 * slot] = __builtin_amdgcn_s_memrealtime();
    out[2 * slot + 1] = __builtin_readcyclecounter();
}
}  // namespace lpm

extern "C" int lpm_clock_sampler(uint64_t* out, int n, int period_ticks, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(out && n > 0 && period_ticks > 0, LPM_ERR_BADARG, "lpm_clock_sampler: null pointer or non-positive count / period");
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n, period_ticks);
    return check_launch("lpm_clock_sampler");
}
extern "C" int lpm_clock_marker(uint64_t* out, int slot, lpm_stream_t stream) {
    using namespace lpm;
    LPM_REQUIRE(out && slot >= 0, LPM_ERR_BADARG, "lpm_clock_marker: null pointer or negative slot");
    hipLaunchKernelGGL(clock_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, slot);
    return check_launch("lpm_clock_marker");
}
