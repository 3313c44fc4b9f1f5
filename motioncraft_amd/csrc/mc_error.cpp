// Last-error string of the C-ABI (thread-local).
#include <stdarg.h>
#include <stdio.h>
static thread_local char g_err[1024] = "";
void mc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mc_last_error(void) { return g_err; }

// ---- debug FLOP ledger (off by default): every launcher records the useful multiply-add work (x 2) of the launch it enqueues under the
// kernel's name, so that tools/kernel_roofline.py prices EVERY MFMA kernel of a step from what was launched instead of from a hand-kept
// table.  Expert MLPs are booked at their routing's slot count (tokens x top-k before capacity drops; the device-side
// tile count is not known to the host).
#include <map>
#include <mutex>
#include <string>
bool mc_ledger_on = false;
namespace {
struct LedgerRow { double flops = 0; long calls = 0; };
std::mutex g_ledger_mu;
std::map<std::string, LedgerRow> g_ledger;
}  // namespace
void mc_ledger_add_(const char* kernel, long grid_threads, double flops) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    LedgerRow& r = g_ledger[std::string(kernel) + "@" + std::to_string(grid_threads)];
    r.flops += flops;
    r.calls += 1;
}
extern "C" int mc_debug_flop_ledger(int enable) {          // enable != 0: clear and start recording; 0: stop
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    if (enable) g_ledger.clear();
    mc_ledger_on = enable != 0;
    return 0;
}
// text dump "kernel\tcalls\tflops\n" per row into buf (NUL-terminated, truncated at cap); returns the bytes the whole dump needs
extern "C" long mc_debug_flop_ledger_dump(char* buf, long cap) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    std::string out;
    char line[256];
    for (const auto& kv : g_ledger) {
        snprintf(line, sizeof(line), "%s\t%ld\t%.0f\n", kv.first.c_str(), kv.second.calls, kv.second.flops);
        out += line;
    }
    if (buf && cap > 0) {
        const long n = (long)out.size() < cap - 1 ? (long)out.size() : cap - 1;
        out.copy(buf, n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}
