// ABI version, thread-local error text, device properties.
#include "ggnn_common.h"
#include <atomic>
#include <cstdlib>

namespace ggnn {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    static std::atomic<int> cached{0};
    int v = cached.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;   // MI355X
    cached.store(n, std::memory_order_relaxed);
    return n;
}

// Matrix path of the fused kernels (ggnn_split.hpp): "bf16x3" (default) = f32 products as six bf16 MFMA products of 3-way split
// operands; GGNN_MATRIX=f32 = the f32 MFMA forms.  Read once: packed weight images are in the format of the mode.
bool split_matrix_path() {
    static const bool v = [] { const char* e = getenv("GGNN_MATRIX"); return !(e && (e[0] == 'f' || e[0] == 'F')); }();
    return v;
}

// Process default of the HOST policy that chooses the fused GRU forward's operand format per launch (formats.py; ggnn_split.hpp):
// 2 = "auto": two f16 pieces / three products where the operand bounds are proven, the exact bf16 x 3 split otherwise (default);
// GGNN_GRU_FMT=3 = always bf16 x 3.  The kernels never read this: the format is an argument of every pack / launch entry point.
int gru_fwd_fmt() {
    static const int v = [] { const char* e = getenv("GGNN_GRU_FMT"); return (e && (atoi(e) == 3 || e[0] == 'e' || e[0] == 'b')) ? 3 : 2; }();
    return v;
}

// Ring form of the gather-fused GRU launches (ggnn_gru_fused.hip: 0 / 1 / 2; 6 = the wide kernel of ggnn_gru_wide.hip): -1 = the
// library's default per (fan-in, format).  GGNN_GRU_FORM sets the process default; ggnn_gru_form_set() overrides it at run time
// (tests and experiments compare the forms -- bit-identical results -- inside one process).
std::atomic<int>& gru_form_override() {
    static std::atomic<int> v{[] { const char* e = getenv("GGNN_GRU_FORM"); return e ? atoi(e) : -1; }()};
    return v;
}

}  // namespace ggnn

extern "C" int ggnn_gru_form_set(int form) { return ggnn::gru_form_override().exchange(form); }
extern "C" int ggnn_matrix_path_is_split(void) { return ggnn::split_matrix_path() ? 1 : 0; }
extern "C" int ggnn_gru_forward_format(void) { return ggnn::split_matrix_path() ? ggnn::gru_fwd_fmt() : 0; }
extern "C" int ggnn_abi_version(void) { return GGNN_ABI_VERSION; }
extern "C" const char* ggnn_last_error(void) { return ggnn::error_buffer(); }
