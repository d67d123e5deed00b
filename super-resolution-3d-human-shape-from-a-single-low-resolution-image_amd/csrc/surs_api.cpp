// ABI housekeeping: version, last-error text, device probe.
#include "surs_common.h"

extern "C" int surs_abi_version(void) { return SURS_ABI_VERSION; }

extern "C" const char *surs_last_error(void) { return surs::err_buf(); }

extern "C" int surs_device_info(int *cu_count, char *arch_out) {
    int dev = 0;
    SURS_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SURS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch_out) {
        strncpy(arch_out, prop.gcnArchName, 31);
        arch_out[31] = 0;
    }
    return 0;
}

// ---------------------------------------------------------------- options
#include <atomic>
#include <cstdlib>
#include <mutex>

namespace {
struct OptionRow {
    const char *name, *env;
    int def;
    char env_first;   // 0: the variable holds an integer; otherwise: the option is `env_value` when the variable starts with this letter
    int env_value;
    const char *what;
};
// (order = enum Opt of surs_common.h)
const OptionRow kOptions[surs::OPT_COUNT] = {
    {"gemm_x3", "SURS_GEMM_X3", 1, 0, 0, "0: the point path's layer GEMMs on the fp32 MFMA kernel (A/B)"},
    {"gemm_big", "SURS_GEMM_BIG", 1, 0, 0, "0: the 128 x 128 layer kernel for every layer (A/B)"},
    {"split_parts", "SURS_SPLIT", 2, 'b', 3, "operand parts of the fp32-grade GEMMs: 2 (f16 x 2) or 3 (bf16 x 3; env: bf16x3)"},
    {"gemm_waves", "SURS_GEMM_WAVES", 8, 0, 0, "waves per workgroup of the 256-point layer kernel: 8 or 16"},
    {"grid_f32_columns", "SURS_GRID_F32", 1, 'g', 0, "0 (env: gemm): the fp32 sweep on the per-point layer kernels"},
    {"grid_kernel", "SURS_GRID_KERNEL", 0, 0, 0, "reduced-precision column kernel: 0 (default 12), 3, 10, 12"},
    {"grid_f32_kernel", "SURS_GRID_F32_KERNEL", 0, 0, 0, "fp32-grade column kernel: 0 (default 11), 5, 11"},
    {"r_parts", "SURS_R_PARTS", 1, 0, 0, "1: the bf16 sweep's R vectors from one f16 part; 0: the sweep's split"},
    {"grid_f32_passes", "SURS_GRID_F32_PASSES", 2, 0, 0, "passes of the fp32-grade column kernel per batch: 2 (lr, then hr) or 1"},
    {"bicubic_block", "SURS_BICUBIC_BLOCK", 1, 0, 0, "0: bicubic x2 with statistics one output per item (A/B; same bits)"},
    {"mc_emit_reclassify", "SURS_MC_EMIT_RECLASSIFY", 0, 0, 0, "1: marching cubes' emit pass classifies again (tests)"},
    {"point_runs_speculate", "SURS_POINT_RUNS_SPECULATE", 1, 0, 0, "0: surs_query_points_columns reads the run count before it launches"},
    {"conv_trace", "SURS_CONV_TRACE", 0, 0, 0, "diagnostic builds (-DSURS_CONV_TRACE): print the 3x3 kernel's phase stamps"},
    {"gemm_trace", "SURS_GEMM_TRACE", 0, 0, 0, "diagnostic builds: print the layer GEMM's stamps"},
    {"v3_trace", "SURS_V3_TRACE", 0, 0, 0, "diagnostic builds (-DSURS_V3_TRACE): print the column kernels' stamps"},
    {"conv_tall_min_wg", "SURS_CONV_TALL_MIN_WG", 256, 0, 0, "workgroups from which a stride-1 3x3 convolution takes the 8-row tile instead of 4 rows (0: never; same bits)"},
    {"conv_wide_min_wg", "SURS_CONV_WIDE_MIN_WG", 512, 0, 0, "workgroups (of 64 channels) from which such a launch takes 8 rows x 64 channels (part of the bits: another order of sums; 0: never)"},
    {"mc_ring", "SURS_MC_RING", 258, 0, 0, "planes of marching cubes' edge -> vertex-id ring = layers per chunk of a one-piece extraction + 1 (66: round 5's)"},
    {"rvec_small", "SURS_RVEC_SMALL", 1, 0, 0, "0: the affine part's GEMM of batches of <= 1024 columns on the 256 x 256-tile kernel (A/B; same bits)"},
};
std::atomic<int> g_option[surs::OPT_COUNT];
std::once_flag g_option_once;
void import_environment() {
    for (int i = 0; i < surs::OPT_COUNT; ++i) {
        const OptionRow &r = kOptions[i];
        int v = r.def;
        if (const char *e = getenv(r.env)) {   // (the library's only getenv)
            if (r.env_first) v = (e[0] == r.env_first) ? r.env_value : r.def;
            else if (e[0]) v = atoi(e);
        }
        g_option[i].store(v, std::memory_order_relaxed);
    }
}
int find_option(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < surs::OPT_COUNT; ++i)
        if (!strcmp(name, kOptions[i].name) || !strcmp(name, kOptions[i].env)) return i;
    return -1;
}
}  // namespace

int surs::option(surs::Opt id) {
    std::call_once(g_option_once, import_environment);
    return g_option[id].load(std::memory_order_relaxed);
}

extern "C" int surs_set_option(const char *name, int value) {
    const int i = find_option(name);
    SURS_REQUIRE(i >= 0, "unknown option '%s' (surs_option_name(i) lists them)", name ? name : "(null)");
    std::call_once(g_option_once, import_environment);
    g_option[i].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int surs_get_option(const char *name, int *value) {
    const int i = find_option(name);
    SURS_REQUIRE(i >= 0 && value, "unknown option '%s'", name ? name : "(null)");
    *value = surs::option((surs::Opt)i);
    return 0;
}

extern "C" const char *surs_option_name(int index) { return (index >= 0 && index < surs::OPT_COUNT) ? kOptions[index].name : nullptr; }
extern "C" const char *surs_option_help(int index) { return (index >= 0 && index < surs::OPT_COUNT) ? kOptions[index].what : nullptr; }
