#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include <map>
#include "common.h"
#include "prof.h"
#include "options.h"

extern "C" int dh_abi_version(void) { return DH_ABI_VERSION; }

extern "C" const char* dh_error_string(int code) {
    switch (code) {
        case DH_OK: return "ok";
        case DH_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment contract violated)";
        case DH_ERR_UNSUPPORTED: return "unsupported dtype for this entry point";
        case DH_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error code";
    }
}

// ---- profiler ---------------------------------------------------------------------------------------
namespace {
struct Rec { std::string key; int e0, e1; double flops, bytes; };
struct Agg { int calls = 0; double ms = 0, flops = 0, bytes = 0; };
bool g_on = false;
std::string g_filter;                       // empty = everything, else ",name1,name2,"
std::vector<hipEvent_t> g_pool;
int g_used = 0;
int g_stride = 1, g_seen = 0;          // record every g_stride-th matching launch
std::vector<Rec> g_recs;
std::vector<std::pair<std::string, Agg>> g_out;
thread_local const char* g_tag = nullptr;
thread_local int g_dims[3] = {0, 0, 0};
std::mutex g_mu;                            // the recorder's state is shared by every host thread / stream that launches

int take_event() {
    if (g_used == (int)g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return -1;
        g_pool.push_back(e);
    }
    return g_used++;
}
}  // namespace

void dh_prof_set_tag(const char* tag) { g_tag = tag; }
void dh_prof_set_dims(int m, int n, int k) { g_dims[0] = m; g_dims[1] = n; g_dims[2] = k; }

DhProfScope::DhProfScope(const char* name, double flops, double bytes, void* stream) : s((hipStream_t)stream), rec(-1) {
    const char* tag = g_tag;
    const int dm = g_dims[0], dn = g_dims[1], dk = g_dims[2];
    g_tag = nullptr;
    g_dims[0] = g_dims[1] = g_dims[2] = 0;
    if (!g_on) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_on) return;
    std::string key(name);
    if (tag) { key += "["; key += tag; key += "]"; }
    const std::string base = key;
    if (dm > 0) { key += "{" + std::to_string(dm) + "x" + std::to_string(dn) + "x" + std::to_string(dk) + "}"; }
    // a filter entry may name the full key ("dh_linear[vocab]{1280x36541x512}"), the entry + role, or the bare entry point
    if (!g_filter.empty() && g_filter.find("," + key + ",") == std::string::npos && g_filter.find("," + base + ",") == std::string::npos &&
        g_filter.find(std::string(",") + name + ",") == std::string::npos) return;
    if ((g_seen++ % g_stride) != 0) return;
    const int e0 = take_event(), e1 = take_event();
    if (e0 < 0 || e1 < 0) return;
    g_recs.push_back(Rec{key, e0, e1, flops, bytes});
    rec = (int)g_recs.size() - 1;
    hipEventRecord(g_pool[e0], s);
}

DhProfScope::~DhProfScope() {
    if (rec < 0) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if (rec < (int)g_recs.size()) hipEventRecord(g_pool[g_recs[rec].e1], s);
}

extern "C" void dh_prof_tag(const char* tag) { g_tag = tag; }

extern "C" int dh_prof_begin(const char* filter) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_filter.clear();
    if (filter && filter[0]) { g_filter = ","; g_filter += filter; g_filter += ","; }
    g_recs.clear();
    g_out.clear();
    g_used = 0;
    g_seen = 0;
    g_on = true;
    return DH_OK;
}

extern "C" int dh_prof_set_stride(int n) {
    if (n < 1) return DH_ERR_BAD_ARG;
    g_stride = n;
    return DH_OK;
}

extern "C" int dh_prof_end(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_on = false;
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(g_pool[r.e1]) != hipSuccess) return DH_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_pool[r.e0], g_pool[r.e1]) != hipSuccess) return DH_ERR_LAUNCH;
        if (!agg.count(r.key)) order.push_back(r.key);
        Agg& a = agg[r.key];
        a.calls += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    for (const std::string& k : order) g_out.emplace_back(k, agg[k]);
    g_recs.clear();
    return DH_OK;
}

extern "C" int dh_prof_num(void) { return (int)g_out.size(); }

extern "C" int dh_prof_get(int i, char* name, int cap, int* calls, double* ms, double* flops, double* bytes) {
    if (i < 0 || i >= (int)g_out.size() || !name || cap <= 0) return DH_ERR_BAD_ARG;
    strncpy(name, g_out[i].first.c_str(), cap - 1);
    name[cap - 1] = 0;
    const Agg& a = g_out[i].second;
    if (calls) *calls = a.calls;
    if (ms) *ms = a.ms;
    if (flops) *flops = a.flops;
    if (bytes) *bytes = a.bytes;
    return DH_OK;
}

// ---- run-time options (options.h) --------------------------------------------------------------------
namespace {
struct OptDef { const char* key; const char* env; int def; };
// order == enum DhOption; the environment variable (DH_ + the key in capitals) supplies the default
const OptDef g_opt_defs[DH_OPT_COUNT] = {
    {"vocab_wreg", "DH_VOCAB_WREG", 1},
    {"vocab_areg", "DH_VOCAB_AREG", 1},
    {"vocab_wreg_transformer", "DH_VOCAB_WREG_TRANSFORMER", 0},
    {"vocab_wreg_transformer_max_rows", "DH_VOCAB_WREG_TRANSFORMER_MAX_ROWS", 640},
    {"decode_wreg", "DH_DECODE_WREG", 1},
    {"decode_wreg_min_rows", "DH_DECODE_WREG_MIN_ROWS", 1},
    {"decode_layers", "DH_DECODE_LAYERS", 0},
    {"lstm_wreg", "DH_LSTM_WREG", 1},
    {"lstm_wreg_min_rows", "DH_LSTM_WREG_MIN_ROWS", 256},
    {"f32_split", "DH_F32_SPLIT", 0},
    {"f32_planes", "DH_F32_PLANES", 1},
    {"deferred_ln", "DH_DEFERRED_LN", 1},
    {"packed_cross", "DH_PACKED_CROSS", 1},
    {"encoder_generic", "DH_ENCODER_GENERIC", 0},
    {"dist_always", "DH_DIST_ALWAYS", 0},
    {"decode_streams", "DH_DECODE_STREAMS", 1},
};
int g_opt_val[DH_OPT_COUNT];
bool g_opt_init[DH_OPT_COUNT];

int opt_default(const OptDef& d) {
    const char* e = getenv(d.env);
    if (!e || !e[0]) return d.def;
    return atoi(e);
}
int opt_find(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < DH_OPT_COUNT; ++i)
        if (strcmp(g_opt_defs[i].key, key) == 0) return i;
    return -1;
}
}  // namespace

int dh_opt(int which) {
    if (which < 0 || which >= DH_OPT_COUNT) return 0;
    if (!g_opt_init[which]) { g_opt_val[which] = opt_default(g_opt_defs[which]); g_opt_init[which] = true; }
    return g_opt_val[which];
}

extern "C" int dh_option_count(void) { return DH_OPT_COUNT; }
extern "C" const char* dh_option_name(int i) { return i >= 0 && i < DH_OPT_COUNT ? g_opt_defs[i].key : nullptr; }
extern "C" const char* dh_option_env(int i) { return i >= 0 && i < DH_OPT_COUNT ? g_opt_defs[i].env : nullptr; }
extern "C" int dh_get_option(const char* key, int* value) {
    const int i = opt_find(key);
    if (i < 0 || !value) return DH_ERR_BAD_ARG;
    *value = dh_opt(i);
    return DH_OK;
}
extern "C" int dh_set_option(const char* key, int value) {
    const int i = opt_find(key);
    if (i < 0) return DH_ERR_BAD_ARG;
    g_opt_val[i] = value;
    g_opt_init[i] = true;
    return DH_OK;
}
